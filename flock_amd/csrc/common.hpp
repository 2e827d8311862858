// Internal to libflockgpu: context, device arena, launch + profiling helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/flockgpu.h"

namespace flockgpu {

struct DeviceBuf {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct KernelStat {
    uint64_t launches = 0;
    double total_ms = 0.0;
};

struct GuardedAlloc {
    void *base = nullptr;
    size_t reserved = 0, mapped = 0;
    hipMemGenericAllocationHandle_t handle{};
};
struct PendingEvent {
    const char *name;
    hipEvent_t start, stop;
};

struct AsyncWorker;  // ctx.hip: the ctx's worker thread (flockgpu_*_async / flockgpu_ctx_wait)

}  // namespace flockgpu

struct flockgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int num_cus = 256;
    std::string last_error;
    // grow-only named device buffers (hash-table arenas, outputs, scan state) reused across calls
    std::map<std::string, flockgpu::DeviceBuf> arena;
    // pinned host staging for small D2H / H2D metadata
    std::map<std::string, flockgpu::DeviceBuf> pinned;
    // host-side result vectors handed out through the result structs
    std::map<std::string, std::vector<int64_t>> host_i64;
    std::map<std::string, std::vector<uint64_t>> host_u64;
    // adaptive hash-table sizing hints (rows per distinct key observed last call)
    double q5_rows_per_group = 8.0;
    double q8_rows_per_seller = 4.0;
    // profiling
    hipEvent_t sync_event = nullptr;  // for waits that must not include work queued after a copy (created on first use)
    bool profiling = false;
    std::string profile_only;  // when set: only launches of this kernel are bracketed
    std::vector<flockgpu::PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, flockgpu::KernelStat> stats;
    std::map<std::string, std::vector<float>> launch_ms;   // per-launch durations (flockgpu_profile_samples), at most 4096 per kernel
    flockgpu::AsyncWorker *worker = nullptr;  // created by the first asynchronous call
    const void *plan_in_flight = nullptr;     // the plan whose flockgpu_plan_execute_async is pending: only flockgpu_plan_wait may collect that call
    // flockgpu_malloc_guarded: pointer handed out -> {reserved base, reserved bytes, mapped bytes, allocation handle}
    std::map<void *, flockgpu::GuardedAlloc> guarded;
};

namespace flockgpu {

inline int fail(flockgpu_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    return code;
}

#define FG_HIP(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return ::flockgpu::fail((ctx), FLOCKGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                                    hipGetErrorString(e_), __FILE__, __LINE__);                             \
    } while (0)

#define FG_TRY(expr)                      \
    do {                                  \
        int rc_ = (expr);                 \
        if (rc_ != FLOCKGPU_OK) return rc_; \
    } while (0)

// Experiment knobs (A/B switches the tools/gpu_*.sh scripts flip through the environment) exist only in builds made with
// -DFLOCKGPU_EXPERIMENTAL (`FLOCKGPU_BUILD_EXPERIMENTAL=1 python -m flock_amd.build`, which writes libflockgpu_experimental.so): the
// shipped library never reads them, so no untested configuration is reachable from a production host's environment.
#ifdef FLOCKGPU_EXPERIMENTAL
inline const char *exp_env(const char *name) { return getenv(name); }
#else
inline const char *exp_env(const char *) { return nullptr; }
#endif

// The library's own device buffers.  Experimental builds with FLOCKGPU_GUARD_ARENA set place every one of them at the END of mapped
// address space (flockgpu_malloc_guarded) and give it exactly the bytes that were asked for: a kernel that reads or writes past what its
// host code requested for it faults (tools/gpu_guard_arena.sh runs the GPU tests that way).
inline bool guard_arena() {
    static const bool on = exp_env("FLOCKGPU_GUARD_ARENA") != nullptr;
    return on;
}
inline hipError_t dev_alloc(flockgpu_ctx *ctx, void **ptr, size_t bytes) {
    if (guard_arena()) return flockgpu_malloc_guarded(ctx, bytes, ptr) == FLOCKGPU_OK ? hipSuccess : hipErrorOutOfMemory;
    return hipMalloc(ptr, bytes);
}
inline void dev_free(flockgpu_ctx *ctx, void *ptr) {
    if (!ptr) return;
    if (guard_arena() && ctx->guarded.count(ptr)) (void)flockgpu_free_guarded(ctx, ptr);
    else (void)hipFree(ptr);
}

// Grow-only device buffer.  Contents are NOT preserved across a grow.
inline int arena_get(flockgpu_ctx *ctx, const char *name, size_t bytes, void **out) {
    DeviceBuf &b = ctx->arena[name];
    if (bytes == 0) bytes = 16;
    if (b.cap < bytes) {
        if (b.ptr) {
            hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) return fail(ctx, FLOCKGPU_ERR_HIP, "sync before arena grow: %s", hipGetErrorString(e));
            dev_free(ctx, b.ptr);
            b.ptr = nullptr;
            b.cap = 0;
        }
        size_t want = bytes + bytes / 8;  // slack so that slowly growing windows do not reallocate every call
        want = (want + 255) & ~size_t(255);
        if (exp_env("FLOCKGPU_ARENA_EXACT")) want = bytes;   // (ordinary memory, no slack: every growth reallocates -- address recycling without the guard)
        if (guard_arena()) {   // exactly what was asked for, its end at the end of the mapping
            // (FLOCKGPU_GUARD_SLACK = a substring: buffers whose name contains it keep the usual slack -- how the buffer behind a failing
            // guarded run is found; FLOCKGPU_GUARD_TRACE: every (re)allocation on stderr)
            const char *keep = exp_env("FLOCKGPU_GUARD_SLACK");
            if (!(keep && strstr(name, keep))) want = bytes;
            if (exp_env("FLOCKGPU_GUARD_TRACE")) fprintf(stderr, "[guard arena] %s %zu -> %zu\n", name, bytes, want);
        }
        hipError_t e = dev_alloc(ctx, &b.ptr, want);
        if (e != hipSuccess) {
            b.ptr = nullptr;
            return fail(ctx, FLOCKGPU_ERR_OOM, "arena '%s': hipMalloc(%zu) failed: %s", name, want, hipGetErrorString(e));
        }
        b.cap = want;
    }
    *out = b.ptr;
    return FLOCKGPU_OK;
}

template <typename T>
inline int arena_get_t(flockgpu_ctx *ctx, const char *name, size_t count, T **out) {
    void *p = nullptr;
    int rc = arena_get(ctx, name, count * sizeof(T), &p);
    *out = static_cast<T *>(p);
    return rc;
}

inline int pinned_get(flockgpu_ctx *ctx, const char *name, size_t bytes, void **out) {
    DeviceBuf &b = ctx->pinned[name];
    if (bytes == 0) bytes = 16;
    if (b.cap < bytes) {
        if (b.ptr) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipHostFree(b.ptr);
        }
        size_t want = (bytes * 2 + 255) & ~size_t(255);
        hipError_t e = hipHostMalloc(&b.ptr, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            b.ptr = nullptr;
            b.cap = 0;
            return fail(ctx, FLOCKGPU_ERR_OOM, "pinned '%s': hipHostMalloc(%zu) failed: %s", name, want, hipGetErrorString(e));
        }
        b.cap = want;
    }
    *out = b.ptr;
    return FLOCKGPU_OK;
}

template <typename T>
inline int pinned_get_t(flockgpu_ctx *ctx, const char *name, size_t count, T **out) {
    void *p = nullptr;
    int rc = pinned_get(ctx, name, count * sizeof(T), &p);
    *out = static_cast<T *>(p);
    return rc;
}

// One kernel launch (or a few) timed when profiling is on.  The two events are BOUND TO THE DISPATCH (hipExtLaunchKernelGGL): they carry the
// kernel's own begin and end timestamps, the ones rocprofv3's kernel trace reads.  Rounds 1-5 recorded an event either side of the launch: each
// is a packet of its own that waits for what is in front of it, which put 2-4 us on every sample (tools/micro/ext_events.hip: +1.9 us on an
// idle stream; q3's 11.8 us probe pass read 15.4 us, q8's 54.9 us sellers pass 58.8 us between back-to-back kernels).  Several launches inside
// one scope: start = the first's begin, stop = the last's end (the stop event is bound again by every launch: the later binding holds).
struct LaunchScope;
inline thread_local LaunchScope *g_launch_scope = nullptr;
struct LaunchScope {
    flockgpu_ctx *ctx;
    const char *name;
    hipEvent_t start = nullptr, stop = nullptr;
    bool on = false;
    int n_launch = 0;
    bool recorded = false;
    LaunchScope *outer = nullptr;
    LaunchScope(flockgpu_ctx *c, const char *n) : ctx(c), name(n) {
        on = ctx->profiling && (ctx->profile_only.empty() || ctx->profile_only == n);
        if (ctx->profiling && !on && ctx->profile_only.find('|') != std::string::npos)   // "a|b": the kernels a call may choose between for one step
            on = ("|" + ctx->profile_only + "|").find("|" + std::string(n) + "|") != std::string::npos;
        outer = g_launch_scope;
        recorded = exp_env("FLOCKGPU_AB_RECORDED_EVENTS") != nullptr;   // (A/B builds only: the events recorded either side of the launch, as in rounds 1-5)
        g_launch_scope = on && !recorded ? this : nullptr;   // (an inner scope that is not sampled must not lend its launches to an outer one)
        if (!on) return;
        auto take = [&]() {
            hipEvent_t e = nullptr;
            if (!ctx->event_pool.empty()) {
                e = ctx->event_pool.back();
                ctx->event_pool.pop_back();
            } else {
                (void)hipEventCreate(&e);
            }
            return e;
        };
        start = take();
        stop = take();
        if (recorded) (void)hipEventRecord(start, ctx->stream);
    }
    ~LaunchScope() {
        g_launch_scope = outer;
        if (!on) return;
        if (recorded) {
            (void)hipEventRecord(stop, ctx->stream);
            n_launch = 1;
        }
        if (n_launch > 0) {
            ctx->pending.push_back({name, start, stop});
        } else {   // nothing was launched (an error path): the events go back unused
            ctx->event_pool.push_back(start);
            ctx->event_pool.push_back(stop);
        }
    }
    LaunchScope(const LaunchScope &) = delete;
    LaunchScope &operator=(const LaunchScope &) = delete;
};

// Every kernel launch of the library goes through here (the macro below): plain outside a sampled scope, with the scope's events bound inside one.
template <typename K, typename... Args>
inline void launch_scoped(K kernel, const dim3 &grid, const dim3 &block, uint32_t shmem, hipStream_t stream, Args... args) {
    LaunchScope *s = g_launch_scope;
    if (s) {
        hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, s->n_launch == 0 ? s->start : (hipEvent_t) nullptr, s->stop, 0, args...);
        ++s->n_launch;
    } else {
        kernel<<<grid, block, shmem, stream>>>(args...);
    }
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) ::flockgpu::launch_scoped((kernel), (grid), (block), (uint32_t)(shmem), (stream), __VA_ARGS__)

inline int check_launch(flockgpu_ctx *ctx, const char *name) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, FLOCKGPU_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e));
    return FLOCKGPU_OK;
}

inline void profile_drain(flockgpu_ctx *ctx) {
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            KernelStat &s = ctx->stats[p.name];
            s.launches += 1;
            s.total_ms += ms;
            auto &v = ctx->launch_ms[p.name];
            if (v.size() < 4096) v.push_back(ms);
        }
        // An event that was bound to a dispatch is NOT used again: a later launch handed a used pair ran measurably slower (q2's flag pass 0.070 ->
        // 0.085 ms, q8's sellers pass 0.057 -> 0.077 in every entry of a process after the first; fresh pairs: as fast as the first).  Creating a pair
        // costs microseconds of host time outside the stream.
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    ctx->pending.clear();
}

// The host's wait for an answer a kernel writes into PINNED memory (a selected-row count, byte totals, a pair total): the words hold
// kPinnedPending when the kernel is queued -- every one of them is written exactly once by the kernel, with a value that is never
// kPinnedPending -- and the host spins until none does.  It learns of such a store ~6 us after it, of a finished stream (hipStreamSynchronize)
// ~12 us after (tools/micro/sync_latency.hip); no fence is asked of the kernel (a system-scope release there writes the L2 back: DESIGN
// section 10), every word is its own "ready" mark.  What the host reads afterwards are these words only; everything else the call wrote stays
// on the device, behind the stream's order.  A kernel that never answers (a fault) ends the spin after 2 ms in hipStreamSynchronize, which
// reports it.
constexpr uint64_t kPinnedPending = ~uint64_t(0);
inline void pinned_pending(uint64_t *words, int n) {
    for (int i = 0; i < n; ++i) __atomic_store_n(&words[i], kPinnedPending, __ATOMIC_RELAXED);
}
template <typename W>
inline int wait_pinned_words(flockgpu_ctx *ctx, const W *words, int n, W pending) {
    static const bool no_poll = exp_env("FLOCKGPU_NO_POLL") != nullptr;   // (A/B knob of the experimental build)
    if (!no_poll) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0;; ++spins) {
            int i = 0;
            while (i < n && __atomic_load_n(&words[i], __ATOMIC_ACQUIRE) != pending) ++i;
            if (i == n) return FLOCKGPU_OK;
            if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FLOCKGPU_OK;
}
// (32-bit words -- error words, the halves of a published 64-bit total: a half that happens to BE 0xFFFFFFFF keeps the host spinning for its
// 2 ms and then in hipStreamSynchronize; right, only slow, and it takes a total no operator accepts)
constexpr uint32_t kPinnedPending32 = ~uint32_t(0);
inline void pinned_pending32(uint32_t *words, int n) {
    for (int i = 0; i < n; ++i) __atomic_store_n(&words[i], kPinnedPending32, __ATOMIC_RELAXED);
}
inline int wait_pinned32(flockgpu_ctx *ctx, const uint32_t *words, int n) { return wait_pinned_words<uint32_t>(ctx, words, n, kPinnedPending32); }
inline int wait_pinned(flockgpu_ctx *ctx, const uint64_t *words, int n) {
    return wait_pinned_words<uint64_t>(ctx, words, n, kPinnedPending);
}
inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Asynchronous calls (flockgpu.h): `fn` runs on the ctx's worker thread, one call in flight per ctx.  ctx_submit returns
// FLOCKGPU_ERR_INVALID when a call is already in flight; ctx_wait returns the call's status (FLOCKGPU_ERR_INVALID: none submitted).
int ctx_submit(flockgpu_ctx *ctx, std::function<int()> fn);
int ctx_wait(flockgpu_ctx *ctx);

// Validates a window schedule against a relation of `rows` rows.
inline int check_windows(flockgpu_ctx *ctx, const flockgpu_windows *w, int64_t rows, const char *what) {
    if (!w || w->n_panes < 0 || w->n_windows < 0 || !w->pane_row_offsets)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: null / negative window schedule", what);
    if (w->pane_row_offsets[0] < 0 || w->pane_row_offsets[w->n_panes] > rows)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: pane offsets [%lld, %lld] outside relation of %lld rows", what,
                    (long long)w->pane_row_offsets[0], (long long)w->pane_row_offsets[w->n_panes], (long long)rows);
    for (int p = 0; p < w->n_panes; ++p)
        if (w->pane_row_offsets[p + 1] < w->pane_row_offsets[p])
            return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: pane offsets decrease at pane %d", what, p);
    for (int i = 0; i < w->n_windows; ++i)
        if (w->win_pane_lo[i] < 0 || w->win_pane_hi[i] < w->win_pane_lo[i] || w->win_pane_hi[i] > w->n_panes)
            return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: window %d pane range [%d, %d) invalid", what, i,
                        w->win_pane_lo[i], w->win_pane_hi[i]);
    return FLOCKGPU_OK;
}

// 16-byte load of a column that is read once and never again (the streaming passes): non-temporal, so the stream does not push
// the data the pass keeps coming back to (direct-address counters, tables, bitmaps) out of L2 / the Infinity Cache.
typedef int flockgpu_v4i __attribute__((ext_vector_type(4)));
// 16-byte / 4-byte stores of results the GPU does not read again (they go to the host or to a later call): non-temporal, so their
// write-back does not land on whatever kernel runs next (the q5 counters' clear cost the count pass 0.04-0.15 ms that way).
typedef unsigned int flockgpu_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store4(void *p, uint4 v) {
#if defined(FLOCKGPU_EXPERIMENTAL) && defined(FLOCKGPU_AB_PLAIN_STORES)   // (A/B builds only)
    *reinterpret_cast<uint4 *>(p) = v;
#else
    flockgpu_v4u t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<flockgpu_v4u *>(p));
#endif
}
__device__ __forceinline__ void stream_store(int32_t *p, int32_t v) {
#if defined(FLOCKGPU_EXPERIMENTAL) && defined(FLOCKGPU_AB_PLAIN_STORES)
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}
__device__ __forceinline__ void stream_store(int64_t *p, int64_t v) {
#if defined(FLOCKGPU_EXPERIMENTAL) && defined(FLOCKGPU_AB_PLAIN_STORES)
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}
__device__ __forceinline__ uint4 stream_load4u(const uint32_t *p) {
    const flockgpu_v4u v = __builtin_nontemporal_load(reinterpret_cast<const flockgpu_v4u *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int4 stream_load4(const int32_t *p) {
    const flockgpu_v4i v = __builtin_nontemporal_load(reinterpret_cast<const flockgpu_v4i *>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}

}  // namespace flockgpu
