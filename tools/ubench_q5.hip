// Design-space microbenchmark for the q5 group-by-count kernel (run on the GPU box; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_q5.hip -Iinclude -Lflock_amd -lflockgpu -o /tmp/ubench_q5
// Times, over the same 1e9-bid NEXMark auction column:
//   read      : pure streaming read (HBM ceiling for this access pattern)
//   ldshash   : LDS open-addressing pre-aggregation only (no flush)
//   ldshist   : block min/max + direct-mapped LDS histogram only (no flush)
//   hist+gadd : ldshist + flush with non-returning global atomicAdd into direct-address counters (2 windows)
//   hist+gret : same with returning atomics + per-window max
//   gatomic   : raw global atomic throughput (returning / non-returning) on an 8 MB region
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/flockgpu.h"

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int kBlock = 256;

template <int ITERS>
__global__ __launch_bounds__(kBlock) void k_read(const int32_t *__restrict__ a, int64_t n, unsigned long long *out) {
    const int64_t tile = (int64_t)kBlock * 4 * ITERS;
    const int64_t base = (int64_t)blockIdx.x * tile + threadIdx.x * 4;
    int acc = 0;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int64_t r = base + (int64_t)it * kBlock * 4;
        if (r + 4 <= n) {
            const int4 t = *reinterpret_cast<const int4 *>(a + r);
            acc += t.x ^ t.y ^ t.z ^ t.w;
        }
    }
    if (acc == 0x7fffffff) atomicAdd(out, 1ull);
}

constexpr int kSlotBits = 11, kSlots = 1 << kSlotBits;
__device__ __forceinline__ bool lds_insert(uint64_t *tab, uint32_t key, uint32_t c) {
    uint32_t s = (key * 0x9E3779B1u) >> (32 - kSlotBits);
    const uint64_t mine = ((uint64_t)key << 32) | c;
#pragma unroll 1
    for (int probe = 0; probe < 24; ++probe) {
        uint64_t cur = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == 0) {
            cur = atomicCAS(reinterpret_cast<unsigned long long *>(&tab[s]), 0ull, (unsigned long long)mine);
            if (cur == 0) return true;
        }
        if ((uint32_t)(cur >> 32) == key) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&tab[s]), (unsigned long long)c);
            return true;
        }
        s = (s + 1) & (kSlots - 1);
    }
    return false;
}

template <int ITERS, bool COLLAPSE>
__global__ __launch_bounds__(kBlock) void k_ldshash(const int32_t *__restrict__ a, int64_t n, unsigned long long *out) {
    __shared__ uint64_t lds[kSlots];
    for (int s = threadIdx.x; s < kSlots; s += kBlock) lds[s] = 0;
    __syncthreads();
    const int64_t tile = (int64_t)kBlock * 4 * ITERS;
    const int64_t base = (int64_t)blockIdx.x * tile + threadIdx.x * 4;
    const int lane = threadIdx.x & 63;
    int32_t k[ITERS][4];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int64_t r = base + (int64_t)it * kBlock * 4;
        int4 t = make_int4(0, 0, 0, 0);
        if (r + 4 <= n) t = *reinterpret_cast<const int4 *>(a + r);
        k[it][0] = t.x; k[it][1] = t.y; k[it][2] = t.z; k[it][3] = t.w;
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int64_t r = base + (int64_t)it * kBlock * 4;
        bool v[4];
        uint32_t c[4] = {1, 1, 1, 1};
        for (int j = 0; j < 4; ++j) v[j] = r + j < n;
        if (COLLAPSE) {
            const uint64_t live = __ballot(v[0]);
            if (live) {
                const int src = __ffsll((unsigned long long)live) - 1;
                const int32_t hot = __shfl(k[it][0], src, 64);
                uint32_t cnt = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool m = v[j] && k[it][j] == hot;
                    cnt += (uint32_t)__popcll((unsigned long long)__ballot(m));
                    v[j] = v[j] && !m;
                }
                if (lane == src) lds_insert(lds, (uint32_t)hot, cnt);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j)
                if (v[i] && v[j] && k[it][i] == k[it][j]) { c[i] += c[j]; v[j] = false; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (v[j]) lds_insert(lds, (uint32_t)k[it][j], c[j]);
    }
    __syncthreads();
    uint64_t acc = 0;
    for (int s = threadIdx.x; s < kSlots; s += kBlock) acc += lds[s];
    if (acc == 0x12345) atomicAdd(out, 1ull);
}

// block min / max of the tile's keys, direct-mapped LDS histogram when the range fits
constexpr int kHist = 4096;
template <int ITERS, bool COLLAPSE, int FLUSH /*0 none, 1 non-returning, 2 returning + max*/>
__global__ __launch_bounds__(kBlock) void k_ldshist(const int32_t *__restrict__ a, int64_t n, uint32_t *counters,
                                                    int32_t base_key, uint32_t range, int n_win, unsigned long long *wmax,
                                                    unsigned long long *out) {
    __shared__ uint32_t hist[kHist];
    __shared__ int32_t s_min[4], s_max[4];
    for (int s = threadIdx.x; s < kHist; s += kBlock) hist[s] = 0;
    const int64_t tile = (int64_t)kBlock * 4 * ITERS;
    const int64_t base = (int64_t)blockIdx.x * tile + threadIdx.x * 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t k[ITERS][4];
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int64_t r = base + (int64_t)it * kBlock * 4;
        int4 t = make_int4(0, 0, 0, 0);
        if (r + 4 <= n) {
            t = *reinterpret_cast<const int4 *>(a + r);
            mn = min(mn, min(min(t.x, t.y), min(t.z, t.w)));
            mx = max(mx, max(max(t.x, t.y), max(t.z, t.w)));
        }
        k[it][0] = t.x; k[it][1] = t.y; k[it][2] = t.z; k[it][3] = t.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if (lane == 0) { s_min[wave] = mn; s_max[wave] = mx; }
    __syncthreads();
    mn = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
    mx = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    const uint32_t span = (uint32_t)(mx - mn);
    if (span >= (uint32_t)kHist) {  // would fall back to the hash path in the product
        if (threadIdx.x == 0) atomicAdd(out, 1ull << 32);
        return;
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int64_t r = base + (int64_t)it * kBlock * 4;
        bool v[4];
        for (int j = 0; j < 4; ++j) v[j] = r + j < n;
        if (COLLAPSE) {
            const uint64_t live = __ballot(v[0]);
            if (live) {
                const int src = __ffsll((unsigned long long)live) - 1;
                const int32_t hot = __shfl(k[it][0], src, 64);
                uint32_t cnt = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool m = v[j] && k[it][j] == hot;
                    cnt += (uint32_t)__popcll((unsigned long long)__ballot(m));
                    v[j] = v[j] && !m;
                }
                if (lane == src) atomicAdd(&hist[hot - mn], cnt);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (v[j]) atomicAdd(&hist[k[it][j] - mn], 1u);
    }
    __syncthreads();
    if (FLUSH == 0) {
        uint32_t acc = 0;
        for (int s = threadIdx.x; s < kHist; s += kBlock) acc += hist[s];
        if (acc == 0x12345) atomicAdd(out, 1ull);
        return;
    }
    // pane of this tile -> windows (pane p belongs to windows p-1 and p, as Hopping(10,5)); emulate with 2 regions
    uint32_t best = 0;
    for (int s = threadIdx.x; s <= (int)span; s += kBlock) {
        const uint32_t c = hist[s];
        if (!c) continue;
        const uint32_t idx = (uint32_t)(mn + s - base_key);
        if (idx >= range) continue;
        for (int w = 0; w < n_win; ++w) {
            uint32_t *cnt = counters + (size_t)w * range;
            if (FLUSH == 1) {
                __hip_atomic_fetch_add(&cnt[idx], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const uint32_t old = __hip_atomic_fetch_add(&cnt[idx], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                best = max(best, old + c);
            }
        }
    }
    if (FLUSH == 2) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = max(best, (uint32_t)__shfl_xor(best, o, 64));
        if (lane == 0 && best) atomicMax(wmax, (unsigned long long)best);
    }
}

template <bool RET>
__global__ __launch_bounds__(kBlock) void k_gatomic(uint32_t *cnt, uint32_t range, int per_thread, unsigned long long *out) {
    uint32_t x = (blockIdx.x * kBlock + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t idx = (uint32_t)(((uint64_t)x * range) >> 32);
        if (RET) acc += __hip_atomic_fetch_add(&cnt[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&cnt[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (RET && acc == 0x12345) atomicAdd(out, 1ull);
}

template <typename F>
static float time_ms(hipStream_t st, int reps, F f) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv) {
    const uint64_t seconds = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1087, eps = 1000000;
    flockgpu_ctx *ctx = nullptr;
    if (flockgpu_ctx_create(0, nullptr, &ctx)) { printf("ctx: %s\n", flockgpu_last_error(ctx)); return 1; }
    flockgpu_nexmark_stream s{20260925ull, 0, eps, 1436918400000ull};
    uint64_t np, na, nb;
    flockgpu_nexmark_counts(&s, 0, seconds * eps, &np, &na, &nb);
    int32_t *auction = nullptr;
    CK(hipMalloc(&auction, nb * 4 + 64));
    if (flockgpu_nexmark_gen_bids(ctx, &s, 0, seconds * eps, auction, nullptr, nullptr, nullptr)) { printf("gen: %s\n", flockgpu_last_error(ctx)); return 1; }
    flockgpu_ctx_synchronize(ctx);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned long long *out;
    CK(hipMalloc(&out, 64));
    CK(hipMemset(out, 0, 64));
    const int64_t n = (int64_t)nb;
    const double gb = n * 4.0 / 1e9;
    printf("bids %lld (%.2f GB)\n", (long long)n, gb);
    auto report = [&](const char *name, float ms) { printf("%-28s %8.3f ms  %8.1f GB/s  %6.1f Grows/s\n", name, ms, gb / (ms * 1e-3), n / (ms * 1e-3) / 1e9); };

    report("read it=4", time_ms(st, 5, [&] { hipLaunchKernelGGL((k_read<4>), dim3((n + 4095) / 4096), dim3(kBlock), 0, st, auction, n, out); }));
    report("read it=8", time_ms(st, 5, [&] { hipLaunchKernelGGL((k_read<8>), dim3((n + 8191) / 8192), dim3(kBlock), 0, st, auction, n, out); }));
    report("read it=16", time_ms(st, 5, [&] { hipLaunchKernelGGL((k_read<16>), dim3((n + 16383) / 16384), dim3(kBlock), 0, st, auction, n, out); }));
    report("ldshash collapse", time_ms(st, 3, [&] { hipLaunchKernelGGL((k_ldshash<8, true>), dim3((n + 8191) / 8192), dim3(kBlock), 0, st, auction, n, out); }));
    report("ldshash nocollapse", time_ms(st, 3, [&] { hipLaunchKernelGGL((k_ldshash<8, false>), dim3((n + 8191) / 8192), dim3(kBlock), 0, st, auction, n, out); }));

    // direct-address counters for "2 windows": the whole key range of the column
    const int32_t base_key = 0;
    const uint32_t range = (uint32_t)(na + 2000 + 1000);
    uint32_t *counters;
    CK(hipMalloc(&counters, (size_t)range * 4 * 2));
    CK(hipMemset(counters, 0, (size_t)range * 4 * 2));
    report("ldshist collapse", time_ms(st, 3, [&] { hipLaunchKernelGGL((k_ldshist<8, true, 0>), dim3((n + 8191) / 8192), dim3(kBlock), 0, st, auction, n, counters, base_key, range, 2, out + 1, out); }));
    report("ldshist nocollapse", time_ms(st, 3, [&] { hipLaunchKernelGGL((k_ldshist<8, false, 0>), dim3((n + 8191) / 8192), dim3(kBlock), 0, st, auction, n, counters, base_key, range, 2, out + 1, out); }));
    report("ldshist it=16 collapse", time_ms(st, 3, [&] { hipLaunchKernelGGL((k_ldshist<16, true, 0>), dim3((n + 16383) / 16384), dim3(kBlock), 0, st, auction, n, counters, base_key, range, 2, out + 1, out); }));
    report("hist+gadd(2w)", time_ms(st, 3, [&] { hipLaunchKernelGGL((k_ldshist<8, true, 1>), dim3((n + 8191) / 8192), dim3(kBlock), 0, st, auction, n, counters, base_key, range, 2, out + 1, out); }));
    report("hist+gret(2w)+max", time_ms(st, 3, [&] { hipLaunchKernelGGL((k_ldshist<8, true, 2>), dim3((n + 8191) / 8192), dim3(kBlock), 0, st, auction, n, counters, base_key, range, 2, out + 1, out); }));
    report("hist+gadd(1w)", time_ms(st, 3, [&] { hipLaunchKernelGGL((k_ldshist<8, true, 1>), dim3((n + 8191) / 8192), dim3(kBlock), 0, st, auction, n, counters, base_key, range, 1, out + 1, out); }));
    report("hist16+gret(2w)+max", time_ms(st, 3, [&] { hipLaunchKernelGGL((k_ldshist<16, true, 2>), dim3((n + 16383) / 16384), dim3(kBlock), 0, st, auction, n, counters, base_key, range, 2, out + 1, out); }));

    // raw global atomic throughput on 2M counters (8 MB), 64M atomics
    const int per = 64, blocks = 4096;
    const double nat = (double)per * blocks * kBlock;
    float t1 = time_ms(st, 3, [&] { hipLaunchKernelGGL((k_gatomic<false>), dim3(blocks), dim3(kBlock), 0, st, counters, 2000000u, per, out); });
    float t2 = time_ms(st, 3, [&] { hipLaunchKernelGGL((k_gatomic<true>), dim3(blocks), dim3(kBlock), 0, st, counters, 2000000u, per, out); });
    printf("gatomic non-returning: %.3f ms -> %.1f G atomics/s ; returning: %.3f ms -> %.1f G atomics/s\n", t1, nat / t1 / 1e6, t2, nat / t2 / 1e6);
    unsigned long long h[2];
    CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    printf("fallback tiles (span too wide): %llu ; max seen %llu\n", h[0] >> 32, h[1]);
    return 0;
}
