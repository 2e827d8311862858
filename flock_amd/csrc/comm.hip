// Multi-GPU entry (include/flockgpu_comm.h): the RepartitionExec Hash([key], n) of the reference's distributed plans as an
// in-library all-to-all.  Two transports behind one interface:
//   rccl  : one process per GPU; ncclSend / ncclRecv pairs inside ncclGroupStart / End on the ctx stream (xGMI is
//           point-to-point: one message per peer per column buffer, never one per window), counts exchanged the same way
//   local : n ranks = n host threads of one process; buffers move with device-to-device copies on each rank's stream,
//           the ranks meet at host barriers (also how the exchange logic is exercised on a one-GPU box)
//   ipc   : n ranks = n PROCESSES on one node (round 5): every rank publishes a hipIpcMemHandle of its send buffer in a POSIX shared-memory
//           segment named after the 128-byte id, peers map it and pull their runs device to device; counts / reductions and the barriers
//           go through the same segment.  The protocol above it is the RCCL transport's, rank for rank and process for process: how the
//           multi-process exchange runs end to end where RCCL cannot (several ranks on one device -- a one-GPU box)
// Everything above the transport -- partition, send order, counts, regroup, the q3 / q5 / q8 drivers -- is shared.
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "../../include/flockgpu_comm.h"
#include "relops.hpp"

using namespace flockgpu;

namespace {

struct LocalGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    std::vector<const void *> send_ptr;            // per rank: base of the buffer being exchanged
    std::vector<const int64_t *> send_off;         // per rank: n + 1 byte offsets of its per-destination runs
    std::vector<const int64_t *> counts;           // per rank: host array of n * m values
    std::vector<uint64_t *> reduce;                // per rank: host array being max-reduced
    bool failed = false;                           // a rank left a collective with an error: the group is dead, nobody waits for it again
    // false: some rank of the group failed (before or while this one waited) -- the caller returns an error instead of waiting forever
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const uint64_t gen = generation;
        if (++waiting == n) {
            waiting = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        cv.wait(lk, [&] { return generation != gen || failed; });
        return generation != gen;
    }
    void poison() {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        cv.notify_all();
    }
};

// ---- ipc transport: the ranks are processes; what LocalGroup keeps in process memory lives in a POSIX shared-memory segment
constexpr int kIpcMaxRanks = 16;
constexpr size_t kIpcMsgWords = size_t(1) << 20;   // per rank: the counts / reduction message (8 MB: n ranks x (windows x (1 + Utf8 columns)) values)
struct IpcRankSlot {
    hipIpcMemHandle_t handle;     // of the ALLOCATION the send buffer lies in
    uint64_t handle_off;          // send buffer - allocation base
    int64_t send_off[kIpcMaxRanks + 1];
    int64_t msg[kIpcMsgWords];
};
struct IpcShared {
    std::atomic<uint32_t> ready;        // rank 0 has initialised the segment
    std::atomic<uint32_t> failed;       // a rank left a collective with an error: nobody waits for it again
    std::atomic<uint32_t> arrived, generation, attached;
    int32_t n;
    IpcRankSlot rank[1];                // n of them
};
static_assert(std::atomic<uint32_t>::is_always_lock_free, "the segment's atomics must work across processes");
struct IpcGroup {
    int n = 0, rank = 0;
    IpcShared *sh = nullptr;
    size_t bytes = 0;
    std::string name;
    double *timeout_s = nullptr;        // the communicator's (flockgpu_comm_set_timeout)
    IpcRankSlot &slot(int r) { return *reinterpret_cast<IpcRankSlot *>(reinterpret_cast<uint8_t *>(sh->rank) + sizeof(IpcRankSlot) * (size_t)r); }
    // false: a rank failed, or did not arrive within the time-out (it is taken for gone: the group is dead from then on)
    bool barrier() {
        if (sh->failed.load()) return false;
        const uint32_t gen = sh->generation.load();
        if (sh->arrived.fetch_add(1) + 1 == (uint32_t)n) {
            sh->arrived.store(0);
            sh->generation.fetch_add(1);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t spin = 0; sh->generation.load() == gen; ++spin) {
            if (sh->failed.load()) return false;
            if (spin > 4000) {
                std::this_thread::sleep_for(std::chrono::microseconds(spin > 40000 ? 500 : 20));
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > (timeout_s ? *timeout_s : 600.0)) {
                    sh->failed.store(1);
                    return false;
                }
            }
        }
        return true;
    }
    void poison() { if (sh) sh->failed.store(1); }
    ~IpcGroup() {
        if (!sh) return;
        const bool last = sh->attached.fetch_sub(1) == 1;
        munmap(sh, bytes);
        if (last || rank == 0) shm_unlink(name.c_str());   // (unlinking twice is harmless; the mapping of the others stays valid)
    }
};

constexpr int64_t kMaxPeerBytes = int64_t(1) << 30;  // RCCL transfers above 2 GiB per peer arrived corrupted (round 1): stay well below
// Piece k of a (source, destination) pair of `bytes` bytes: [lo, hi).  Both ends of a pair know its size from the counts exchange, so
// they walk the same pieces -- no agreement on a global round count is needed.  `cap`: the communicator's piece limit (kMaxPeerBytes;
// flockgpu_comm_set_max_piece_bytes lowers it -- on EVERY rank alike -- which is how the tests drive a relation through several
// rounds with ragged last pieces).
static void peer_piece(int64_t bytes, uint64_t k, int64_t cap, int64_t *lo, int64_t *hi) {
    *lo = std::min<int64_t>(bytes, (int64_t)k * cap);
    *hi = std::min<int64_t>(bytes, (int64_t)(k + 1) * cap);
}

}  // namespace

struct PhaseMark {
    const char *name;
    hipEvent_t ev;
};

struct flockgpu_comm {
    int n = 1, rank = 0;
    bool is_rccl = false;
    bool dead = false;   // a collective failed on this rank: every later call returns an error at once (the peers' state is unknown)
    ncclComm_t nccl = nullptr;
    std::shared_ptr<LocalGroup> local;
    std::shared_ptr<IpcGroup> ipc;
    // per-phase stream timeline of the exchange calls (flockgpu_comm_phase_*): events on the ctx stream at the phase boundaries
    bool phases_on = false;
    std::vector<PhaseMark> marks;
    std::vector<hipEvent_t> ev_pool;
    std::map<std::string, flockgpu::KernelStat> phase_stats;
    std::vector<std::string> phase_order;
    int inject = 0;      // flockgpu_comm_inject_failure: 1 = the next exchange's preparation fails, 2 = its data movement fails
    int64_t max_piece = kMaxPeerBytes;   // flockgpu_comm_set_max_piece_bytes
    double timeout_s = 600.0;   // flockgpu_comm_set_timeout: how long ONE wait behind RCCL work may last, start to end, before the peers are given up
};

namespace {

// A rank that fails inside the protocol takes its communicator down: the local group is poisoned (peers waiting at a barrier wake
// up with an error), the RCCL communicator is aborted (this rank's queued sends / receives are cancelled; peers that already wait
// on them are beyond help from here, which is why every data-dependent failure is AGREED on in the counts exchange before any
// data moves -- see exchange_relation).
void kill_comm(flockgpu_comm *c) {
    if (c->dead) return;
    c->dead = true;
    if (c->local) c->local->poison();
    if (c->ipc) c->ipc->poison();
    if (c->nccl) {
        (void)ncclCommAbort(c->nccl);
        c->nccl = nullptr;
    }
}

void phase_begin(flockgpu_comm *c) {   // marks a failed call left behind
    for (auto &m : c->marks) c->ev_pool.push_back(m.ev);
    c->marks.clear();
}
void phase_mark(flockgpu_ctx *ctx, flockgpu_comm *c, const char *name) {
    if (!c->phases_on) return;
    hipEvent_t e = nullptr;
    if (!c->ev_pool.empty()) {
        e = c->ev_pool.back();
        c->ev_pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        return;
    }
    (void)hipEventRecord(e, ctx->stream);
    c->marks.push_back({name, e});
}
// closes the call's timeline: phase i lasted from its mark to the next one (idle gaps while the host waited belong to the phase they follow)
void phase_finish(flockgpu_ctx *ctx, flockgpu_comm *c) {
    if (!c->phases_on || c->marks.empty()) return;
    phase_mark(ctx, c, "end");
    (void)hipEventSynchronize(c->marks.back().ev);
    for (size_t i = 0; i + 1 < c->marks.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->marks[i].ev, c->marks[i + 1].ev) != hipSuccess) continue;
        if (!c->phase_stats.count(c->marks[i].name)) c->phase_order.push_back(c->marks[i].name);
        flockgpu::KernelStat &st = c->phase_stats[c->marks[i].name];
        st.launches += 1;
        st.total_ms += ms;
    }
    for (auto &m : c->marks) c->ev_pool.push_back(m.ev);
    c->marks.clear();
}

// The host wait behind RCCL work.  A blind hipStreamSynchronize sits there forever when a peer died after the counts agreement (its
// sends never arrive): instead the stream is polled together with the communicator's asynchronous error state, under a deadline.
// On an error or a time-out this rank aborts its communicator -- which also cancels its own queued sends / receives, so the
// stream drains -- and returns FLOCKGPU_ERR_PEER: every rank comes back, none waits for a rank that has gone.
int comm_stream_wait(flockgpu_ctx *ctx, flockgpu_comm *c, const char *what) {
    if (!c->is_rccl || c->n == 1 || !c->nccl) {
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FLOCKGPU_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) return FLOCKGPU_OK;
        if (q != hipErrorNotReady) {
            kill_comm(c);
            return fail(ctx, FLOCKGPU_ERR_HIP, "%s: stream error behind a collective: %s", what, hipGetErrorString(q));
        }
        ncclResult_t async = ncclSuccess;
        const ncclResult_t r = ncclCommGetAsyncError(c->nccl, &async);
        if (r != ncclSuccess || (async != ncclSuccess && async != ncclInProgress)) {
            const char *msg = ncclGetErrorString(r != ncclSuccess ? r : async);
            kill_comm(c);
            (void)hipStreamSynchronize(ctx->stream);   // (the abort cancelled what was queued)
            return fail(ctx, FLOCKGPU_ERR_PEER, "%s: the communicator reported '%s' while this rank waited: a peer is gone", what, msg);
        }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > c->timeout_s) {
            kill_comm(c);
            (void)hipStreamSynchronize(ctx->stream);
            return fail(ctx, FLOCKGPU_ERR_PEER, "%s: the wait behind a collective has lasted %.0f s, the communicator's limit: a peer is taken for gone (flockgpu_comm_set_timeout)", what, waited);
        }
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(spin > 20000 ? 1000 : 50));   // the first ~ms is a busy poll: the common case ends there
    }
}

#define FG_NCCL(ctx, expr)                                                                                               \
    do {                                                                                                                 \
        ncclResult_t r_ = (expr);                                                                                        \
        if (r_ != ncclSuccess)                                                                                           \
            return fail((ctx), FLOCKGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// recv[s * m + j] = what rank s sent to this rank = its send[me * m + j].  Host arrays; synchronises.
int exchange_counts(flockgpu_ctx *ctx, flockgpu_comm *c, const int64_t *send, int m, int64_t *recv) {
    const int n = c->n;
    if (n == 1) {
        std::copy(send, send + m, recv);
        return FLOCKGPU_OK;
    }
    if (c->ipc) {
        IpcGroup &g = *c->ipc;
        if ((size_t)n * m > kIpcMsgWords) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "exchange: a counts message of %zu values exceeds the ipc transport's %zu", (size_t)n * m, kIpcMsgWords);
        std::copy(send, send + (size_t)n * m, g.slot(c->rank).msg);
        if (!g.barrier()) return fail(ctx, FLOCKGPU_ERR_PEER, "exchange: a rank of the ipc group failed");
        for (int s = 0; s < n; ++s) std::copy(g.slot(s).msg + (size_t)c->rank * m, g.slot(s).msg + (size_t)(c->rank + 1) * m, recv + (size_t)s * m);
        if (!g.barrier()) return fail(ctx, FLOCKGPU_ERR_PEER, "exchange: a rank of the ipc group failed");
        return FLOCKGPU_OK;
    }
    if (!c->is_rccl) {
        LocalGroup &g = *c->local;
        g.counts[(size_t)c->rank] = send;
        if (!g.barrier()) return fail(ctx, FLOCKGPU_ERR_PEER, "exchange: a rank of the local group failed");
        for (int s = 0; s < n; ++s) std::copy(g.counts[(size_t)s] + (size_t)c->rank * m, g.counts[(size_t)s] + (size_t)(c->rank + 1) * m, recv + (size_t)s * m);
        if (!g.barrier()) return fail(ctx, FLOCKGPU_ERR_PEER, "exchange: a rank of the local group failed");  // nobody rewrites its send array before everyone has read it
        return FLOCKGPU_OK;
    }
    int64_t *d = nullptr, *h = nullptr;
    FG_TRY(arena_get_t(ctx, "comm.counts", (size_t)2 * n * m + 2, &d));
    FG_TRY(pinned_get_t(ctx, "comm.counts", (size_t)2 * n * m + 2, &h));
    std::copy(send, send + (size_t)n * m, h);  // (the staging's previous use was read back under a synchronisation)
    FG_HIP(ctx, hipMemcpyAsync(d, h, sizeof(int64_t) * (size_t)n * m, hipMemcpyHostToDevice, ctx->stream));
    FG_NCCL(ctx, ncclGroupStart());
    for (int p = 0; p < n; ++p) {
        FG_NCCL(ctx, ncclSend(d + (size_t)p * m, (size_t)m, ncclInt64, p, c->nccl, ctx->stream));
        FG_NCCL(ctx, ncclRecv(d + (size_t)(n + p) * m, (size_t)m, ncclInt64, p, c->nccl, ctx->stream));
    }
    FG_NCCL(ctx, ncclGroupEnd());
    FG_HIP(ctx, hipMemcpyAsync(h + (size_t)n * m, d + (size_t)n * m, sizeof(int64_t) * (size_t)n * m, hipMemcpyDeviceToHost, ctx->stream));
    FG_TRY(comm_stream_wait(ctx, c, "counts exchange"));
    std::copy(h + (size_t)n * m, h + (size_t)2 * n * m, recv);
    return FLOCKGPU_OK;
}

// Variable-size all-to-all of one device buffer: bytes [send_off[p], send_off[p+1]) go to rank p, bytes from rank s land at
// [recv_off[s], recv_off[s+1]).  Offsets are host arrays of n + 1 entries.  Stream-ordered (rccl) / synchronising (local).
// skip_self: the rank's own chunk stays where it is (the caller reads it from the send buffer).
int all_to_all(flockgpu_ctx *ctx, flockgpu_comm *c, const void *send, const int64_t *send_off, void *recv, const int64_t *recv_off,
               bool skip_self = false) {
    const int n = c->n;
    const uint8_t *s8 = static_cast<const uint8_t *>(send);
    uint8_t *r8 = static_cast<uint8_t *>(recv);
    if (n == 1) {
        if (send_off[1] > send_off[0] && !skip_self) FG_HIP(ctx, hipMemcpyAsync(r8 + recv_off[0], s8 + send_off[0], (size_t)(send_off[1] - send_off[0]), hipMemcpyDeviceToDevice, ctx->stream));
        return FLOCKGPU_OK;
    }
    if (c->ipc) {
        IpcGroup &g = *c->ipc;
        // (as the local transport below: a failure between the two barriers still arrives at the second one, then reports)
        int rc = FLOCKGPU_OK;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, FLOCKGPU_ERR_HIP, "exchange: stream synchronisation before the all-to-all failed");
        IpcRankSlot &me = g.slot(c->rank);
        std::copy(send_off, send_off + n + 1, me.send_off);
        me.handle_off = 0;
        std::memset(&me.handle, 0, sizeof me.handle);
        if (rc == FLOCKGPU_OK && send_off[n] > send_off[0]) {
            void *alloc_base = nullptr;
            size_t alloc_bytes = 0;
            if (hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t *>(&alloc_base), &alloc_bytes, const_cast<void *>(send)) != hipSuccess ||
                hipIpcGetMemHandle(&me.handle, alloc_base) != hipSuccess)
                rc = fail(ctx, FLOCKGPU_ERR_HIP, "exchange: no ipc handle for the send buffer (HSA_ENABLE_IPC_MODE_LEGACY=0 ?)");
            else
                me.handle_off = (uint64_t)(s8 - static_cast<const uint8_t *>(alloc_base));
        }
        if (!g.barrier()) return fail(ctx, FLOCKGPU_ERR_PEER, "exchange: a rank of the ipc group failed");
        std::vector<void *> opened;
        for (int s = 0; s < n && rc == FLOCKGPU_OK; ++s) {
            const IpcRankSlot &peer = g.slot(s);
            const int64_t *so = peer.send_off;
            const int64_t bytes = so[c->rank + 1] - so[c->rank];
            if (bytes != recv_off[s + 1] - recv_off[s]) {
                rc = fail(ctx, FLOCKGPU_ERR_INVALID, "exchange: rank %d announced %lld bytes, sends %lld", s, (long long)(recv_off[s + 1] - recv_off[s]), (long long)bytes);
                break;
            }
            if (!bytes || (skip_self && s == c->rank)) continue;
            const uint8_t *src = nullptr;
            if (s == c->rank) {
                src = s8;
            } else {
                void *mapped = nullptr;
                hipIpcMemHandle_t h = peer.handle;
                if (hipIpcOpenMemHandle(&mapped, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                    rc = fail(ctx, FLOCKGPU_ERR_HIP, "exchange: rank %d's send buffer cannot be mapped (hipIpcOpenMemHandle)", s);
                    break;
                }
                opened.push_back(mapped);
                src = static_cast<const uint8_t *>(mapped) + peer.handle_off;
            }
            for (uint64_t k = 0;; ++k) {   // in the pieces the RCCL transport would post
                int64_t lo, hi;
                peer_piece(bytes, k, c->max_piece, &lo, &hi);
                if (hi <= lo) break;
                const hipError_t e = hipMemcpyAsync(r8 + recv_off[s] + lo, src + so[c->rank] + lo, (size_t)(hi - lo), hipMemcpyDeviceToDevice, ctx->stream);
                if (e != hipSuccess) { rc = fail(ctx, FLOCKGPU_ERR_HIP, "exchange: copy from rank %d failed: %s", s, hipGetErrorString(e)); break; }
            }
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == FLOCKGPU_OK) rc = fail(ctx, FLOCKGPU_ERR_HIP, "exchange: stream synchronisation after the all-to-all failed");
        for (void *m : opened) (void)hipIpcCloseMemHandle(m);
        const bool met = g.barrier();  // every rank has pulled its runs: send buffers may be reused
        if (rc != FLOCKGPU_OK) return rc;
        return met ? FLOCKGPU_OK : fail(ctx, FLOCKGPU_ERR_PEER, "exchange: a rank of the ipc group failed");
    }
    if (!c->is_rccl) {
        LocalGroup &g = *c->local;
        // A failure of this rank between the two barriers must not leave the peers waiting at the second one: the work in between
        // runs to its end (or to its first error), the rank still arrives at the barrier, and only then reports -- the caller
        // (exchange_relation) takes the communicator down, which wakes every peer's NEXT wait with an error.
        int rc = FLOCKGPU_OK;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, FLOCKGPU_ERR_HIP, "exchange: stream synchronisation before the all-to-all failed");  // this rank's send buffer is complete
        g.send_ptr[(size_t)c->rank] = send;
        g.send_off[(size_t)c->rank] = send_off;
        if (!g.barrier()) return fail(ctx, FLOCKGPU_ERR_PEER, "exchange: a rank of the local group failed");
        for (int s = 0; s < n && rc == FLOCKGPU_OK; ++s) {
            const int64_t *so = g.send_off[(size_t)s];
            const int64_t bytes = so[c->rank + 1] - so[c->rank];
            if (bytes != recv_off[s + 1] - recv_off[s]) {
                rc = fail(ctx, FLOCKGPU_ERR_INVALID, "exchange: rank %d announced %lld bytes, sends %lld", s, (long long)(recv_off[s + 1] - recv_off[s]), (long long)bytes);
                break;
            }
            if (bytes && !(skip_self && s == c->rank)) {   // in the pieces the RCCL transport would post (same arithmetic, tested here)
                for (uint64_t k = 0;; ++k) {
                    int64_t lo, hi;
                    peer_piece(bytes, k, c->max_piece, &lo, &hi);
                    if (hi <= lo) break;
                    const hipError_t e = hipMemcpyAsync(r8 + recv_off[s] + lo, static_cast<const uint8_t *>(g.send_ptr[(size_t)s]) + so[c->rank] + lo, (size_t)(hi - lo),
                                                        hipMemcpyDefault, ctx->stream);
                    if (e != hipSuccess) { rc = fail(ctx, FLOCKGPU_ERR_HIP, "exchange: copy from rank %d failed: %s", s, hipGetErrorString(e)); break; }
                }
            }
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == FLOCKGPU_OK) rc = fail(ctx, FLOCKGPU_ERR_HIP, "exchange: stream synchronisation after the all-to-all failed");
        const bool met = g.barrier();  // every rank has pulled its runs: send buffers may be reused
        if (rc != FLOCKGPU_OK) return rc;
        return met ? FLOCKGPU_OK : fail(ctx, FLOCKGPU_ERR_PEER, "exchange: a rank of the local group failed");
    }
    // rounds of at most c->max_piece bytes per (source, destination) pair
    int64_t biggest = 0;
    for (int p = 0; p < n; ++p) biggest = std::max({biggest, send_off[p + 1] - send_off[p], recv_off[p + 1] - recv_off[p]});
    const uint64_t rounds = (uint64_t)std::max<int64_t>(1, div_up(biggest, c->max_piece));
    for (uint64_t k = 0; k < rounds; ++k) {
        FG_NCCL(ctx, ncclGroupStart());
        for (int p = 0; p < n; ++p) {
            if (skip_self && p == c->rank) continue;
            int64_t s0, s1, r0, r1;
            peer_piece(send_off[p + 1] - send_off[p], k, c->max_piece, &s0, &s1);
            peer_piece(recv_off[p + 1] - recv_off[p], k, c->max_piece, &r0, &r1);
            if (s1 > s0) FG_NCCL(ctx, ncclSend(s8 + send_off[p] + s0, (size_t)(s1 - s0), ncclUint8, p, c->nccl, ctx->stream));
            if (r1 > r0) FG_NCCL(ctx, ncclRecv(r8 + recv_off[p] + r0, (size_t)(r1 - r0), ncclUint8, p, c->nccl, ctx->stream));
        }
        FG_NCCL(ctx, ncclGroupEnd());
    }
    return FLOCKGPU_OK;
}

// host array of m values, max over the ranks (synchronises).  `entry_rc`: this rank's status so far -- it travels as one more value,
// so a rank that failed earlier in the call still takes part (its peers do not wait for it forever) and EVERY rank returns an error.
int all_reduce_max(flockgpu_ctx *ctx, flockgpu_comm *c, int entry_rc, uint64_t *vals, int m) {
    const int n = c->n;
    if (n == 1) return entry_rc;
    if (c->dead) return entry_rc != FLOCKGPU_OK ? entry_rc : fail(ctx, FLOCKGPU_ERR_PEER, "exchange: the communicator is dead (an earlier collective failed)");
    const std::string first_error = ctx->last_error;
    std::vector<uint64_t> &buf = ctx->host_u64["comm.reduce_buf"];
    buf.assign(vals, vals + m);
    buf.push_back((uint64_t)entry_rc);
    const int mm = m + 1;
    if (c->ipc) {
        IpcGroup &g = *c->ipc;
        bool met = (size_t)mm <= kIpcMsgWords;
        if (met) {
            std::copy(buf.begin(), buf.end(), reinterpret_cast<uint64_t *>(g.slot(c->rank).msg));
            met = g.barrier();
        }
        std::vector<uint64_t> mx(buf);
        for (int s = 0; s < n && met; ++s)
            for (int j = 0; j < mm; ++j) mx[(size_t)j] = std::max(mx[(size_t)j], reinterpret_cast<const uint64_t *>(g.slot(s).msg)[j]);
        met = met && g.barrier();  // everyone has read the inputs
        if (!met) return entry_rc != FLOCKGPU_OK ? entry_rc : fail(ctx, FLOCKGPU_ERR_PEER, "exchange: a rank of the ipc group failed");
        buf = mx;
    } else if (!c->is_rccl) {
        LocalGroup &g = *c->local;
        g.reduce[(size_t)c->rank] = buf.data();
        bool met = g.barrier();
        std::vector<uint64_t> mx(buf);
        for (int s = 0; s < n && met; ++s)
            for (int j = 0; j < mm; ++j) mx[(size_t)j] = std::max(mx[(size_t)j], g.reduce[(size_t)s][j]);
        met = met && g.barrier();  // everyone has read the inputs
        if (!met) return entry_rc != FLOCKGPU_OK ? entry_rc : fail(ctx, FLOCKGPU_ERR_PEER, "exchange: a rank of the local group failed");
        buf = mx;
    } else {
        uint64_t *d = nullptr, *h = nullptr;
        int rc = arena_get_t(ctx, "comm.reduce", (size_t)mm + 2, &d);
        if (rc == FLOCKGPU_OK) rc = pinned_get_t(ctx, "comm.reduce", (size_t)mm + 2, &h);
        if (rc == FLOCKGPU_OK) rc = [&]() -> int {
            std::copy(buf.begin(), buf.end(), h);
            FG_HIP(ctx, hipMemcpyAsync(d, h, sizeof(uint64_t) * (size_t)mm, hipMemcpyHostToDevice, ctx->stream));
            FG_NCCL(ctx, ncclAllReduce(d, d, (size_t)mm, ncclUint64, ncclMax, c->nccl, ctx->stream));
            FG_HIP(ctx, hipMemcpyAsync(h, d, sizeof(uint64_t) * (size_t)mm, hipMemcpyDeviceToHost, ctx->stream));
            FG_TRY(comm_stream_wait(ctx, c, "all-reduce"));
            std::copy(h, h + mm, buf.begin());
            return FLOCKGPU_OK;
        }();
        if (rc != FLOCKGPU_OK) {   // the transport itself failed on this rank
            kill_comm(c);
            return entry_rc != FLOCKGPU_OK ? entry_rc : rc;
        }
    }
    if (entry_rc != FLOCKGPU_OK) {
        ctx->last_error = first_error;
        return entry_rc;
    }
    if (buf[(size_t)m] != 0) return fail(ctx, FLOCKGPU_ERR_PEER, "exchange: another rank failed with status %llu", (unsigned long long)buf[(size_t)m]);
    std::copy(buf.begin(), buf.begin() + m, vals);
    return FLOCKGPU_OK;
}

// ------------------------------------------------------------------ device helpers of the shuffle
// (The bytes of the Utf8 values each destination run will carry come from the send take's own scan: gather.hip utf8_run_bytes_kernel.)
// out[dst[w] + i] = in[src[w] + i] for i < len[w]: the kept windows' winners, closed up
__global__ __launch_bounds__(kBlock) void copy_runs_kernel(const int64_t *__restrict__ src, const int64_t *__restrict__ dst, const int64_t *__restrict__ len,
                                                           const int32_t *__restrict__ in_a, const uint64_t *__restrict__ in_n, int32_t *__restrict__ out_a,
                                                           uint64_t *__restrict__ out_n) {
    const int w = blockIdx.x;
    const int64_t s = src[w], d = dst[w], l = len[w];
    for (int64_t i = threadIdx.x; i < l; i += kBlock) {
        out_a[d + i] = in_a[s + i];
        out_n[d + i] = in_n[s + i];
    }
}
// run_start[d] = first send row of destination d = group_off[d * n_win]
__global__ void pick_run_starts_kernel(const int64_t *__restrict__ group_off, int32_t n_win, int32_t n, int64_t *__restrict__ run_start) {
    for (int d = threadIdx.x; d <= n; d += 64) run_start[d] = group_off[(int64_t)d * n_win];
}
// Utf8 end offsets of a gathered column, made relative to the byte start of their run (out[i] = off[i + 1] - off[start of run]) on the
// sender; on the receiver the byte position of the run in the received buffer is added back (delta per run)
__global__ __launch_bounds__(kBlock) void run_rebase_kernel(const int32_t *__restrict__ in, int64_t n, const int64_t *__restrict__ run_start,
                                                            const int64_t *__restrict__ run_delta, int32_t n_runs, int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        int32_t d = 0;
        while (d + 1 < n_runs && run_start[d + 1] <= i) ++d;
        out[i] = (int32_t)((int64_t)in[i] + run_delta[d]);
    }
}
// index[i] = position in the received (source-major) buffer of the i-th row in (window, source) order; one workgroup per run
__global__ __launch_bounds__(kBlock) void regroup_index_kernel(const int64_t *__restrict__ out_start, const int64_t *__restrict__ src_start,
                                                               int32_t n_runs, int32_t *__restrict__ index) {
    for (int32_t run = blockIdx.x; run < n_runs; run += gridDim.x) {
        const int64_t o = out_start[run], n = out_start[run + 1] - o, s0 = src_start[run];
        for (int64_t j = threadIdx.x; j < n; j += kBlock) index[o + j] = (int32_t)(s0 + j);
    }
}

// dst[out_start[run] ...] = src[src_start[run] ...] for every run, `width` bytes per row: grid = per_run shares x runs.  Runs start at
// arbitrary rows (rows are 4 or 8 bytes wide: always whole 4-byte words, at dword-aligned addresses).
// src_start[run] >= 0: a row of the received buffer; < 0: row -1 - src_start[run] of this rank's own send buffer (its chunk is not copied
// to itself first).
__global__ __launch_bounds__(kBlock) void regroup_copy_kernel(const int64_t *__restrict__ out_start, const int64_t *__restrict__ src_start,
                                                              const uint8_t *__restrict__ src, const uint8_t *__restrict__ self_src,
                                                              uint8_t *__restrict__ dst, int32_t width, int32_t per_run) {
    const int run = (int)(blockIdx.x / (unsigned)per_run);   // (runs in grid.x: grid.y stops at 65535, 64 ranks x 1087 panes do not)
    const int64_t words = (out_start[run + 1] - out_start[run]) * (width / 4);
    const int64_t s0 = src_start[run];
    const uint32_t *s = reinterpret_cast<const uint32_t *>(s0 >= 0 ? src + s0 * width : self_src + (-1 - s0) * width);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst + out_start[run] * width);
    // four words per lane and step as ONE 16-byte load and store at dword-aligned addresses (all the hardware asks of a global access; source and
    // destination are offset against each other by whatever rows lie in front of the run), the last one to three words one by one
    const int64_t quads = words >> 2;
    for (int64_t i = (int64_t)(blockIdx.x % (unsigned)per_run) * kBlock + threadIdx.x; i < quads; i += (int64_t)per_run * kBlock) {
        uint4 v;
        __builtin_memcpy(&v, s + 4 * i, 16);
        __builtin_memcpy(d + 4 * i, &v, 16);
    }
    if (blockIdx.x % (unsigned)per_run == 0 && threadIdx.x < (unsigned)(words & 3)) d[4 * quads + threadIdx.x] = s[4 * quads + threadIdx.x];
}

inline unsigned grid_for(flockgpu_ctx *ctx, int64_t n) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(div_up(n, kBlock), (int64_t)ctx->num_cus * 16));
}

// host array -> device (stream-ordered, through pinned staging owned by `name`)
template <typename T>
int upload(flockgpu_ctx *ctx, const std::string &name, const T *host, size_t n, T **dev) {
    T *h = nullptr;
    FG_TRY(arena_get_t(ctx, name.c_str(), n + 2, dev));
    FG_TRY(pinned_get_t(ctx, name.c_str(), n + 2, &h));
    std::copy(host, host + n, h);
    FG_HIP(ctx, hipMemcpyAsync(*dev, h, sizeof(T) * n, hipMemcpyHostToDevice, ctx->stream));
    return FLOCKGPU_OK;
}

struct XCol {  // a column to shuffle: fixed width (4 or 8 bytes per row) or Utf8
    const void *values = nullptr;
    const int32_t *offsets = nullptr;  // Utf8 only
    int width = 4;                     // 4 | 8; ignored for Utf8
    bool indirect = false;             // the ORIGINAL column: row i of the relation being shuffled is its row base_rows[i]
    bool utf8() const { return offsets != nullptr; }
};
struct XRecv {
    std::vector<DevColumn> cols;     // received columns, (window, source) order
    std::vector<int64_t> win_off;    // n_win + 1 row offsets of the windows
    int64_t rows = 0;
};

// RepartitionExec Hash([key], n_ranks) for one relation: every row of every window goes to rank part(key).
// `name` keys the arena buffers (they must outlive the operator that consumes the result).
// `base_rows` (may be null): the relation is a row selection of a larger one (a filter ran first); columns marked `indirect` are taken
// straight from the original columns through it -- their values are never compacted on their own, only taken in send order.
//
// Failure protocol (a per-rank error must become an error on EVERY rank, never a rank waiting for a peer that has left):
//   * `entry_rc`: the rank's status when it arrives (a stage before the exchange may have failed); a failed rank skips its
//     preparation but STILL takes part in the counts exchange, whose message carries the status -- every rank then returns;
//   * data-dependent limits of the RECEIVING side (2^31 rows / Utf8 bytes per rank) are decided from totals every rank announces to
//     every rank, so all ranks reach the same verdict before a single data byte moves;
//   * what can fail after that is the transport or the device itself: that rank takes its communicator down (kill_comm).
int exchange_relation(flockgpu_ctx *ctx, flockgpu_comm *c, int entry_rc, const std::string &name, const std::vector<XCol> &cols, int key_col, int64_t rows,
                      const flockgpu_windows *win, XRecv *out, const int32_t *base_rows = nullptr) {
    const int n = c->n, n_win = win->n_windows;
    if (c->dead) return entry_rc != FLOCKGPU_OK ? entry_rc : fail(ctx, FLOCKGPU_ERR_PEER, "exchange: the communicator is dead (an earlier collective failed)");
    phase_mark(ctx, c, "partition+take");
    int n_utf8 = 0;
    std::vector<size_t> ucols;  // the Utf8 columns: gathered together (one row list, one length pass, one scan, one emit)
    for (size_t i = 0; i < cols.size(); ++i)
        if (cols[i].utf8()) {
            ucols.push_back(i);
            ++n_utf8;
        }
    struct Sent {
        const void *values = nullptr;
        flockgpu_utf8 u{};
        int64_t bytes = 0;
        unsigned long long *d_run_bytes = nullptr, *h_run_bytes = nullptr;
    };
    std::vector<Sent> sent(cols.size());
    const int64_t *pw = nullptr;  // n * n_win + 1 group offsets, destination-major (pinned, valid after the wait)
    int64_t n_send = 0;
    int64_t *d_run_start = nullptr;
    std::vector<int64_t> run_start((size_t)n + 1, 0);

    // ---- partition, send-order gathers and the per-destination Utf8 byte counts are queued back to back; the host waits ONCE
    // for the group offsets, the Utf8 totals and the run bytes together.  (No wait for the pinned staging of `upload`: the previous
    // call that used these names ended with the operator's own synchronisation.)
    auto prepare = [&]() -> int {
        if (c->inject == 1) {
            c->inject = 0;
            return fail(ctx, FLOCKGPU_ERR_CAPACITY, "exchange of '%s': injected failure on rank %d (test hook)", name.c_str(), c->rank);
        }
        if (cols[(size_t)key_col].indirect) return fail(ctx, FLOCKGPU_ERR_INVALID, "exchange: the key column must be given compacted");
        for (auto &col : cols)
            if (col.indirect && !base_rows) return fail(ctx, FLOCKGPU_ERR_INVALID, "exchange: an indirect column without the row selection");
        if (ucols.size() > 4) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "exchange: more than four Utf8 columns in one relation");
        const int32_t *part_rows = nullptr;
        const int64_t *d_group_off = nullptr;
        // the 4-byte columns ride in the partition's emit pass (send order written directly); 8-byte and Utf8 columns are taken below
        int64_t covered = 0;
        for (int w = 0; w < n_win; ++w) covered += win->pane_row_offsets[win->win_pane_hi[w]] - win->pane_row_offsets[win->win_pane_lo[w]];
        PartPayload payload;
        std::vector<int> payload_of(cols.size(), -1);
        for (size_t i = 0; i < cols.size(); ++i) {
            if (cols[i].utf8() || cols[i].width != 4 || payload.n == 4 || cols[i].indirect) continue;
            void *p = nullptr;
            FG_TRY(arena_get(ctx, (name + ".send" + std::to_string(i)).c_str(), (size_t)covered * 4 + 16, &p));
            payload.src[payload.n] = static_cast<const int32_t *>(cols[i].values);
            payload.dst[payload.n] = static_cast<int32_t *>(p);
            payload_of[i] = payload.n++;
        }
        payload.skip_rows = true;
        for (size_t i = 0; i < cols.size(); ++i) payload.skip_rows = payload.skip_rows && payload_of[i] >= 0;
        FG_TRY(partition_by_key_async(ctx, static_cast<const int32_t *>(cols[(size_t)key_col].values), rows, win, n, &part_rows, &d_group_off, &pw, &n_send, &payload,
                                      (name + ".part").c_str()));
        const int32_t *src_rows = part_rows;   // rows of the columns' own row space, in send order
        bool any_indirect = false;
        for (auto &col : cols) any_indirect = any_indirect || col.indirect;
        if (any_indirect) {
            int32_t *comp = nullptr;
            FG_TRY(arena_get_t(ctx, (name + ".comp_rows").c_str(), (size_t)n_send + 4, &comp));
            FG_TRY(gather_i32(ctx, base_rows, part_rows, n_send, comp));
            src_rows = comp;
        }
        auto rows_of = [&](const XCol &col) { return col.indirect ? src_rows : part_rows; };
        FG_TRY(arena_get_t(ctx, (name + ".run_start").c_str(), (size_t)n + 2, &d_run_start));
        hipLaunchKernelGGL(pick_run_starts_kernel, dim3(1), dim3(64), 0, ctx->stream, d_group_off, n_win, n, d_run_start);
        FG_TRY(check_launch(ctx, "pick_run_starts_kernel"));
        Utf8MultiGather g_send;
        if (!ucols.empty()) {
            flockgpu_utf8 srcs[4];
            for (size_t j = 0; j < ucols.size(); ++j) {
                srcs[j] = flockgpu_utf8{cols[ucols[j]].offsets, static_cast<const uint8_t *>(cols[ucols[j]].values)};
                if (cols[ucols[j]].indirect != cols[ucols[0]].indirect) return fail(ctx, FLOCKGPU_ERR_INVALID, "exchange: Utf8 columns of one relation must share their row space");
            }
            FG_TRY(gather_utf8_multi_begin(ctx, (name + ".sendu").c_str(), srcs, (int)ucols.size(), rows_of(cols[ucols[0]]), n_send, &g_send));
        }
        for (size_t i = 0; i < cols.size(); ++i) {
            const XCol &col = cols[i];
            const std::string key = name + ".send" + std::to_string(i);
            if (col.utf8()) {
                FG_TRY(arena_get_t(ctx, (key + ".run_bytes").c_str(), (size_t)n + 1, &sent[i].d_run_bytes));
                FG_TRY(pinned_get_t(ctx, (key + ".run_bytes").c_str(), (size_t)n + 1, &sent[i].h_run_bytes));
                if (i == ucols.back()) {   // every Utf8 column's buffers are there: the runs' bytes of all of them from the take's scan, one launch
                    unsigned long long *rb[4] = {};
                    for (size_t j = 0; j < ucols.size(); ++j) rb[j] = sent[ucols[j]].d_run_bytes;
                    FG_TRY(gather_utf8_multi_run_bytes(ctx, g_send, d_run_start, n, rb));
                    for (size_t j = 0; j < ucols.size(); ++j)
                        FG_HIP(ctx, hipMemcpyAsync(sent[ucols[j]].h_run_bytes, sent[ucols[j]].d_run_bytes, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
                }
            } else if (payload_of[i] >= 0) {
                sent[i].values = payload.dst[payload_of[i]];
            } else {
                void *p = nullptr;
                FG_TRY(arena_get(ctx, key.c_str(), (size_t)n_send * col.width + 16, &p));
                if (col.width == 4) FG_TRY(gather_i32(ctx, static_cast<const int32_t *>(col.values), rows_of(col), n_send, static_cast<int32_t *>(p)));
                else FG_TRY(gather_i64(ctx, static_cast<const int64_t *>(col.values), rows_of(col), n_send, static_cast<int64_t *>(p)));
                sent[i].values = p;
            }
        }
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (!ucols.empty()) {
            flockgpu_utf8 outs[4];
            int64_t nb[4];
            FG_TRY(gather_utf8_multi_finish(ctx, g_send, outs, nb));
            for (size_t j = 0; j < ucols.size(); ++j) {
                sent[ucols[j]].u = outs[j];
                sent[ucols[j]].bytes = nb[j];
            }
        }
        for (int d = 0; d <= n; ++d) run_start[(size_t)d] = pw[(size_t)d * n_win];
        if (run_start[(size_t)n] != n_send) return fail(ctx, FLOCKGPU_ERR_HIP, "exchange: the partition pass covered %lld of %lld rows", (long long)run_start[(size_t)n], (long long)n_send);
        return FLOCKGPU_OK;
    };
    int rc = entry_rc;
    if (rc == FLOCKGPU_OK) rc = prepare();
    const std::string first_error = ctx->last_error;

    // ---- counts: rows per (destination, window) + bytes per Utf8 column per destination, one message per peer.  The message also
    // carries this rank's status and -- the same in every copy -- its row / byte totals for EVERY destination, from which every rank
    // computes what every rank will receive.
    phase_mark(ctx, c, "counts");
    const int m = n_win + n_utf8;                 // per-destination part
    const int tot = n * (1 + n_utf8);             // totals[d * (1 + n_utf8) + {0: rows, 1 + u: bytes of Utf8 column u}]
    const int mm = m + 1 + tot;
    std::vector<int64_t> send_counts((size_t)n * mm, 0), recv_counts((size_t)n * mm, 0);
    if (rc == FLOCKGPU_OK) {
        std::vector<int64_t> totals((size_t)tot, 0);
        for (int d = 0; d < n; ++d) {
            totals[(size_t)d * (1 + n_utf8)] = run_start[(size_t)d + 1] - run_start[(size_t)d];
            for (int w = 0; w < n_win; ++w) send_counts[(size_t)d * mm + w] = pw[(size_t)d * n_win + w + 1] - pw[(size_t)d * n_win + w];
            int u = 0;
            for (size_t i = 0; i < cols.size(); ++i)
                if (cols[i].utf8()) {
                    send_counts[(size_t)d * mm + n_win + u] = (int64_t)sent[i].h_run_bytes[d];
                    totals[(size_t)d * (1 + n_utf8) + 1 + u] = (int64_t)sent[i].h_run_bytes[d];
                    ++u;
                }
        }
        for (int d = 0; d < n; ++d) std::copy(totals.begin(), totals.end(), send_counts.begin() + (size_t)d * mm + m + 1);
    }
    for (int d = 0; d < n; ++d) send_counts[(size_t)d * mm + m] = rc;
    {
        const int rc2 = exchange_counts(ctx, c, send_counts.data(), mm, recv_counts.data());
        if (rc2 != FLOCKGPU_OK) {   // the transport failed under this rank
            kill_comm(c);
            if (rc != FLOCKGPU_OK) ctx->last_error = first_error;
            return rc != FLOCKGPU_OK ? rc : rc2;
        }
    }
    if (rc != FLOCKGPU_OK) {   // the peers have been told
        ctx->last_error = first_error;
        return rc;
    }
    for (int s = 0; s < n; ++s)
        if (recv_counts[(size_t)s * mm + m] != 0)
            return fail(ctx, FLOCKGPU_ERR_PEER, "exchange of '%s': rank %d failed with status %lld before the repartition", name.c_str(), s, (long long)recv_counts[(size_t)s * mm + m]);
    for (int d = 0; d < n; ++d) {   // the same verdict on every rank: what rank d is going to receive
        int64_t r = 0;
        for (int s = 0; s < n; ++s) r += recv_counts[(size_t)s * mm + m + 1 + (size_t)d * (1 + n_utf8)];
        if (r >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "exchange of '%s': rank %d would receive %lld rows (limit 2^31 per rank and call)", name.c_str(), d, (long long)r);
        for (int u = 0; u < n_utf8; ++u) {
            int64_t by = 0;
            for (int s = 0; s < n; ++s) by += recv_counts[(size_t)s * mm + m + 1 + (size_t)d * (1 + n_utf8) + 1 + u];
            if (by >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "exchange of '%s': a Utf8 column received by rank %d would hold %lld bytes (Arrow int32 offsets)", name.c_str(), d, (long long)by);
        }
    }
    phase_mark(ctx, c, "all_to_all+regroup");

    out->cols.assign(cols.size(), DevColumn{});
    if (n == 1) {
        // Hash([key], 1): one destination, so send order IS window order -- the send buffers are the result (the regrouping copies and
        // the Utf8 regroup take were a pure copy of them: 0.32 ms per 8e7 q5 groups, the person strings of q3 / q8 moved a second time)
        out->rows = n_send;
        out->win_off.assign(pw, pw + n_win + 1);
        for (size_t i = 0; i < cols.size(); ++i) {
            DevColumn &o = out->cols[i];
            if (cols[i].utf8()) {
                o.type = ColType::UTF8;
                o.values = sent[i].u.data;
                o.offsets = sent[i].u.offsets;
                o.bytes = sent[i].bytes;
            } else {
                o.type = cols[i].width == 4 ? ColType::I32 : ColType::I64;
                o.values = sent[i].values;
            }
        }
        return FLOCKGPU_OK;
    }

    // ---- received layout: source-major runs; windows want (window, source) order
    auto move = [&]() -> int {
    if (c->inject == 2) {
        c->inject = 0;
        return fail(ctx, FLOCKGPU_ERR_HIP, "exchange of '%s': injected transport failure on rank %d (test hook)", name.c_str(), c->rank);
    }
    std::vector<int64_t> recv_rows_off((size_t)n + 1, 0);
    for (int s = 0; s < n; ++s) {
        int64_t r = 0;
        for (int w = 0; w < n_win; ++w) r += recv_counts[(size_t)s * mm + w];
        recv_rows_off[(size_t)s + 1] = recv_rows_off[(size_t)s] + r;
    }
    const int64_t n_recv = recv_rows_off[(size_t)n];
    const int n_runs = n * n_win;
    std::vector<int64_t> out_start((size_t)n_runs + 1), src_start((size_t)n_runs + 1);
    out->win_off.assign((size_t)n_win + 1, 0);
    {
        std::vector<int64_t> src_pos((size_t)n);  // running position inside each source's chunk
        for (int s = 0; s < n; ++s) src_pos[(size_t)s] = recv_rows_off[(size_t)s];
        int64_t pos = 0;
        for (int w = 0; w < n_win; ++w) {
            out->win_off[(size_t)w] = pos;
            for (int s = 0; s < n; ++s) {
                const int64_t cnt = recv_counts[(size_t)s * mm + w];
                out_start[(size_t)w * n + s] = pos;
                src_start[(size_t)w * n + s] = src_pos[(size_t)s];
                src_pos[(size_t)s] += cnt;
                pos += cnt;
            }
        }
        out_start[(size_t)n_runs] = src_start[(size_t)n_runs] = pos;
        out->win_off[(size_t)n_win] = pos;
    }
    int64_t *d_out_start = nullptr, *d_src_start = nullptr, *d_src_self = nullptr;
    int32_t *d_index = nullptr;
    FG_TRY(upload(ctx, name + ".out_start", out_start.data(), out_start.size(), &d_out_start));
    FG_TRY(upload(ctx, name + ".src_start", src_start.data(), src_start.size(), &d_src_start));
    // fixed-width columns: the rank's own chunk never leaves the send buffer -- its runs are regrouped from there (src = -1 - row)
    std::vector<int64_t> src_self(src_start);
    for (int w = 0; w < n_win; ++w) src_self[(size_t)w * n + c->rank] = -1 - pw[(size_t)c->rank * n_win + w];
    FG_TRY(upload(ctx, name + ".src_self", src_self.data(), src_self.size(), &d_src_self));
    FG_TRY(arena_get_t(ctx, (name + ".index").c_str(), (size_t)n_recv + 4, &d_index));
    if (n_recv > 0 && n_runs > 0 && !ucols.empty()) {   // (only the Utf8 regroup reads the row-level index)
        LaunchScope ls(ctx, "regroup_index_kernel");
        hipLaunchKernelGGL(regroup_index_kernel, dim3((unsigned)std::min<int64_t>(n_runs, (int64_t)ctx->num_cus * 32)), dim3(kBlock), 0, ctx->stream, d_out_start,
                           d_src_start, n_runs, d_index);
    }
    FG_TRY(check_launch(ctx, "regroup_index_kernel"));

    // ---- one all-to-all per column buffer, then the regrouping take
    out->rows = n_recv;
    std::vector<int64_t> so((size_t)n + 1), ro((size_t)n + 1);
    int u = 0;
    // Utf8 regroup takes share one synchronisation as well
    std::vector<flockgpu_utf8> recv_u(cols.size());
    std::vector<int64_t> recv_bytes_total(cols.size(), 0);
    for (size_t i = 0; i < cols.size(); ++i) {
        const XCol &col = cols[i];
        const std::string key = name + ".recv" + std::to_string(i);
        if (!col.utf8()) {
            void *raw = nullptr, *fin = nullptr;
            FG_TRY(arena_get(ctx, (key + ".raw").c_str(), (size_t)n_recv * col.width + 16, &raw));
            FG_TRY(arena_get(ctx, key.c_str(), (size_t)n_recv * col.width + 16, &fin));
            for (int p = 0; p <= n; ++p) {
                so[(size_t)p] = run_start[(size_t)p] * col.width;
                ro[(size_t)p] = recv_rows_off[(size_t)p] * col.width;
            }
            FG_TRY(all_to_all(ctx, c, sent[i].values, so.data(), raw, ro.data(), true));
            if (n_recv > 0 && n_runs > 0) {   // (source, window) runs -> window order: contiguous runs, copied as such
                LaunchScope ls(ctx, "regroup_copy_kernel");
                const unsigned per_run = (unsigned)std::max<int64_t>(1, std::min<int64_t>(div_up(div_up(n_recv, n_runs) * col.width, (int64_t)kBlock * 16 * 4),
                                                                                           std::max<int64_t>(1, (int64_t)ctx->num_cus * 16 / n_runs)));
                hipLaunchKernelGGL(regroup_copy_kernel, dim3(per_run * (unsigned)n_runs), dim3(kBlock), 0, ctx->stream, d_out_start, d_src_self,
                                   static_cast<const uint8_t *>(raw), static_cast<const uint8_t *>(sent[i].values), static_cast<uint8_t *>(fin), col.width, (int32_t)per_run);
            }
            FG_TRY(check_launch(ctx, "regroup_copy_kernel"));
            out->cols[i].type = col.width == 4 ? ColType::I32 : ColType::I64;
            out->cols[i].values = fin;
            continue;
        }
        // Utf8: the rows' END offsets travel relative to the start of their run, the bytes as they are
        std::vector<int64_t> send_delta((size_t)n), recv_delta((size_t)n), sb((size_t)n + 1, 0), rb((size_t)n + 1, 0);
        for (int p = 0; p < n; ++p) {
            sb[(size_t)p + 1] = sb[(size_t)p] + (int64_t)sent[i].h_run_bytes[p];
            rb[(size_t)p + 1] = rb[(size_t)p] + recv_counts[(size_t)p * mm + n_win + u];
            send_delta[(size_t)p] = -sb[(size_t)p];
            recv_delta[(size_t)p] = rb[(size_t)p];
        }
        ++u;
        int64_t *d_sd = nullptr, *d_rd = nullptr, *d_rro = nullptr;
        FG_TRY(upload(ctx, key + ".sdelta", send_delta.data(), (size_t)n, &d_sd));
        FG_TRY(upload(ctx, key + ".rdelta", recv_delta.data(), (size_t)n, &d_rd));
        FG_TRY(upload(ctx, key + ".rrows", recv_rows_off.data(), (size_t)n + 1, &d_rro));
        int32_t *ends_send = nullptr, *off_recv = nullptr;
        uint8_t *bytes_recv = nullptr;
        FG_TRY(arena_get_t(ctx, (key + ".ends").c_str(), (size_t)n_send + 4, &ends_send));
        FG_TRY(arena_get_t(ctx, (key + ".rawoff").c_str(), (size_t)n_recv + 4, &off_recv));
        FG_TRY(arena_get_t(ctx, (key + ".rawbytes").c_str(), (size_t)rb[(size_t)n] + 16, &bytes_recv));
        if (n_send > 0) {
            hipLaunchKernelGGL(run_rebase_kernel, dim3(grid_for(ctx, n_send)), dim3(kBlock), 0, ctx->stream, sent[i].u.offsets + 1, n_send, d_run_start, d_sd, n, ends_send);
            FG_TRY(check_launch(ctx, "run_rebase_kernel"));
        }
        for (int p = 0; p <= n; ++p) {
            so[(size_t)p] = run_start[(size_t)p] * 4;
            ro[(size_t)p] = (recv_rows_off[(size_t)p]) * 4;
        }
        FG_HIP(ctx, hipMemsetAsync(off_recv, 0, 4, ctx->stream));
        FG_TRY(all_to_all(ctx, c, ends_send, so.data(), off_recv + 1, ro.data()));
        FG_TRY(all_to_all(ctx, c, sent[i].u.data, sb.data(), bytes_recv, rb.data()));
        if (n_recv > 0) {
            hipLaunchKernelGGL(run_rebase_kernel, dim3(grid_for(ctx, n_recv)), dim3(kBlock), 0, ctx->stream, off_recv + 1, n_recv, d_rro, d_rd, n, off_recv + 1);
            FG_TRY(check_launch(ctx, "run_rebase_kernel"));
        }
        recv_u[i] = flockgpu_utf8{off_recv, bytes_recv};
        recv_bytes_total[i] = rb[(size_t)n];
    }
    if (!ucols.empty()) {  // the regrouping take of the Utf8 columns: a permutation keeps the byte totals, so no host wait
        Utf8MultiGather g_recv;
        flockgpu_utf8 srcs[4], outs[4];
        int64_t known[4], nb[4];
        for (size_t j = 0; j < ucols.size(); ++j) {
            srcs[j] = recv_u[ucols[j]];
            known[j] = recv_bytes_total[ucols[j]];
        }
        FG_TRY(gather_utf8_multi_begin(ctx, (name + ".recvu").c_str(), srcs, (int)ucols.size(), d_index, n_recv, &g_recv));
        FG_TRY(gather_utf8_multi_finish(ctx, g_recv, outs, nb, known));
        for (size_t j = 0; j < ucols.size(); ++j) {
            DevColumn &o = out->cols[ucols[j]];
            o.type = ColType::UTF8;
            o.values = outs[j].data;
            o.offsets = outs[j].offsets;
            o.bytes = known[j];
        }
    }
    return FLOCKGPU_OK;
    };
    rc = move();
    // The single-GPU operator that follows waits on this stream with a plain synchronisation: the received bytes must have ARRIVED --
    // or the peers must be known gone -- before it starts (one more host wait per exchanged relation, on the RCCL transport only).
    if (rc == FLOCKGPU_OK && c->is_rccl && c->n > 1) rc = comm_stream_wait(ctx, c, "all-to-all");
    if (rc != FLOCKGPU_OK) kill_comm(c);   // past the agreement: the peers are already moving data
    return rc;
}

flockgpu_windows single_pane_windows(const std::vector<int64_t> &win_off, std::vector<int32_t> &lo, std::vector<int32_t> &hi) {
    const int n_win = (int)win_off.size() - 1;
    lo.resize((size_t)std::max(n_win, 1));
    hi.resize(lo.size());
    for (int w = 0; w < n_win; ++w) {
        lo[(size_t)w] = w;
        hi[(size_t)w] = w + 1;
    }
    return flockgpu_windows{win_off.data(), n_win, lo.data(), hi.data(), n_win};
}

int check_comm(flockgpu_ctx *ctx, const flockgpu_comm *c, const char *what) {
    if (!c) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: null communicator", what);
    if (c->n > 64) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than 64 ranks", what);
    return FLOCKGPU_OK;
}
// schedules whose windows are single panes [w, w + 1) (ElementWise / Tumbling): what the join shuffles need
int check_single_panes(flockgpu_ctx *ctx, const flockgpu_windows *w, const char *what) {
    for (int i = 0; i < w->n_windows; ++i)
        if (w->win_pane_hi[i] != w->win_pane_lo[i] + 1) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: the join shuffle needs single-pane windows", what);
    return FLOCKGPU_OK;
}

}  // namespace

extern "C" {

int flockgpu_comm_unique_id(uint8_t out_id[FLOCKGPU_COMM_ID_BYTES]) {
    static_assert(FLOCKGPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!out_id) return FLOCKGPU_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return FLOCKGPU_ERR_HIP;
    std::memcpy(out_id, id.internal, NCCL_UNIQUE_ID_BYTES);
    return FLOCKGPU_OK;
}

int flockgpu_comm_init_rank(flockgpu_ctx *ctx, const uint8_t id[FLOCKGPU_COMM_ID_BYTES], int n_ranks, int rank, flockgpu_comm **out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ctx, FLOCKGPU_ERR_INVALID, "comm_init_rank: bad argument");
    *out = nullptr;
    FG_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<flockgpu_comm> c(new flockgpu_comm());
    c->n = n_ranks;
    c->rank = rank;
    c->is_rccl = true;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    FG_NCCL(ctx, ncclCommInitRank(&c->nccl, n_ranks, uid, rank));
    *out = c.release();
    return FLOCKGPU_OK;
}

int flockgpu_comm_init_local(int n_ranks, flockgpu_comm **out) {
    if (!out || n_ranks < 1 || n_ranks > 64) return FLOCKGPU_ERR_INVALID;
    auto g = std::make_shared<LocalGroup>();
    g->n = n_ranks;
    g->send_ptr.assign((size_t)n_ranks, nullptr);
    g->send_off.assign((size_t)n_ranks, nullptr);
    g->counts.assign((size_t)n_ranks, nullptr);
    g->reduce.assign((size_t)n_ranks, nullptr);
    for (int r = 0; r < n_ranks; ++r) {
        out[r] = new flockgpu_comm();
        out[r]->n = n_ranks;
        out[r]->rank = r;
        out[r]->local = g;
    }
    return FLOCKGPU_OK;
}

int flockgpu_comm_init_ipc(flockgpu_ctx *ctx, const uint8_t id[FLOCKGPU_COMM_ID_BYTES], int n_ranks, int rank, flockgpu_comm **out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!id || !out || n_ranks < 1 || n_ranks > kIpcMaxRanks || rank < 0 || rank >= n_ranks) return fail(ctx, FLOCKGPU_ERR_INVALID, "comm_init_ipc: bad argument (1 to %d ranks)", kIpcMaxRanks);
    *out = nullptr;
    FG_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t h = 0xCBF29CE484222325ull;   // the segment's name: a hash of the id every rank was handed
    for (int i = 0; i < FLOCKGPU_COMM_ID_BYTES; ++i) h = (h ^ id[i]) * 0x100000001B3ull;
    auto g = std::make_shared<IpcGroup>();
    g->n = n_ranks;
    g->rank = rank;
    char name[64];
    std::snprintf(name, sizeof name, "/flockgpu-%016llx", (unsigned long long)h);
    g->name = name;
    g->bytes = sizeof(IpcShared) + sizeof(IpcRankSlot) * (size_t)n_ranks;
    int fd = -1;
    if (rank == 0) {
        (void)shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)g->bytes) != 0) {
            if (fd >= 0) close(fd);
            return fail(ctx, FLOCKGPU_ERR_OOM, "comm_init_ipc: cannot create the shared segment %s (%zu bytes)", name, g->bytes);
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {   // rank 0 may not be there yet
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= g->bytes) break;
            if (fd >= 0) close(fd);
            fd = -1;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0)
                return fail(ctx, FLOCKGPU_ERR_PEER, "comm_init_ipc: rank 0's shared segment %s did not appear", name);
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    void *p = mmap(nullptr, g->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(ctx, FLOCKGPU_ERR_OOM, "comm_init_ipc: cannot map the shared segment");
    g->sh = static_cast<IpcShared *>(p);
    if (rank == 0) {   // (a fresh segment is all zero: the counters start at 0)
        g->sh->n = n_ranks;
        g->sh->attached.store(1);
        g->sh->ready.store(1);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (!g->sh->ready.load()) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) return fail(ctx, FLOCKGPU_ERR_PEER, "comm_init_ipc: rank 0 never initialised the segment");
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (g->sh->n != n_ranks) return fail(ctx, FLOCKGPU_ERR_INVALID, "comm_init_ipc: the group has %d ranks, this rank was told %d", g->sh->n, n_ranks);
        g->sh->attached.fetch_add(1);
    }
    std::unique_ptr<flockgpu_comm> c(new flockgpu_comm());
    c->n = n_ranks;
    c->rank = rank;
    c->ipc = g;
    g->timeout_s = &c->timeout_s;
    if (!g->barrier()) return fail(ctx, FLOCKGPU_ERR_PEER, "comm_init_ipc: not every rank of the group arrived");   // collective, like ncclCommInitRank
    *out = c.release();
    return FLOCKGPU_OK;
}

void flockgpu_comm_destroy(flockgpu_comm *comm) {
    if (!comm) return;
    if (comm->nccl) (void)ncclCommDestroy(comm->nccl);
    for (auto &m : comm->marks) (void)hipEventDestroy(m.ev);
    for (hipEvent_t e : comm->ev_pool) (void)hipEventDestroy(e);
    delete comm;
}
int flockgpu_comm_rank(const flockgpu_comm *comm) { return comm ? comm->rank : -1; }
int flockgpu_comm_size(const flockgpu_comm *comm) { return comm ? comm->n : 0; }
const char *flockgpu_comm_transport(const flockgpu_comm *comm) { return !comm ? "" : (comm->is_rccl ? "rccl" : comm->ipc ? "ipc" : "local"); }

int flockgpu_comm_barrier(flockgpu_ctx *ctx, flockgpu_comm *comm) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    FG_TRY(check_comm(ctx, comm, "barrier"));
    FG_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t one = 1;
    int rc = FLOCKGPU_OK;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, FLOCKGPU_ERR_HIP, "barrier: stream synchronisation failed");
    return all_reduce_max(ctx, comm, rc, &one, 1);
}

int flockgpu_comm_set_timeout(flockgpu_comm *comm, double seconds) {
    if (!comm || !(seconds > 0)) return FLOCKGPU_ERR_INVALID;
    comm->timeout_s = seconds;
    return FLOCKGPU_OK;
}

int flockgpu_comm_set_max_piece_bytes(flockgpu_comm *comm, int64_t bytes) {
    if (!comm || bytes < 1) return FLOCKGPU_ERR_INVALID;
    comm->max_piece = std::min<int64_t>(bytes, kMaxPeerBytes);
    return FLOCKGPU_OK;
}

int flockgpu_comm_inject_failure(flockgpu_comm *comm, int where) {
    if (!comm || where < 0 || where > 2) return FLOCKGPU_ERR_INVALID;
    comm->inject = where;
    return FLOCKGPU_OK;
}

int flockgpu_comm_phase_enable(flockgpu_comm *comm, int on) {
    if (!comm) return FLOCKGPU_ERR_INVALID;
    comm->phases_on = on != 0;
    return FLOCKGPU_OK;
}
int flockgpu_comm_phase_reset(flockgpu_comm *comm) {
    if (!comm) return FLOCKGPU_ERR_INVALID;
    comm->phase_stats.clear();
    comm->phase_order.clear();
    return FLOCKGPU_OK;
}
int flockgpu_comm_phase_read(flockgpu_comm *comm, flockgpu_kernel_stat *out, int cap, int *n) {
    if (!comm || !n || (cap > 0 && !out)) return FLOCKGPU_ERR_INVALID;
    int i = 0;
    for (const std::string &name : comm->phase_order) {
        if (i < cap) {
            const flockgpu::KernelStat &st = comm->phase_stats[name];
            std::snprintf(out[i].name, sizeof out[i].name, "%s", name.c_str());
            out[i].launches = st.launches;
            out[i].total_ms = st.total_ms;
        }
        ++i;
    }
    *n = i;
    return FLOCKGPU_OK;
}

// Argument errors (null pointers, malformed schedules) are the HOST's bug and are reported at once, before any collective: a host
// that passes them on one rank only breaks the "every rank calls the same entry points" contract.  Everything that can fail at
// RUN time (device errors, capacity limits, data-dependent refusals) goes through the status-carrying collectives, so that it
// becomes an error on every rank (exchange_relation, all_reduce_max).
int flockgpu_q5_hot_items_exchange(flockgpu_ctx *ctx, flockgpu_comm *comm, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                                   flockgpu_q5_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    FG_TRY(check_comm(ctx, comm, "q5 exchange"));
    phase_begin(comm);
    if (!bid || !win || !out || bid->rows < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "q5 exchange: null argument");
    FG_TRY(check_windows(ctx, win, bid->rows, "q5 exchange"));
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_panes = win->n_panes, n_win = win->n_windows;
    phase_mark(ctx, comm, "partial");
    // stage 0 (q5.dag): HashAggregateExec mode=Partial on this rank's rows, pane by pane
    // (per 8192-row tile: one pass over the bids, pairs written straight into the pane's region -- gather.hpp: Q5TilePartial)
    Q5TilePartial part;
    int rc = q5_partial_by_tile(ctx, bid, win, &part);
    // RepartitionExec Hash([auction], n): the groups of every pane, one "window" per pane.  The schedule names the used part of every
    // region (pane 2p) and leaves the unused space behind it (pane 2p + 1) out of every window.
    std::vector<int32_t> lo((size_t)std::max(n_panes, 1)), hi(lo.size());
    for (int p = 0; p < n_panes; ++p) {
        lo[(size_t)p] = 2 * p;
        hi[(size_t)p] = 2 * p + 1;
    }
    if (rc != FLOCKGPU_OK) part.offsets.assign((size_t)2 * n_panes + 1, 0);
    const flockgpu_windows panes{part.offsets.data(), 2 * n_panes, lo.data(), hi.data(), n_panes};
    XRecv got;
    rc = exchange_relation(ctx, comm, rc, "xq5", {XCol{part.auction, nullptr, 4}, XCol{part.count, nullptr, 4}}, 0, part.capacity, &panes, &got);
    // stage 1: FinalPartitioned COUNT + MAX + join over the groups this rank owns (windows over the received panes)
    phase_mark(ctx, comm, "final");
    flockgpu_q5_result local{};
    if (rc == FLOCKGPU_OK) {
        const flockgpu_windows recv_win{got.win_off.data(), n_panes, win->win_pane_lo, win->win_pane_hi, n_win};
        rc = flockgpu_q5_hot_items_weighted(ctx, static_cast<const int32_t *>(got.cols[0].values), static_cast<const uint32_t *>(got.cols[1].values), got.rows,
                                            &recv_win, &local);
    }
    // MAX across the partitions (q5.dag: Partial MAX per partition -> CoalescePartitions -> Final MAX), 8 bytes per window
    phase_mark(ctx, comm, "all_reduce_max");
    std::vector<uint64_t> &gmax = ctx->host_u64["xq5.win_max"];
    if (rc == FLOCKGPU_OK) gmax.assign(local.win_max, local.win_max + n_win);
    else gmax.assign((size_t)n_win, 0);
    rc = all_reduce_max(ctx, comm, rc, gmax.data(), n_win);
    if (rc != FLOCKGPU_OK) {
        phase_finish(ctx, comm);
        return rc;
    }
    // the join num = maxn: windows whose local maximum is below the global one keep none of their rows.  Winners of a
    // window are contiguous; keep-runs are copied down on the device.
    std::vector<int64_t> &offs = ctx->host_i64["xq5.win_out_offsets"];
    offs.assign((size_t)n_win + 1, 0);
    std::vector<int64_t> runs((size_t)3 * std::max(n_win, 1), 0);  // src | dst | len per window
    int64_t pos = 0;
    bool all_kept = true;
    for (int w = 0; w < n_win; ++w) {
        const int64_t a = local.win_out_offsets[w], b = local.win_out_offsets[w + 1];
        const bool keep = b > a && gmax[(size_t)w] > 0 && local.win_max[w] == gmax[(size_t)w];
        all_kept = all_kept && (keep || b == a);
        offs[(size_t)w] = pos;
        runs[(size_t)w] = a;
        runs[(size_t)n_win + w] = pos;
        runs[(size_t)2 * n_win + w] = keep ? b - a : 0;
        if (keep) pos += b - a;
    }
    offs[(size_t)n_win] = pos;
    std::vector<uint64_t> &grp = ctx->host_u64["xq5.win_groups"];
    grp.assign(local.win_groups, local.win_groups + n_win);
    out->win_out_offsets = offs.data();
    out->win_max = gmax.data();
    out->win_groups = grp.data();
    out->rows = pos;
    if (all_kept) {   // every window's local maximum is the global one (always so on one rank): the local winners ARE the result
        out->auction = local.auction;
        out->num = local.num;
        phase_finish(ctx, comm);
        return FLOCKGPU_OK;
    }
    int32_t *o_a = nullptr;
    uint64_t *o_n = nullptr;
    int64_t *d_runs = nullptr;
    FG_TRY(arena_get_t(ctx, "xq5.out_auction", (size_t)pos + 2, &o_a));
    FG_TRY(arena_get_t(ctx, "xq5.out_num", (size_t)pos + 2, &o_n));
    if (n_win > 0 && pos > 0) {
        FG_TRY(upload(ctx, "xq5.runs", runs.data(), runs.size(), &d_runs));
        hipLaunchKernelGGL(copy_runs_kernel, dim3((unsigned)n_win), dim3(kBlock), 0, ctx->stream, d_runs, d_runs + n_win, d_runs + 2 * n_win, local.auction,
                           local.num, o_a, o_n);
        FG_TRY(check_launch(ctx, "copy_runs_kernel"));
    }
    out->auction = o_a;
    out->num = o_n;
    phase_finish(ctx, comm);
    return FLOCKGPU_OK;
}

int flockgpu_q3_join_exchange(flockgpu_ctx *ctx, flockgpu_comm *comm, const flockgpu_auction_cols *auction, const flockgpu_windows *auction_win,
                              const flockgpu_person_cols *person, const flockgpu_windows *person_win, int64_t category_lit,
                              const char *const *state_lits, int n_state_lits, flockgpu_q3_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    FG_TRY(check_comm(ctx, comm, "q3 exchange"));
    phase_begin(comm);
    if (!auction || !person || !auction_win || !person_win || !out) return fail(ctx, FLOCKGPU_ERR_INVALID, "q3 exchange: null argument");
    FG_TRY(check_windows(ctx, auction_win, auction->rows, "q3 exchange"));
    FG_TRY(check_windows(ctx, person_win, person->rows, "q3 exchange"));
    FG_TRY(check_single_panes(ctx, auction_win, "q3 exchange"));
    FG_TRY(check_single_panes(ctx, person_win, "q3 exchange"));
    if (auction_win->n_windows != person_win->n_windows) return fail(ctx, FLOCKGPU_ERR_INVALID, "q3 exchange: the two schedules differ in their window count");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    phase_mark(ctx, comm, "stage0 filters");
    // stage 0 of planner.rs:152-171 runs BEFORE the repartition: FilterExec category = lit on the auctions, state = a OR b OR ... on the
    // persons; only the rows they keep are partitioned and travel (the join below filters again: a no-op on what arrives)
    Q3Stage0 f;
    int32_t *a_key = nullptr, *p_key = nullptr;   // the partition keys of the kept rows (the other columns are taken through the row lists)
    int rc = [&]() -> int {
        FG_TRY(q3_stage0_filters(ctx, auction, auction_win, person, person_win, category_lit, state_lits, n_state_lits, &f));
        FG_TRY(arena_get_t(ctx, "xq3a.key", (size_t)f.n_auctions + 4, &a_key));
        FG_TRY(arena_get_t(ctx, "xq3p.key", (size_t)f.n_persons + 4, &p_key));
        FG_TRY(gather_i32(ctx, auction->seller, f.auction_rows, f.n_auctions, a_key));
        FG_TRY(gather_i32(ctx, person->p_id, f.person_rows, f.n_persons, p_key));
        return FLOCKGPU_OK;
    }();
    if (rc != FLOCKGPU_OK) {   // a failed rank still walks through both exchanges (with empty schedules of the right shape)
        f.auction_off.assign((size_t)auction_win->n_windows + 1, 0);
        f.person_off.assign((size_t)person_win->n_windows + 1, 0);
    }
    std::vector<int32_t> falo, fahi, fplo, fphi;
    const flockgpu_windows faw = single_pane_windows(f.auction_off, falo, fahi), fpw = single_pane_windows(f.person_off, fplo, fphi);
    XRecv a, p;
    rc = exchange_relation(ctx, comm, rc, "xq3a", {XCol{auction->a_id, nullptr, 4, true}, XCol{a_key, nullptr, 4}, XCol{auction->category, nullptr, 4, true}}, 1,
                           f.n_auctions, &faw, &a, f.auction_rows);
    rc = exchange_relation(ctx, comm, rc, "xq3p",
                           {XCol{p_key, nullptr, 4}, XCol{person->name.data, person->name.offsets, 0, true}, XCol{person->city.data, person->city.offsets, 0, true},
                            XCol{person->state.data, person->state.offsets, 0, true}},
                           0, f.n_persons, &fpw, &p, f.person_rows);
    phase_mark(ctx, comm, "join");
    if (rc == FLOCKGPU_OK) {
        std::vector<int32_t> alo, ahi, plo, phi;
        const flockgpu_windows aw = single_pane_windows(a.win_off, alo, ahi), pw = single_pane_windows(p.win_off, plo, phi);
        auto u = [](const DevColumn &c) { return flockgpu_utf8{c.offsets, static_cast<const uint8_t *>(c.values)}; };
        const flockgpu_auction_cols ac{static_cast<const int32_t *>(a.cols[0].values), static_cast<const int32_t *>(a.cols[1].values),
                                       static_cast<const int32_t *>(a.cols[2].values), a.rows};
        const flockgpu_person_cols pc{static_cast<const int32_t *>(p.cols[0].values), u(p.cols[1]), u(p.cols[2]), u(p.cols[3]), p.rows};
        rc = flockgpu_q3_join(ctx, &ac, &aw, &pc, &pw, category_lit, state_lits, n_state_lits, out);
    }
    phase_finish(ctx, comm);
    return rc;
}

int flockgpu_q8_join_exchange(flockgpu_ctx *ctx, flockgpu_comm *comm, const flockgpu_person_cols *person, const flockgpu_windows *person_win,
                              const flockgpu_auction_cols *auction, const flockgpu_windows *auction_win, flockgpu_q8_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    FG_TRY(check_comm(ctx, comm, "q8 exchange"));
    phase_begin(comm);
    if (!auction || !person || !auction_win || !person_win || !out) return fail(ctx, FLOCKGPU_ERR_INVALID, "q8 exchange: null argument");
    FG_TRY(check_windows(ctx, auction_win, auction->rows, "q8 exchange"));
    FG_TRY(check_windows(ctx, person_win, person->rows, "q8 exchange"));
    FG_TRY(check_single_panes(ctx, auction_win, "q8 exchange"));
    FG_TRY(check_single_panes(ctx, person_win, "q8 exchange"));
    if (auction_win->n_windows != person_win->n_windows) return fail(ctx, FLOCKGPU_ERR_INVALID, "q8 exchange: the two schedules differ in their window count");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    XRecv a, p;
    int rc = exchange_relation(ctx, comm, FLOCKGPU_OK, "xq8p", {XCol{person->p_id, nullptr, 4}, XCol{person->name.data, person->name.offsets, 0}}, 0, person->rows,
                               person_win, &p);
    // q8.dag: HashAggregateExec(Partial) gby=[seller] before the repartition -- the sellers travel as their per-tile DISTINCT values
    // (3/4 of a window's auctions name one of a few hot sellers), the FinalPartitioned DISTINCT is the join's own seller set
    phase_mark(ctx, comm, "partial distinct");
    const int32_t *sellers = nullptr;
    std::vector<int64_t> seller_off;
    int64_t n_sellers = 0;
    if (rc == FLOCKGPU_OK) rc = tile_distinct_i32(ctx, "xq8a.distinct", auction->seller, auction->rows, auction_win, &sellers, &seller_off, &n_sellers);
    if (rc != FLOCKGPU_OK) seller_off.assign((size_t)auction_win->n_windows + 1, 0);
    std::vector<int32_t> slo, shi;
    const flockgpu_windows seller_win = single_pane_windows(seller_off, slo, shi);
    rc = exchange_relation(ctx, comm, rc, "xq8a", {XCol{sellers, nullptr, 4}}, 0, n_sellers, &seller_win, &a);
    phase_mark(ctx, comm, "join");
    if (rc == FLOCKGPU_OK) {
        std::vector<int32_t> alo, ahi, plo, phi;
        const flockgpu_windows aw = single_pane_windows(a.win_off, alo, ahi), pw = single_pane_windows(p.win_off, plo, phi);
        const flockgpu_person_cols pc{static_cast<const int32_t *>(p.cols[0].values),
                                      flockgpu_utf8{p.cols[1].offsets, static_cast<const uint8_t *>(p.cols[1].values)}, {nullptr, nullptr}, {nullptr, nullptr}, p.rows};
        const flockgpu_auction_cols ac{nullptr, static_cast<const int32_t *>(a.cols[0].values), nullptr, a.rows};
        rc = flockgpu_q8_join(ctx, &pc, &pw, &ac, &aw, out);
    }
    phase_finish(ctx, comm);
    return rc;
}

}  // extern "C"
