// NEXMark q11 (user sessions) for gfx950 -- SURVEY.md section 8(f) rank 1 -- over a whole run of epochs at once:
//   Window::Session(10 s) (benchmarks/src/nexmark/main.rs:120), keyed by bidder (main.rs:346-349), and per closed session
//   SELECT bidder, COUNT(*) AS bid_count, MIN(b_date_time) AS start_time, MAX(b_date_time) AS end_time
//   FROM bid GROUP BY bidder                                            (benchmarks/src/nexmark/query/q11.sql)
//
// The reference walks the epochs one by one (flock-function/src/aws/window/session.rs:205-262): the epoch's bids are
// split into one partition per bidder (HashDiff repartition), a partition joins the bidder's open session unless its
// FIRST bid lies more than `timeout` whole seconds after the session's LAST bid (then the session is closed and a new
// one starts, :64-134), and after that every open session whose last bid is more than `timeout` whole seconds older
// than the epoch clock  BASE_TIME/1000 + epoch  is closed (:144-178).  Sessions closed in an epoch are sent to the
// query together; sessions still open after the last epoch are never sent.
//
// Every decision above only looks at two NEIGHBOURING partitions of one bidder, so the walk unrolls into data-parallel
// passes over the bids grouped by bidder (HBM-bound integer work, no MFMA):
//   group  : stable radix sort of (bidder, row) (sort.hip)
//   pack   : (epoch, b_date_time - reference) of every row in 32 bits, carried through the sort as its payload, so that the sorted
//            rows need no 8-byte gather at random afterwards (round 5; 32-byte sectors for 8-byte values: 3.2 GB for 0.8 GB);
//            a stream whose times or epoch count do not fit takes  gather : b_date_time and epoch of every sorted row
//   cut    : boundary j between sorted rows j-1 and j is a cut when the bidder changes, or the epoch changes and
//            (a) the session timed out before the next partition's epoch: max(ep_a, sec_a - base_s + timeout + 1) < ep_b
//            or (b) sec_b - sec_a > timeout.                 count -> scan -> emit over 2048-boundary tiles
//   emit   : per cut: where the session starts, whose it is, in which epoch the one before it closed; MIN / MAX of
//            every (thread, session) piece by one atomic each
//   merge  : the query groups by bidder, so two sessions of one bidder closed in the SAME epoch (event time lagging
//            the epoch clock by more than the timeout) are one output row
//   order  : stable radix sort of the sessions by closing epoch ("never" last), per-epoch offsets, take.
#include <algorithm>

#include "sort.hpp"

using namespace flockgpu;

namespace {

constexpr int kCutItems = 8;
constexpr int kCutTile = kBlock * kCutItems;  // 2048 boundaries; thread t owns boundaries  t*8 .. t*8+7  of its tile

struct SessionParams {
    int64_t base_s;     // BASE_TIME / 1000
    int32_t timeout_s;
    int32_t n_epochs;
    int32_t t_bits;     // packed payload: low t_bits = b_date_time - *t_ref, the bits above = epoch (0: rows carry ts_s / ep_s instead)
};

__device__ __forceinline__ int64_t close_clock(int32_t ep_a, int64_t ts_a, const SessionParams &p) {
    const int64_t t = ts_a / 1000 - p.base_s + p.timeout_s + 1;  // first epoch whose clock is > timeout past the last bid
    return t > ep_a ? t : (int64_t)ep_a;                          // the check runs from the partition's own epoch on
}

// a = last bid of a partition, b = first bid of the same bidder's next row
__device__ __forceinline__ bool same_session(int32_t ep_a, int64_t ts_a, int32_t ep_b, int64_t ts_b, const SessionParams &p) {
    if (ep_a == ep_b) return true;
    if (close_clock(ep_a, ts_a, p) < (int64_t)ep_b) return false;
    return !(ts_b / 1000 - ts_a / 1000 > (int64_t)p.timeout_s);
}

// ts_s[i] = b_date_time[rows[i]], ep_s[i] = epoch of rows[i] (last epoch whose first row is <= rows[i])
__global__ __launch_bounds__(kBlock) void q11_gather_kernel(const int64_t *__restrict__ dt, const int32_t *__restrict__ rows,
                                                            int64_t n, const int64_t *__restrict__ epoch_off, int32_t n_epochs,
                                                            int64_t *__restrict__ ts_s, int32_t *__restrict__ ep_s) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int64_t r = rows[i];
    int32_t lo = 0, hi = n_epochs;  // epoch_off[lo] <= r < epoch_off[hi]
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (epoch_off[mid] <= r) lo = mid;
        else hi = mid;
    }
    ts_s[i] = dt[r];
    ep_s[i] = lo;
}

// Packed payload of row i:  epoch << t_bits | (b_date_time[i] - t_ref),  t_ref = b_date_time[0] - 2^(t_bits-1)  (the stream's first
// time in the middle of the representable span; all arithmetic modulo 2^64, so a value that fits is restored exactly).  A row that does
// not fit raises `unfit` and the host falls back to the gather.  Rows arrive in epoch order, so a thread finds its first row's epoch by
// bisection and walks on from there.  The pass reads the bidder column as well and leaves its minimum / maximum (the sort's digit range):
// one launch instead of key_min_max's.
constexpr int kPackItems = 8;
constexpr int kPackTile = kBlock * kPackItems;   // 2048 rows per workgroup
constexpr int kPackSlots = 64, kPackSlotStride = 32, kPackSlot0 = 32;   // bidder minimum / maximum: slot s at minmax[kPackSlot0 + s * kPackSlotStride]
constexpr int kPackWords = kPackSlot0 + kPackSlots * kPackSlotStride;
__global__ __launch_bounds__(kBlock) void q11_pack_kernel(const int32_t *__restrict__ keys, const int64_t *__restrict__ dt, int64_t n,
                                                          const int64_t *__restrict__ epoch_off, int32_t n_epochs, int32_t t_bits,
                                                          uint32_t *__restrict__ pay, int32_t *minmax, uint64_t *t_ref_out,
                                                          uint32_t *unfit, int mode) {
    __shared__ int32_t s_red[2 * kWavesPerBlock];
    const uint64_t half = uint64_t(1) << (t_bits - 1), t_ref = (uint64_t)dt[0] - half, lim = uint64_t(1) << t_bits;
    if (blockIdx.x == 0 && threadIdx.x == 0) *t_ref_out = t_ref;
    // the workgroup's first row decides where its rows start looking (uniform: scalar loads); nearly every workgroup lies inside one
    // epoch and inside the relation: that case is sixteen-byte loads and stores with nothing to decide per row
    const int64_t rb = (int64_t)blockIdx.x * kPackTile;
    int32_t e0 = 0, hi = n_epochs;        // epoch_off[e0] <= rb < epoch_off[hi]
    while (hi - e0 > 1 && !(mode & 2)) {
        const int32_t mid = (e0 + hi) >> 1;
        if (epoch_off[mid] <= rb) e0 = mid;
        else hi = mid;
    }
    const int64_t next0 = (mode & 2) ? INT64_MAX : e0 + 1 < n_epochs ? epoch_off[e0 + 1] : INT64_MAX;
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
    bool bad = false;
    if (rb + kPackTile <= n && next0 >= rb + kPackTile && !(reinterpret_cast<uintptr_t>(dt) & 15)) {   // (block-uniform)
        const uint32_t e_word = (uint32_t)e0 << t_bits;
        int4 kk[2];
        int4 tt[2][2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int64_t r = rb + it * (kBlock * 4) + threadIdx.x * 4;
            kk[it] = *reinterpret_cast<const int4 *>(keys + r);       // (the sort reads the keys next: left in the caches)
            const int32_t *tp = reinterpret_cast<const int32_t *>(dt + r);
            tt[it][0] = stream_load4(tp);
            tt[it][1] = stream_load4(tp + 4);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int64_t r = rb + it * (kBlock * 4) + threadIdx.x * 4;
            const uint64_t d0 = (((uint64_t)(uint32_t)tt[it][0].y << 32) | (uint32_t)tt[it][0].x) - t_ref;
            const uint64_t d1 = (((uint64_t)(uint32_t)tt[it][0].w << 32) | (uint32_t)tt[it][0].z) - t_ref;
            const uint64_t d2 = (((uint64_t)(uint32_t)tt[it][1].y << 32) | (uint32_t)tt[it][1].x) - t_ref;
            const uint64_t d3 = (((uint64_t)(uint32_t)tt[it][1].w << 32) | (uint32_t)tt[it][1].z) - t_ref;
            bad |= (d0 | d1 | d2 | d3) >= lim;
            if (!(mode & 4)) *reinterpret_cast<uint4 *>(pay + r) = make_uint4(e_word | (uint32_t)d0, e_word | (uint32_t)d1, e_word | (uint32_t)d2, e_word | (uint32_t)d3);
            mn = min(mn, min(min(kk[it].x, kk[it].y), min(kk[it].z, kk[it].w)));
            mx = max(mx, max(max(kk[it].x, kk[it].y), max(kk[it].z, kk[it].w)));
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < kPackItems; ++i) {
            const int64_t r = rb + (int64_t)i * kBlock + threadIdx.x;
            if (r >= n) break;
            int32_t e = e0;
            if (r >= next0) {
                ++e;
                while (e + 1 < n_epochs && epoch_off[e + 1] <= r) ++e;
            }
            const uint64_t d = (uint64_t)dt[r] - t_ref;
            const int32_t k = keys[r];
            bad |= d >= lim;
            pay[r] = ((uint32_t)e << t_bits) | (uint32_t)d;
            mn = min(mn, k);
            mx = max(mx, k);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if (__ballot(bad) && lane_id() == 0) *unfit = 1u;
    if (lane_id() == 0) {
        s_red[threadIdx.x >> 6] = mn;
        s_red[kWavesPerBlock + (threadIdx.x >> 6)] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) {
            mn = min(mn, s_red[w]);
            mx = max(mx, s_red[kWavesPerBlock + w]);
        }
        // (1e5 workgroups on ONE pair of words took 1.28 ms of a 1.53 ms kernel -- accesses to one address are served one after the other,
        // ~13 ns each, whether atomic or a load that looks before it leaps: 64 pairs, a cache line apart, the host folds them)
        if (mn <= mx && !(mode & 1)) {
            int32_t *slot = minmax + kPackSlot0 + (blockIdx.x % kPackSlots) * kPackSlotStride;
            atomicMin(&slot[0], mn);
            atomicMax(&slot[1], mx);
        }
    }
}

__global__ void q11_pack_init_kernel(int32_t *minmax, uint32_t *unfit) {
    int32_t *slot = minmax + kPackSlot0 + threadIdx.x * kPackSlotStride;   // (kPackSlots threads)
    slot[0] = 0x7fffffff;
    slot[1] = (int32_t)0x80000000;
    if (threadIdx.x == 0) *unfit = 0u;
}

struct CutRows {  // the nine sorted rows around a thread's eight boundaries
    int32_t key[kCutItems + 1], ep[kCutItems + 1];
    int64_t ts[kCutItems + 1];
};

// boundaries j0 .. j0+7 need rows j0-1 .. j0+7, clamped into [0, n): the lane's own eight rows through 16-byte loads
// (j0 is a multiple of 8), row j0-1 from the lane below (lane 0 reads it)
// (kPacked: `ep_s` is the sorted payload column of q11_pack_kernel, `ts_s` the one-word t_ref: 8 bytes per row instead of 16)
template <bool kPacked>
__device__ __forceinline__ void load_cut_rows(const int32_t *__restrict__ keys, const int64_t *__restrict__ ts_s,
                                              const int32_t *__restrict__ ep_s, int64_t n, int64_t j0, const SessionParams &p, CutRows &c) {
    if (kPacked) {
        const uint32_t *__restrict__ pay = reinterpret_cast<const uint32_t *>(ep_s);
        const uint64_t t_ref = *reinterpret_cast<const uint64_t *>(ts_s);
        const uint32_t t_mask = (1u << p.t_bits) - 1u;
        int32_t k_prev = 0;
        uint32_t p_prev = 0;
        if (lane_id() == 0) {
            const int64_t r = j0 > 0 ? (j0 - 1 < n ? j0 - 1 : n - 1) : 0;
            k_prev = keys[r];
            p_prev = pay[r];
        }
        uint32_t w[kCutItems + 1];
        if (j0 + kCutItems <= n) {
            const int4 k0 = *reinterpret_cast<const int4 *>(keys + j0), k1 = *reinterpret_cast<const int4 *>(keys + j0 + 4);
            const uint4 p0 = *reinterpret_cast<const uint4 *>(pay + j0), p1 = *reinterpret_cast<const uint4 *>(pay + j0 + 4);
            c.key[1] = k0.x; c.key[2] = k0.y; c.key[3] = k0.z; c.key[4] = k0.w;
            c.key[5] = k1.x; c.key[6] = k1.y; c.key[7] = k1.z; c.key[8] = k1.w;
            w[1] = p0.x; w[2] = p0.y; w[3] = p0.z; w[4] = p0.w;
            w[5] = p1.x; w[6] = p1.y; w[7] = p1.z; w[8] = p1.w;
        } else {
#pragma unroll
            for (int i = 1; i <= kCutItems; ++i) {
                int64_t r = j0 - 1 + i;
                r = r >= n ? n - 1 : r;
                c.key[i] = keys[r];
                w[i] = pay[r];
            }
        }
        const int32_t pk = __shfl_up(c.key[kCutItems], 1, 64);
        const uint32_t pw = __shfl_up(w[kCutItems], 1, 64);
        const bool first = lane_id() == 0;
        c.key[0] = first ? k_prev : pk;
        w[0] = first ? p_prev : pw;
#pragma unroll
        for (int i = 0; i <= kCutItems; ++i) {
            c.ep[i] = (int32_t)(w[i] >> p.t_bits);
            c.ts[i] = (int64_t)(t_ref + (uint64_t)(w[i] & t_mask));
        }
        return;
    }
    // lane 0's row j0 - 1 is asked for FIRST, so that it is in flight together with the rows below instead of after the shuffles
    // (a second memory round trip per workgroup of a kernel that makes one pass over 2048 boundaries and leaves)
    int32_t k_prev = 0, e_prev = 0;
    int64_t t_prev = 0;
    if (lane_id() == 0) {
        const int64_t r = j0 > 0 ? (j0 - 1 < n ? j0 - 1 : n - 1) : 0;
        k_prev = keys[r];
        e_prev = ep_s[r];
        t_prev = ts_s[r];
    }
    if (j0 + kCutItems <= n) {
        const int4 k0 = *reinterpret_cast<const int4 *>(keys + j0), k1 = *reinterpret_cast<const int4 *>(keys + j0 + 4);
        const int4 e0 = *reinterpret_cast<const int4 *>(ep_s + j0), e1 = *reinterpret_cast<const int4 *>(ep_s + j0 + 4);
        c.key[1] = k0.x; c.key[2] = k0.y; c.key[3] = k0.z; c.key[4] = k0.w;
        c.key[5] = k1.x; c.key[6] = k1.y; c.key[7] = k1.z; c.key[8] = k1.w;
        c.ep[1] = e0.x; c.ep[2] = e0.y; c.ep[3] = e0.z; c.ep[4] = e0.w;
        c.ep[5] = e1.x; c.ep[6] = e1.y; c.ep[7] = e1.z; c.ep[8] = e1.w;
#pragma unroll
        for (int i = 0; i < kCutItems / 2; ++i) {
            const longlong2 t = *reinterpret_cast<const longlong2 *>(ts_s + j0 + 2 * i);
            c.ts[1 + 2 * i] = t.x;
            c.ts[2 + 2 * i] = t.y;
        }
    } else {
#pragma unroll
        for (int i = 1; i <= kCutItems; ++i) {
            int64_t r = j0 - 1 + i;
            r = r >= n ? n - 1 : r;
            c.key[i] = keys[r];
            c.ep[i] = ep_s[r];
            c.ts[i] = ts_s[r];
        }
    }
    // every lane of the wave executes the shuffles (j0 <= n for the whole wave or clamped rows above)
    const int32_t pk = __shfl_up(c.key[kCutItems], 1, 64), pe = __shfl_up(c.ep[kCutItems], 1, 64);
    const int64_t pt = __shfl_up(c.ts[kCutItems], 1, 64);
    const bool first = lane_id() == 0;
    c.key[0] = first ? k_prev : pk;
    c.ep[0] = first ? e_prev : pe;
    c.ts[0] = first ? t_prev : pt;
}

__device__ __forceinline__ uint32_t cut_mask(const CutRows &c, int64_t n, int64_t j0, const SessionParams &p) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < kCutItems; ++i) {
        const int64_t j = j0 + i;
        bool cut = false;
        if (j <= n)
            cut = j == 0 || j == n || c.key[i] != c.key[i + 1] || !same_session(c.ep[i], c.ts[i], c.ep[i + 1], c.ts[i + 1], p);
        m |= (uint32_t)cut << i;
    }
    return m;
}

template <bool kPacked>
__global__ __launch_bounds__(kBlock) void q11_cut_count_kernel(const int32_t *__restrict__ keys, const int64_t *__restrict__ ts_s,
                                                               const int32_t *__restrict__ ep_s, int64_t n, SessionParams p,
                                                               uint32_t *__restrict__ counts) {
    const int64_t j0 = (int64_t)blockIdx.x * kCutTile + (int64_t)threadIdx.x * kCutItems;
    CutRows c;
    load_cut_rows<kPacked>(keys, ts_s, ep_s, n, j0, p, c);
    const uint32_t incl = wave_incl_scan_u32((uint32_t)__popc(cut_mask(c, n, j0, p)));
    if (lane_id() == 63) counts[(size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)] = incl;
}

// Session s = sorted rows [cut_pos[s], cut_pos[s+1]).  close[s]: epoch in which it is closed, -1 = never.
template <bool kPacked>
__global__ __launch_bounds__(kBlock) void q11_cut_emit_kernel(const int32_t *__restrict__ keys, const int64_t *__restrict__ ts_s,
                                                              const int32_t *__restrict__ ep_s, int64_t n, SessionParams p,
                                                              const uint32_t *__restrict__ counts,
                                                              const uint64_t *__restrict__ tile_base, int32_t *__restrict__ cut_pos,
                                                              int32_t *__restrict__ s_bidder, int32_t *__restrict__ s_close,
                                                              unsigned long long *s_min, unsigned long long *s_max) {
    const int64_t j0 = (int64_t)blockIdx.x * kCutTile + (int64_t)threadIdx.x * kCutItems;
    CutRows c;
    load_cut_rows<kPacked>(keys, ts_s, ep_s, n, j0, p, c);
    const uint32_t m = cut_mask(c, n, j0, p);
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)blockIdx.x * kWavesPerBlock);
    const int wave = threadIdx.x >> 6;
    const uint32_t cnt = (uint32_t)__popc(m);
    uint64_t idx = tile_base[blockIdx.x] + (wave > 0 ? wc.x : 0u) + (wave > 1 ? wc.y : 0u) + (wave > 2 ? wc.z : 0u) +
                   wave_incl_scan_u32(cnt) - cnt;  // cuts before this thread's first boundary
    unsigned long long mn = ~0ull, mx = 0ull;
    // A long session (a hot bidder) covers whole waves and workgroups: their rows meet in registers / LDS first, so the
    // session's MIN / MAX see one atomic per workgroup instead of one per lane on the same address.
    __shared__ unsigned long long s_red[2 * kWavesPerBlock];
    const bool block_cut = __syncthreads_or(m != 0);
    if (!__ballot(m != 0)) {   // no cut in this wave: every row of it belongs to session idx - 1
#pragma unroll
        for (int i = 1; i <= kCutItems; ++i) {
            if (j0 + i - 1 < n) {
                const unsigned long long t = (unsigned long long)c.ts[i];
                mn = t < mn ? t : mn;
                mx = t > mx ? t : mx;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if (block_cut) {
            if (lane_id() == 0 && mn != ~0ull) {
                atomicMin(&s_min[idx - 1], mn);
                atomicMax(&s_max[idx - 1], mx);
            }
            return;
        }
        if (lane_id() == 0) {
            s_red[wave] = mn;
            s_red[kWavesPerBlock + wave] = mx;
        }
    }
    if (!block_cut) {          // (uniform) no cut in the workgroup
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) {
                mn = s_red[w] < mn ? s_red[w] : mn;
                mx = s_red[kWavesPerBlock + w] > mx ? s_red[kWavesPerBlock + w] : mx;
            }
            if (mn != ~0ull) {
                atomicMin(&s_min[idx - 1], mn);
                atomicMax(&s_max[idx - 1], mx);
            }
        }
        return;
    }
    // the piece of session (idx - 1) that this thread holds: rows j0 .. (first cut), and so on
    bool have = false;
#pragma unroll
    for (int i = 0; i < kCutItems; ++i) {
        const int64_t j = j0 + i;
        if (j > n) break;
        if ((m >> i) & 1u) {
            if (have) {  // rows before the cut belong to session idx - 1
                atomicMin(&s_min[idx - 1], mn);
                atomicMax(&s_max[idx - 1], mx);
                have = false;
                mn = ~0ull;
                mx = 0ull;
            }
            cut_pos[idx] = (int32_t)j;
            if (j < n) s_bidder[idx] = c.key[i + 1];
            if (j > 0) {  // the session that ends with row j-1
                const int64_t tc = close_clock(c.ep[i], c.ts[i], p);
                int64_t e;
                if (j < n && c.key[i] == c.key[i + 1]) e = tc < (int64_t)c.ep[i + 1] ? tc : (int64_t)c.ep[i + 1];
                else e = tc <= (int64_t)p.n_epochs - 1 ? tc : -1;
                s_close[idx - 1] = (int32_t)e;
            }
            ++idx;
        }
        if (j < n) {  // row j belongs to session idx - 1
            const unsigned long long t = (unsigned long long)c.ts[i + 1];
            mn = t < mn ? t : mn;
            mx = t > mx ? t : mx;
            have = true;
        }
    }
    if (have) {
        atomicMin(&s_min[idx - 1], mn);
        atomicMax(&s_max[idx - 1], mx);
    }
}

// Final row of every session + its sort key: the closing epoch, or n_epochs ("never": open at the end of the run, or
// folded into the session of the same bidder that closed in the same epoch just before it).
__global__ __launch_bounds__(kBlock) void q11_merge_kernel(const int32_t *__restrict__ cut_pos, const int32_t *__restrict__ s_bidder,
                                                           const int32_t *__restrict__ s_close,
                                                           const unsigned long long *__restrict__ s_min,
                                                           const unsigned long long *__restrict__ s_max, int64_t n_sessions,
                                                           int32_t n_epochs, int32_t *__restrict__ key,
                                                           uint64_t *__restrict__ f_count, int64_t *__restrict__ f_min,
                                                           int64_t *__restrict__ f_max) {
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_sessions) return;
    const int32_t b = s_bidder[s], e = s_close[s];
    const bool absorbed = s > 0 && e >= 0 && s_bidder[s - 1] == b && s_close[s - 1] == e;
    uint64_t cnt = (uint64_t)(cut_pos[s + 1] - cut_pos[s]);
    unsigned long long mn = s_min[s], mx = s_max[s];
    if (!absorbed && e >= 0 && s + 1 < n_sessions && s_bidder[s + 1] == b && s_close[s + 1] == e) {
        cnt += (uint64_t)(cut_pos[s + 2] - cut_pos[s + 1]);
        mn = min(mn, s_min[s + 1]);
        mx = max(mx, s_max[s + 1]);
    }
    key[s] = (e < 0 || absorbed) ? n_epochs : e;
    f_count[s] = cnt;
    f_min[s] = (int64_t)mn;
    f_max[s] = (int64_t)mx;
}

__global__ __launch_bounds__(kBlock) void q11_take_u64_kernel(const uint64_t *__restrict__ src, const int32_t *__restrict__ rows,
                                                              int64_t n, uint64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = src[rows[i]];
}

}  // namespace

extern "C" {

int flockgpu_q11_user_sessions(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const int64_t *epoch_row_offsets, int32_t n_epochs,
                               int32_t timeout_s, int64_t base_time_ms, flockgpu_q11_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!bid || !out || !epoch_row_offsets || n_epochs < 0 || timeout_s < 0 || bid->rows < 0)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q11: null / negative argument");
    for (int32_t t = 0; t < n_epochs; ++t)
        if (epoch_row_offsets[t + 1] < epoch_row_offsets[t]) return fail(ctx, FLOCKGPU_ERR_INVALID, "q11: epoch offsets decrease at epoch %d", t);
    const int64_t r0 = epoch_row_offsets[0], n = epoch_row_offsets[n_epochs] - r0;
    if (r0 < 0 || r0 + n > bid->rows) return fail(ctx, FLOCKGPU_ERR_INVALID, "q11: epoch offsets outside the relation");
    if (n > 0 && (!bid->bidder || !bid->b_date_time)) return fail(ctx, FLOCKGPU_ERR_INVALID, "q11: null bid column");
    if (n >= (int64_t(1) << 31) - 1) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q11: relations are limited to 2^31 rows per call");
    if (n > 0 && (reinterpret_cast<uintptr_t>(bid->bidder + r0) & 15))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q11: first bidder value of the run must be 16-byte aligned");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int64_t> &offs = ctx->host_i64["q11.epoch_out_offsets"];
    offs.assign((size_t)n_epochs + 1, 0);
    *out = flockgpu_q11_result{};
    out->epoch_out_offsets = offs.data();
    if (n == 0 || n_epochs == 0) return FLOCKGPU_OK;
    const int32_t *keys = bid->bidder + r0;
    const int64_t *dt = bid->b_date_time + r0;
    SessionParams p{base_time_ms / 1000, timeout_s, n_epochs, 0};

    // epoch offsets relative to the run, on the device
    int64_t *d_eoff = nullptr, *h_eoff = nullptr;
    FG_TRY(arena_get_t(ctx, "q11.epoch_off", (size_t)n_epochs + 1, &d_eoff));
    FG_TRY(pinned_get_t(ctx, "q11.epoch_off", (size_t)n_epochs + 1, &h_eoff));
    int32_t *d_mm = nullptr, *h_mm = nullptr;
    FG_TRY(arena_get_t(ctx, "q11.minmax", (size_t)kPackWords, &d_mm));     // [0..1] bidder min / max, [2] "a row does not fit the payload", [4..5] t_ref, slots
    FG_TRY(pinned_get_t(ctx, "q11.minmax", (size_t)kPackWords, &h_mm));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pinned staging may be in flight from the previous call
    for (int32_t t = 0; t <= n_epochs; ++t) h_eoff[t] = epoch_row_offsets[t] - r0;
    FG_HIP(ctx, hipMemcpyAsync(d_eoff, h_eoff, sizeof(int64_t) * ((size_t)n_epochs + 1), hipMemcpyHostToDevice, ctx->stream));

    // ---- the payload the sort carries: (epoch, time) in 32 bits when the run allows it, else the row number (gathered afterwards)
    int e_bits = 1;
    while (e_bits < 31 && ((uint32_t)(n_epochs - 1) >> e_bits)) ++e_bits;
    int t_bits = 32 - e_bits;
    if (t_bits < 12 || exp_env("FLOCKGPU_Q11_GATHER") != nullptr) t_bits = 0;   // (A/B knob of the experimental build)
    uint32_t *pay = nullptr;
    uint32_t *d_unfit = reinterpret_cast<uint32_t *>(d_mm + 2);
    uint64_t *d_tref = reinterpret_cast<uint64_t *>(d_mm + 4);
    if (t_bits) {
        FG_TRY(arena_get_t(ctx, "q11.payload", (size_t)n + 4, &pay));
        hipLaunchKernelGGL(q11_pack_init_kernel, dim3(1), dim3(kPackSlots), 0, ctx->stream, d_mm, d_unfit);
        FG_TRY(check_launch(ctx, "q11_pack_init_kernel"));
        {
            LaunchScope ls(ctx, "q11_pack_kernel");
            hipLaunchKernelGGL(q11_pack_kernel, dim3((unsigned)div_up(n, (int64_t)kPackTile)), dim3(kBlock), 0, ctx->stream, keys, dt, n,
                               d_eoff, n_epochs, t_bits, pay, d_mm, d_tref, d_unfit, exp_env("FLOCKGPU_Q11_PACK_MODE") ? atoi(exp_env("FLOCKGPU_Q11_PACK_MODE")) : 0);
        }
        FG_TRY(check_launch(ctx, "q11_pack_kernel"));
    } else {
        FG_TRY(key_min_max(ctx, keys, n, d_mm));
    }
    FG_HIP(ctx, hipMemcpyAsync(h_mm, d_mm, (t_bits ? (size_t)kPackWords : 3) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (t_bits) {
        h_mm[0] = 0x7fffffff;
        h_mm[1] = (int32_t)0x80000000;
        for (int sl = 0; sl < kPackSlots; ++sl) {
            h_mm[0] = std::min(h_mm[0], h_mm[kPackSlot0 + sl * kPackSlotStride]);
            h_mm[1] = std::max(h_mm[1], h_mm[kPackSlot0 + sl * kPackSlotStride + 1]);
        }
        if (h_mm[2]) t_bits = 0;   // a time outside the payload's span: this run takes the gather
    }
    p.t_bits = t_bits;

    // ---- group the bids by bidder
    int bits = 1;
    {
        const uint64_t span = (uint64_t)((int64_t)h_mm[1] - (int64_t)h_mm[0]);
        while (bits < 32 && (span >> bits)) ++bits;
    }
    int32_t *sk = nullptr;
    uint32_t *sv = nullptr;
    FG_TRY(radix_sort_pairs(ctx, "q11.rows", keys, t_bits ? pay : nullptr, n, h_mm[0], bits, &sk, &sv));

    const int64_t *ts_s = nullptr;
    const int32_t *ep_s = nullptr;
    if (t_bits) {
        ts_s = reinterpret_cast<const int64_t *>(d_tref);
        ep_s = reinterpret_cast<const int32_t *>(sv);
    } else {
        int64_t *ts_g = nullptr;
        int32_t *ep_g = nullptr;
        FG_TRY(arena_get_t(ctx, "q11.ts_sorted", (size_t)n, &ts_g));
        FG_TRY(arena_get_t(ctx, "q11.ep_sorted", (size_t)n, &ep_g));
        {
            LaunchScope ls(ctx, "q11_gather_kernel");
            hipLaunchKernelGGL(q11_gather_kernel, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, dt,
                               reinterpret_cast<const int32_t *>(sv), n, d_eoff, n_epochs, ts_g, ep_g);
        }
        FG_TRY(check_launch(ctx, "q11_gather_kernel"));
        ts_s = ts_g;
        ep_s = ep_g;
    }

    // ---- cuts: count -> scan -> emit over the n + 1 boundaries
    const int64_t tiles = div_up(n + 1, kCutTile);
    uint32_t *counts = nullptr;
    uint64_t *tile_base = nullptr, *h_total = nullptr;
    FG_TRY(arena_get_t(ctx, "q11.cut_counts", (size_t)tiles * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, "q11.cut_base", (size_t)tiles + 1, &tile_base));
    FG_TRY(pinned_get_t(ctx, "q11.cut_total", 1, &h_total));
    {
        LaunchScope ls(ctx, "q11_cut_count_kernel");
        hipLaunchKernelGGL(t_bits ? q11_cut_count_kernel<true> : q11_cut_count_kernel<false>, dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream, sk,
                           ts_s, ep_s, n, p, counts);
    }
    FG_TRY(check_launch(ctx, "q11_cut_count_kernel"));
    FG_TRY(launch_tile_scan(ctx, counts, (int32_t)tiles, tile_base, nullptr, 0, nullptr));
    FG_HIP(ctx, hipMemcpyAsync(h_total, tile_base + tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t n_sessions = (int64_t)*h_total - 1;  // cuts at 0 and n bracket the sessions
    if (n_sessions < 1) return fail(ctx, FLOCKGPU_ERR_HIP, "q11: %lld cuts over %lld rows", (long long)*h_total, (long long)n);

    int32_t *cut_pos = nullptr, *s_bidder = nullptr, *s_close = nullptr, *s_key = nullptr;
    unsigned long long *s_min = nullptr, *s_max = nullptr;
    uint64_t *f_count = nullptr;
    int64_t *f_min = nullptr, *f_max = nullptr;
    const size_t ns = (size_t)n_sessions;
    FG_TRY(arena_get_t(ctx, "q11.cut_pos", ns + 2, &cut_pos));
    FG_TRY(arena_get_t(ctx, "q11.s_bidder", ns + 2, &s_bidder));
    FG_TRY(arena_get_t(ctx, "q11.s_close", ns + 2, &s_close));
    FG_TRY(arena_get_t(ctx, "q11.s_key", ns + 4, &s_key));
    FG_TRY(arena_get_t(ctx, "q11.s_min", ns + 2, &s_min));
    FG_TRY(arena_get_t(ctx, "q11.s_max", ns + 2, &s_max));
    FG_TRY(arena_get_t(ctx, "q11.f_count", ns + 2, &f_count));
    FG_TRY(arena_get_t(ctx, "q11.f_min", ns + 2, &f_min));
    FG_TRY(arena_get_t(ctx, "q11.f_max", ns + 2, &f_max));
    FG_HIP(ctx, hipMemsetAsync(s_min, 0xFF, sizeof(unsigned long long) * (ns + 1), ctx->stream));
    FG_HIP(ctx, hipMemsetAsync(s_max, 0, sizeof(unsigned long long) * (ns + 1), ctx->stream));
    {
        LaunchScope ls(ctx, "q11_cut_emit_kernel");
        hipLaunchKernelGGL(t_bits ? q11_cut_emit_kernel<true> : q11_cut_emit_kernel<false>, dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream, sk,
                           ts_s, ep_s, n, p, counts,
                           tile_base, cut_pos, s_bidder, s_close, s_min, s_max);
    }
    FG_TRY(check_launch(ctx, "q11_cut_emit_kernel"));
    {
        LaunchScope ls(ctx, "q11_merge_kernel");
        hipLaunchKernelGGL(q11_merge_kernel, dim3((unsigned)div_up(n_sessions, kBlock)), dim3(kBlock), 0, ctx->stream, cut_pos, s_bidder,
                           s_close, s_min, s_max, n_sessions, n_epochs, s_key, f_count, f_min, f_max);
    }
    FG_TRY(check_launch(ctx, "q11_merge_kernel"));

    // ---- order by closing epoch
    int ebits = 1;
    while (ebits < 31 && ((uint32_t)n_epochs >> ebits)) ++ebits;
    int32_t *ok = nullptr;
    uint32_t *ov = nullptr;
    FG_TRY(radix_sort_pairs(ctx, "q11.sessions", s_key, nullptr, n_sessions, 0, ebits, &ok, &ov));
    int64_t *d_off = nullptr, *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, "q11.out_off", (size_t)n_epochs + 2, &d_off));
    FG_TRY(pinned_get_t(ctx, "q11.out_off", (size_t)n_epochs + 2, &h_off));
    FG_TRY(sorted_key_offsets(ctx, ok, n_sessions, n_epochs, d_off));
    FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * ((size_t)n_epochs + 1), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    offs.assign(h_off, h_off + n_epochs + 1);
    const int64_t m = offs[n_epochs];  // sessions with a closing epoch

    int32_t *o_bidder = nullptr;
    uint64_t *o_count = nullptr;
    int64_t *o_min = nullptr, *o_max = nullptr;
    FG_TRY(arena_get_t(ctx, "q11.out_bidder", (size_t)m + 1, &o_bidder));
    FG_TRY(arena_get_t(ctx, "q11.out_count", (size_t)m + 1, &o_count));
    FG_TRY(arena_get_t(ctx, "q11.out_start", (size_t)m + 1, &o_min));
    FG_TRY(arena_get_t(ctx, "q11.out_end", (size_t)m + 1, &o_max));
    if (m > 0) {
        const int32_t *order = reinterpret_cast<const int32_t *>(ov);
        FG_TRY(gather_i32(ctx, s_bidder, order, m, o_bidder));
        FG_TRY(gather_i64(ctx, f_min, order, m, o_min));
        FG_TRY(gather_i64(ctx, f_max, order, m, o_max));
        hipLaunchKernelGGL(q11_take_u64_kernel, dim3((unsigned)div_up(m, kBlock)), dim3(kBlock), 0, ctx->stream, f_count, order, m,
                           o_count);
        FG_TRY(check_launch(ctx, "q11_take_u64_kernel"));
    }
    out->bidder = o_bidder;
    out->bid_count = o_count;
    out->start_time = o_min;
    out->end_time = o_max;
    out->epoch_out_offsets = offs.data();
    out->rows = m;
    out->sessions_total = n_sessions;
    return FLOCKGPU_OK;
}

}  // extern "C"
