// NEXMark q5 "hot items" for gfx950: per hopping window
//   COUNT(*) GROUP BY auction  ->  MAX(num)  ->  rows with num = maxn (ties kept)
// (benchmarks/src/nexmark/query/q5.sql, q5_plan.fmt:1-13, playground/.../nexmark/q5.dag).
//
// HBM-bound integer work, no MFMA.  One pass over the `auction` column (4 B / bid):
//   range  : a sampling kernel (<1 % of the column) estimates every pane's key range (pane = gcd(window, hop)
//            seconds of rows); the host turns it into one direct-address counter array [base, base + range) per
//            PANE when the ranges are affordable ("dense" group-by: NEXMark ids are dense and time-ordered).
//            Keys outside a pane's estimate -- or every key when the ranges are not affordable -- go to a
//            per-WINDOW open-addressing hash table instead, so the result is exact for any input.
//   count  : rows are cut into 8192-row tiles that never straddle a pane.  A workgroup pre-aggregates its tile in
//            LDS: a direct-mapped histogram over [tile min, tile max] (ds_add_u32, no CAS).  Half of all bids hit
//            one auction id (event.rs:355-359): each wave keeps that id and its count in scalar registers
//            (ballot + s_bcnt1) and only sends the other keys to LDS, so the hot key costs no LDS conflicts.
//            The tile's distinct (key, count) pairs are added with ONE fire-and-forget global atomic each to the
//            pane's counters (pane sharing: a bid is read once and counted once although it belongs to
//            window/hop windows).  Adjacent tiles touch adjacent counters, so the atomics coalesce.
//            Ragged tiles and tiles whose keys spread wider than the histogram are queued for a small second
//            kernel with the general LDS-hash path.
//   max    : per window, count(key) = sum of its panes' counters + its hash table; maximum and group count.
//   select : rows whose count equals the window maximum.
#include <algorithm>

#include "sort.hpp"

using namespace flockgpu;

namespace {

constexpr int kQ5Iters = 8;
constexpr int kQ5Tile = kBlock * 4 * kQ5Iters;  // 8192 rows
constexpr int kHist = 4096;                     // direct-mapped LDS histogram bins (u32)
constexpr int kHistPad = 64;                    // one scratch bin per lane for "not mine" adds
constexpr int kSlots = kHist / 2;               // the same LDS viewed as packed {key:32,count:32} hash slots
constexpr int kSlotBits = 11;
static_assert(kSlots == (1 << kSlotBits), "slot bits");
constexpr int kLdsMaxProbe = 24;
constexpr uint32_t kFib = 0x9E3779B1u;
// The count kernel's LDS phase: 1 (shipped) = lanes holding the wave's hot key issue no LDS atomic at all (exec-masked); 0 = they add to
// a private scratch bin instead (branch-free, rounds 1-3); 2 = 1 + four lane-indexed replicas of every bin while the tile's span leaves
// room.  A/B on one box, alternating (tools/gpu_ab_libs.sh, round 4): 0.741 / 0.728 / 0.730 and 0.753 / 0.732 / 0.728 ms per 1e9 bids --
// the scratch-bin adds cost 2 %, and replicas buy nothing on top: with half the lanes masked off, the ~32 that remain spread over the
// ~110 auctions in flight collide rarely enough (the counters' "two thirds of the LDS cycles are bank conflicts" was mostly the hot
// lanes' 64 scratch adds folding twice over the 32 banks).  Round 6, after tools/micro/stream_read.hip showed the phase costing 0.08 ms on top
// of the stream: 3 = variant 1 with a row's instructions written out in assembly (2 VALU + 4 SALU + 1 DS per row instead of the compiler's
// 5 + 4 + 1): 0.694 against 0.693 ms, nothing; 5 = the same without the exec mask, the hot lanes adding to scratch bins (3 + 2 + 1): 0.711
// against 0.688; hot rows counted per lane (3 + 2 + 1, no ballot): 1.30 ms, the wave-wide sum at every change of candidate costs more than
// the ballots.  The phase is bound by the LDS adds that are executed, lane by lane -- not by the instructions around them.  Variants other than 1
// exist in experimental builds only.
#if defined(FLOCKGPU_EXPERIMENTAL) && defined(FLOCKGPU_AB_Q5_VARIANT)
constexpr int kQ5Variant = FLOCKGPU_AB_Q5_VARIANT;
#else
constexpr int kQ5Variant = 1;
#endif
constexpr int kQ5WaveForm = 0;                  // count kernel form of the shipped build: 0 = one workgroup per tile, 1 / 2 / 4 = one WAVE per tile, that many waves per workgroup
constexpr int kQ5WavesPerCu = 20;               // persistent wave form: waves per CU in the grid (8 KB of LDS each)
// 16-bit counters (round 6): two keys share a 32-bit word of the counter arena (key - base even: low half).  What bounds the q5 step is the
// bytes it moves, reads and writes together (profiles/r06/q5_variants_ab.md): half-width counters halve the zeroes stored per call, the max /
// select sweeps and the lines the flush atomics touch.  A (pane, key) count above 65535 carries into its neighbour or out of the word --
// either way the pane's counter SUM comes out below its row count, which the max pass adds up anyway: q5_finish_kernel compares, and a
// mismatch makes the host repeat the call with 32-bit counters and keep them for this ctx (NEXMark's hottest auction draws ~770 bids).
constexpr bool kQ5Counters16 = true;
constexpr int kHotMin = 16;                     // a candidate seen in fewer lanes than this is not "hot"
constexpr int kMaxWinPanes = 8;                 // windows of more panes use the hash tables only
constexpr uint32_t kWideTile = 0x40000000u;     // slow-list tag: declined for its key spread (not for being ragged)

struct PaneDesc {
    int64_t base;      // first key of the pane's direct-address range (multiple of 8)
    uint64_t cnt_off;  // offset of the pane's counters in the counter arena (in counters, multiple of 8)
    uint32_t range;    // number of counters (multiple of 8; 0: every key of this pane goes to the hash tables)
    uint32_t pad;
};

struct WinDesc {
    int64_t base;      // union of the window's pane ranges (multiple of 4)
    uint32_t range;    // multiple of 4; 0 when the window has no direct-address panes
    int32_t pane_lo, pane_hi;
    uint32_t pad;
};

// 0: no room within kLdsMaxProbe slots, 1: added to the key's slot, 2: claimed a new slot
__device__ __forceinline__ int lds_hash_insert(uint64_t *tab, uint32_t key, uint32_t c) {
    uint32_t s = (key * kFib) >> (32 - kSlotBits);
    const uint64_t mine = ((uint64_t)key << 32) | c;
#pragma unroll 1
    for (int probe = 0; probe < kLdsMaxProbe; ++probe) {
        uint64_t cur = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == 0) {
            cur = atomicCAS(reinterpret_cast<unsigned long long *>(&tab[s]), 0ull, (unsigned long long)mine);
            if (cur == 0) return 2;
        }
        if ((uint32_t)(cur >> 32) == key) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&tab[s]), (unsigned long long)c);
            return 1;
        }
        s = (s + 1) & (kSlots - 1);
    }
    return 0;
}

// Window hash table (packed {key:32,count:32}, 0 = empty; counts >= 1 so a live slot is never 0).
__device__ __forceinline__ void table_add(uint64_t *tab, uint32_t cap, uint32_t key, uint32_t c, uint32_t *used, uint32_t *err) {
    uint32_t s = (uint32_t)(((uint64_t)(key * kFib) * cap) >> 32);
    const uint64_t mine = ((uint64_t)key << 32) | c;
#pragma unroll 1
    for (uint32_t probe = 0, lim = cap < 2048u ? cap : 2048u; probe < lim; ++probe) {
        uint64_t cur = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            uint64_t expected = 0;
            if (__hip_atomic_compare_exchange_strong(&tab[s], &expected, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                // `used` only says "this window has table entries": a counter here is one same-address atomic per new
                // group (1e8 distinct keys: 150 ms of serialised atomics); a cached read + rare store is free
                if (*used == 0) *used = 1u;
                return;
            }
            cur = expected;
        }
        if ((uint32_t)(cur >> 32) == key) {
            __hip_atomic_fetch_add(&tab[s], (uint64_t)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    atomicOr(err, 1u);
}

__device__ __forceinline__ uint32_t table_find(const uint64_t *tab, uint32_t cap, uint32_t key) {
    uint32_t s = (uint32_t)(((uint64_t)(key * kFib) * cap) >> 32);
#pragma unroll 1
    for (uint32_t probe = 0, lim = cap < 2048u ? cap : 2048u; probe < lim; ++probe) {
        const uint64_t cur = tab[s];
        if (cur == 0) return 0;
        if ((uint32_t)(cur >> 32) == key) return (uint32_t)cur;
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    return 0;
}

struct FlushArgs {
    PaneDesc pane;             // the tile's pane
    int32_t wp0, wp1;          // CSR range of the windows that contain the pane
    const int32_t *pane_win_idx;
    uint32_t *counters;
    uint64_t *tables;
    uint32_t cap;
    uint32_t *tab_used;        // per window: non-zero once its hash table holds an entry
    uint32_t *err;
    bool c16 = false;                        // 16-bit counters, two per word of `counters`
    unsigned long long *tab_rows = nullptr;  // c16: this pane's rows that went to the straggler tables instead of its counters (the sum check leaves them out)
};

// Adds one aggregated (key, count) pair of a tile: one atomic on the pane's counters, or -- for a key outside the
// pane's estimated range -- one hash-table update per window that contains the pane.
__device__ __forceinline__ void emit_pair(int32_t key, uint32_t c, const FlushArgs &f) {
    const uint64_t idx = (uint64_t)((int64_t)key - f.pane.base);
    if (idx < (uint64_t)f.pane.range) {
        if (f.c16) __hip_atomic_fetch_add(&f.counters[(f.pane.cnt_off + idx) >> 1], c << (16u * ((uint32_t)idx & 1u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(&f.counters[f.pane.cnt_off + idx], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (f.tab_rows) atomicAdd(f.tab_rows, (unsigned long long)c);
    for (int wi = f.wp0; wi < f.wp1; ++wi) {
        const int32_t w = f.pane_win_idx[wi];
        table_add(f.tables + (size_t)w * f.cap, f.cap, (uint32_t)key, c, &f.tab_used[w], f.err);
    }
}

// 16-bit counters: the counts of an EVEN key and its successor as ONE atomic on their shared word (bases and ranges are multiples of 8, so the
// two are inside or outside a pane's range together)
__device__ __forceinline__ void emit_pair16(int32_t even_key, uint32_t c0, uint32_t c1, const FlushArgs &f) {
    const uint64_t idx = (uint64_t)((int64_t)even_key - f.pane.base);
    if (idx < (uint64_t)f.pane.range) {
        __hip_atomic_fetch_add(&f.counters[(f.pane.cnt_off + idx) >> 1], c0 | (c1 << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (c0) emit_pair(even_key, c0, f);
    if (c1) emit_pair(even_key + 1, c1, f);
}

// ---- pane key-range estimate: kRangeBlocks x 256 lanes x 4 strided 16-byte samples per pane (<1 % of a pane) ------
constexpr int kRangeBlocks = 2;   // (8: 0.043 ms per 1e9 bids -- every 16-byte sample costs a whole line; the 6 % margins of q5_pane_layout absorb the coarser estimate)
// Every (pane, block) writes its own sampled extrema (no initialisation, no atomics): rng[(pane * kRangeBlocks + block) * 2 + {0, 1}].
__global__ __launch_bounds__(kBlock) void q5_range_kernel(const int32_t *__restrict__ auction, int64_t n_rows,
                                                          const int64_t *__restrict__ seg_off, int32_t *__restrict__ rng) {
    __shared__ int32_t s_red[8];
    const int32_t pane = blockIdx.y;
    const int64_t sb = seg_off[2 * pane], se = seg_off[2 * pane + 1];
    const int64_t groups = (se - (sb & ~int64_t(3)) + 3) >> 2;  // aligned 4-row groups covering the pane
    const int64_t n_samples = (int64_t)kRangeBlocks * kBlock * 4;
    const int64_t stride = groups > n_samples ? groups / n_samples : 1;
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t g = ((int64_t)(i * kRangeBlocks + blockIdx.x) * kBlock + threadIdx.x) * stride;
        const int64_t r0 = (sb & ~int64_t(3)) + g * 4;
        if (g < groups && r0 + 4 <= n_rows) {
            const int4 t = *reinterpret_cast<const int4 *>(auction + r0);
            const int32_t k[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r0 + j >= sb && r0 + j < se) {
                    mn = min(mn, k[j]);
                    mx = max(mx, k[j]);
                }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if (lane_id() == 0) {
        s_red[threadIdx.x >> 6] = mn;
        s_red[4 + (threadIdx.x >> 6)] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t *o = rng + ((size_t)pane * kRangeBlocks + blockIdx.x) * 2;
        o[0] = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
        o[1] = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    }
}

// The layout rules, shared by the host (first call, Partial stage, fallback) and the device (speculated calls).
__host__ __device__ inline bool q5_pane_layout(int64_t lo, int64_t hi, int64_t *base, int64_t *range) {
    const int64_t span = hi - lo, margin = span / 16 > 4096 ? span / 16 : 4096;
    *base = (lo - margin) & ~int64_t(7);   // (multiples of 8: an aligned 16-byte group of counters -- four 32-bit, eight 16-bit -- is inside a pane's range as a whole or not at all)
    *range = ((hi + margin + 1 - *base) + 7) & ~int64_t(7);
    return *range < (int64_t(1) << 31);
}

// The direct-address layout decided ON THE DEVICE from the sampled pane ranges, so that the host does not wait for them in the
// middle of the call: PaneDesc per pane (base, counter offset, range), WinDesc per window (union range of its panes).
// info[0] = counters in use, info[1] = window-range total, info[2] = 1 when the layout is dense, affordable and fits the
// `capacity` counters the host allocated from the previous call's size; 0 declines the call (every kernel behind returns at
// once; the host sees it at its single synchronisation and repeats the call the slow way).
// (Round 6: not a launch of its own any more -- workgroup 0 of q5_clear_kernel runs it, the other workgroups of that launch zero what they
// zero without looking at its verdict: one 7 us launch less per call.)
struct LayoutArgs {
    const int32_t *rng;   // null: the layout was made on the host
    const int32_t *pane_win_ptr;
    const int64_t *seg_off;
    int32_t n_panes, n_win;
    uint64_t capacity, budget_bytes;
    PaneDesc *panes;
    WinDesc *wins;
    uint64_t *info;
};
__device__ __forceinline__ void q5_layout_block(const int32_t *__restrict__ rng, const int32_t *__restrict__ pane_win_ptr,
                                                const int64_t *__restrict__ seg_off, int32_t n_panes, int32_t n_win,
                                                uint64_t capacity, uint64_t budget_bytes, PaneDesc *__restrict__ panes,
                                                WinDesc *__restrict__ wins, uint64_t *__restrict__ info) {
    __shared__ uint64_t s_wave[kWavesPerBlock];
    __shared__ uint64_t s_carry;
    __shared__ int s_ok;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        s_carry = 0;
        s_ok = 1;
    }
    __syncthreads();
    for (int32_t p0 = 0; p0 < n_panes; p0 += kBlock) {
        const int32_t p = p0 + (int32_t)threadIdx.x;
        int64_t base = 0, range = 0;
        if (p < n_panes && pane_win_ptr[p + 1] != pane_win_ptr[p] && seg_off[2 * p + 1] > seg_off[2 * p]) {
            int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
            for (int b = 0; b < kRangeBlocks; ++b) {
                mn = min(mn, rng[((size_t)p * kRangeBlocks + b) * 2]);
                mx = max(mx, rng[((size_t)p * kRangeBlocks + b) * 2 + 1]);
            }
            if (mn <= mx && !q5_pane_layout(mn, mx, &base, &range)) {
                s_ok = 0;
                range = 0;
            }
        }
        const uint64_t incl = wave_incl_scan_u64((uint64_t)range);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint64_t off = s_carry + incl - (uint64_t)range;
        for (int v = 0; v < wave; ++v) off += s_wave[v];
        if (p < n_panes) panes[p] = PaneDesc{base, off, (uint32_t)range, 0};
        __syncthreads();
        if (threadIdx.x == kBlock - 1) s_carry = off + (uint64_t)range;
        __syncthreads();
    }
    const uint64_t cnt_total = s_carry;
    __syncthreads();  // panes[] written above are read below (same workgroup: visible after the barrier)
    uint64_t scan = 0;
    for (int32_t w = threadIdx.x; w < n_win; w += kBlock) {
        WinDesc d = wins[w];
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (int32_t p = d.pane_lo; p < d.pane_hi; ++p) {
            const PaneDesc pd = panes[p];
            if (pd.range) {
                lo = lo < pd.base ? lo : pd.base;
                hi = hi > pd.base + (int64_t)pd.range ? hi : pd.base + (int64_t)pd.range;
            }
        }
        d.base = 0;
        d.range = 0;
        if (lo < hi) {
            if (hi - lo >= (int64_t(1) << 31)) s_ok = 0;
            else {
                d.base = lo;
                d.range = (uint32_t)(hi - lo);
                scan += (uint64_t)(hi - lo);
            }
        }
        wins[w] = d;
    }
    scan = wave_sum_u64(scan);
    if (lane == 0) s_wave[wave] = scan;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t scan_total = 0;
        for (int v = 0; v < kWavesPerBlock; ++v) scan_total += s_wave[v];
        const bool ok = s_ok && cnt_total <= capacity && cnt_total * 4 <= budget_bytes && scan_total * 4 <= 2 * budget_bytes;
        info[0] = cnt_total;
        info[1] = scan_total;
        info[2] = ok ? 1 : 0;
    }
}

// One launch instead of five memsets: the counters in use (their number from the device when the layout was made there),
// the window tables, the scalars, the slow list head and the per-workgroup maxima.
// cnt_from: the words below it are known to be zero already (the previous call cleaned up after itself, see q5_run).
__global__ __launch_bounds__(kBlock) void q5_clear_kernel(uint32_t *__restrict__ counters, uint64_t cnt_host,
                                                          uint64_t cnt_from, uint64_t *__restrict__ tables, uint64_t table_words,
                                                          uint64_t *__restrict__ meta, uint64_t meta_words, int32_t *__restrict__ slow_list,
                                                          uint32_t *__restrict__ block_max, uint64_t block_max_words, int plain_stores, int c16, LayoutArgs lay) {
    if (lay.rng && blockIdx.x == 0) q5_layout_block(lay.rng, lay.pane_win_ptr, lay.seg_off, lay.n_panes, lay.n_win, lay.capacity, lay.budget_bytes, lay.panes, lay.wins, lay.info);
    uint64_t cnt = cnt_host;                // counters to zero (a speculating call: all the arena holds -- the layout is being made next door) ...
    if (c16) cnt = (cnt + 1) / 2;           // ... and the 32-bit words that hold them (cnt_from is in words, too)
    const uint64_t i0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x, stride = (uint64_t)gridDim.x * kBlock;
    const uint4 z = make_uint4(0, 0, 0, 0);
    // (non-temporal: the zeroes go to HBM as they are written instead of lingering as dirty lines whose write-back lands on the kernel that
    // runs next -- measured: +0.04 ms on whichever kernel followed the clear)
    if (plain_stores) {   // (FLOCKGPU_Q5_PLAIN_CLEAR: the round-1 behaviour, kept for the counter-level A/B in profiles/r02/q5_clear_policy.txt)
        for (uint64_t i = i0 + cnt_from / 4; i * 4 < cnt + 3; i += stride) reinterpret_cast<uint4 *>(counters)[i] = z;
    } else
    for (uint64_t i = i0 + cnt_from / 4; i * 4 < cnt + 3; i += stride) {
        uint32_t *p = counters + i * 4;
        __builtin_nontemporal_store(0u, p);
        __builtin_nontemporal_store(0u, p + 1);
        __builtin_nontemporal_store(0u, p + 2);
        __builtin_nontemporal_store(0u, p + 3);
    }
    for (uint64_t i = i0; i * 2 < table_words; i += stride) {
        if (i * 2 + 1 < table_words) reinterpret_cast<uint4 *>(tables)[i] = z;
        else tables[i * 2] = 0;
    }
    for (uint64_t i = i0; i < meta_words; i += stride) meta[i] = 0;
    for (uint64_t i = i0; i < block_max_words; i += stride) block_max[i] = 0;
    if (i0 == 0) slow_list[0] = 0;
}

// ---- count: fast kernel (full tiles whose keys fit the LDS histogram) -------------------------------------------
// kWeighted: every row carries a count (the FinalPartitioned side of q5.dag: rows are the partial groups another
// partition sent); keys are then (nearly) distinct inside a tile, so no hot-key bookkeeping.
template <bool kWeighted>
__device__ __forceinline__ void q5_count_tile(const int32_t *__restrict__ auction, const uint32_t *__restrict__ weight, const SegTiles &st,
                                              const PaneDesc *__restrict__ panes, const int32_t *__restrict__ pane_win_ptr,
                                              const int32_t *__restrict__ pane_win_idx, uint32_t *counters, uint64_t *tables, uint32_t cap,
                                              uint32_t *tab_used, uint32_t *err, int32_t *slow_list, unsigned long long *pane_wsum, const int32_t tile,
                                              uint32_t *hist, int32_t *s_red, unsigned long long *s_w, const bool c16, unsigned long long *tab_rows) {
    const TileRange tr = locate_tile(st, tile, kQ5Tile);
    FlushArgs f;
    f.c16 = c16;
    f.tab_rows = tab_rows ? tab_rows + tr.seg : nullptr;
    f.wp0 = pane_win_ptr[tr.seg];
    f.wp1 = pane_win_ptr[tr.seg + 1];
    if (f.wp0 == f.wp1) return;  // pane belongs to no (full) window
    f.pane = panes[tr.seg];
    f.pane_win_idx = pane_win_idx;
    f.counters = counters;
    f.tables = tables;
    f.cap = cap;
    f.tab_used = tab_used;
    f.err = err;
    const bool full_tile = tr.lo == tr.tile_begin && tr.hi == tr.tile_begin + kQ5Tile;
    if (kWeighted) {
        // Rows are the partial groups other partitions sent: per source they arrive as the source's counters in ascending key
        // order, i.e. an 8192-row tile holds ~8192 DISTINCT keys over a range wider than the LDS histogram.  Nothing to
        // aggregate in LDS: every row is one fire-and-forget atomic on the pane's counters, consecutive lanes on consecutive
        // counters (0.34 ms of LDS hashing in q5_count_slow_kernel before).  Ragged tiles take the same path, row-guarded;
        // the per-pane weight totals of the 2^32 guard ride along (a pass of their own before: 0.085 ms).
        unsigned long long wsum = 0;
        // lane l takes row  chunk * 256 + l : a wave's 64 atomics of one instruction land on 64 consecutive counters (two cache lines).
        // With the 16-byte loads of the unweighted path (lane l = rows 4l .. 4l + 3) they would stride 16 bytes over eight lines: 0.6 ms
        // instead of 0.1 ms for 6.5e7 groups.
        constexpr int kChunks = kQ5Tile / kBlock, kBatch = 8;
#pragma unroll 1
        for (int c0 = 0; c0 < kChunks; c0 += kBatch) {
            int32_t kk[kBatch];
            uint32_t ww[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                const int64_t r = tr.tile_begin + (int64_t)(c0 + b) * kBlock + threadIdx.x;
                const bool in = full_tile || (r >= tr.lo && r < tr.hi);
                kk[b] = in ? __builtin_nontemporal_load(&auction[r]) : 0;   // (pairs are read once)
                ww[b] = in ? __builtin_nontemporal_load(&weight[r]) : 0u;
            }
#pragma unroll
            for (int b = 0; b < kBatch; ++b)
                if (ww[b]) {   // (a zero count changes nothing; rows outside the pane carry zero)
                    emit_pair(kk[b], ww[b], f);
                    wsum += ww[b];
                }
        }
        wsum = wave_sum_u64(wsum);
        if (lane_id() == 0) s_w[threadIdx.x >> 6] = wsum;
        __syncthreads();
        if (threadIdx.x == 0 && pane_wsum) {
            const unsigned long long tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
            if (tot) atomicAdd(&pane_wsum[tr.seg], tot);
        }
        return;
    }
    {
        uint4 *z = reinterpret_cast<uint4 *>(hist);
        for (int s = threadIdx.x; s < (kHist + kHistPad) / 4; s += kBlock) z[s] = make_uint4(0, 0, 0, 0);
    }
    // A ragged tile (the first / last of a pane: ~430 of 122 K at 1e9 bids) runs the same code: the positions outside the pane hold a
    // COPY of the pane's first key in this tile, and that key's bin gives the copies back before the flush (it occurs at least once for
    // real, so the bin stays positive).  They went through the general LDS-hash kernel before: 0.025 ms per call for 0.35 % of the rows.
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    int32_t k[kQ5Iters][4];
    const int32_t key0 = full_tile ? 0 : auction[tr.lo];
    if (full_tile) {
#pragma unroll
        for (int it = 0; it < kQ5Iters; ++it) {
            const int64_t r0 = tr.tile_begin + it * (kBlock * 4) + threadIdx.x * 4;
            const int4 t = stream_load4(auction + r0);
            k[it][0] = t.x; k[it][1] = t.y; k[it][2] = t.z; k[it][3] = t.w;
        }
    } else {
#pragma unroll
        for (int it = 0; it < kQ5Iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = tr.tile_begin + it * (kBlock * 4) + threadIdx.x * 4 + j;
                k[it][j] = (r >= tr.lo && r < tr.hi) ? auction[r] : key0;
            }
    }
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
#pragma unroll
    for (int it = 0; it < kQ5Iters; ++it) {
        mn = min(mn, min(min(k[it][0], k[it][1]), min(k[it][2], k[it][3])));
        mx = max(mx, max(max(k[it][0], k[it][1]), max(k[it][2], k[it][3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if (lane == 0) {
        s_red[wave] = mn;
        s_red[4 + wave] = mx;
    }
    __syncthreads();  // also orders the zeroing of `hist`
    mn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    mx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    const uint32_t span = (uint32_t)mx - (uint32_t)mn;
    if (span >= (uint32_t)kHist) {  // keys spread wider than the histogram: general path in q5_count_slow_kernel
        if (threadIdx.x == 0) slow_list[1 + atomicAdd(&slow_list[0], 1)] = (int32_t)((uint32_t)tile | kWideTile);
        return;
    }
    // hot key of this wave, kept in scalar registers across iterations
    int32_t hot = __builtin_amdgcn_readfirstlane(k[0][0]);
    uint32_t hot_cnt = 0;
    const bool rep = kQ5Variant == 2 && span * 4 + 3 < (uint32_t)kHist;   // (block-uniform) room for four replicas of every bin
    const uint32_t rmul = rep ? 4u : 1u, radd = rep ? (uint32_t)(lane & 3) : 0u;
    if (kQ5Variant == 3 || kQ5Variant == 5) {
        // Variant 1 with the row's instructions written out: the hot key's rows are counted as the complement of the rows that go to the
        // histogram (one population count of the SAME mask that becomes the exec mask), a row's bin is one shift-add from its key, and the
        // constant 1 stays in a register: 2 VALU + 4 SALU + 1 DS instruction per row where the compiler's form of variant 1 issues 5 + 4 + 1.
        // (A wave64 instruction holds its SIMD16 for four cycles and the CU's scalar unit serves a SIMD every fourth cycle: at eight waves per
        // SIMD this phase is bound by its instruction count, and while a workgroup is in it, it has no loads in flight -- tools/micro/stream_read.hip.
        // Counting the cold rows per LANE instead -- 3 VALU + 2 SALU -- was run, too: the wave-wide sum it needs whenever the candidate changes
        // costs more than it saves, the candidate changes in most iterations of NEXMark's bids: 1.30 ms against 0.71.)
        uint32_t cold = 0;                         // (wave-uniform) rows since the last switch of candidates that were NOT the candidate's
        uint32_t since = 0;                        // (wave-uniform) rows since that switch
        uint32_t one = 1;
        asm volatile("" : "+v"(one));              // (kept in a register: the compiler re-made the constant in front of every LDS add)
        // the LDS byte address of bin 0 minus four times the tile's minimum: a key's bin is at (key << 2) + this
        const uint32_t hist_at_mn = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)hist - ((uint32_t)mn << 2));
#pragma unroll
        for (int it = 0; it < kQ5Iters; ++it) {
            const uint64_t b0 = __ballot(k[it][0] == hot);
            if ((int)__builtin_popcountll(b0) < kHotMin) {
                // the candidate went cold: park its count, then try this iteration's first two distinct keys
                if (since != cold) {
                    if (lane == 0) atomicAdd(&hist[(uint32_t)hot - (uint32_t)mn], since - cold);
                }
                cold = 0;
                since = 0;
                const int32_t c1 = __builtin_amdgcn_readfirstlane(k[it][0]);
                const uint64_t m1 = __ballot(k[it][0] == c1);
                hot = c1;
                if ((int)__builtin_popcountll(m1) < kHotMin && ~m1) {
                    const int l2 = __ffsll((unsigned long long)~m1) - 1;
                    const int32_t c2 = __builtin_amdgcn_readlane(k[it][0], l2);
                    const uint64_t m2 = __ballot(k[it][0] == c2);
                    if (__builtin_popcountll(m2) > __builtin_popcountll(m1)) hot = c2;
                }
            }
            since += 256;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // if (k != hot) hist[k - mn] += 1, and cold += the number of such lanes
                uint32_t bin, n_cold;
                unsigned long long saved;
                if (kQ5Variant == 5) {   // (A/B builds: no exec mask -- the hot key's lanes add to a scratch bin of their own)
                    const uint32_t scratch = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)hist + (uint32_t)(kHist + lane) * 4u;
                    asm volatile("v_cmp_ne_u32 vcc, %[hot], %[key]\n\t"
                                 "s_bcnt1_i32_b64 %[n_cold], vcc\n\t"
                                 "v_lshl_add_u32 %[bin], %[key], 2, %[base]\n\t"
                                 "v_cndmask_b32 %[bin], %[scratch], %[bin], vcc\n\t"
                                 "ds_add_u32 %[bin], %[one]"
                                 : [bin] "=&v"(bin), [n_cold] "=&s"(n_cold)
                                 : [hot] "s"(hot), [key] "v"(k[it][j]), [base] "s"(hist_at_mn), [one] "v"(one), [scratch] "v"(scratch)
                                 : "vcc", "scc", "memory");
                    cold += n_cold;
                    continue;
                }
                asm volatile("v_cmp_ne_u32 vcc, %[hot], %[key]\n\t"
                             "s_bcnt1_i32_b64 %[n_cold], vcc\n\t"
                             "s_and_saveexec_b64 %[saved], vcc\n\t"
                             "v_lshl_add_u32 %[bin], %[key], 2, %[base]\n\t"
                             "ds_add_u32 %[bin], %[one]\n\t"
                             "s_mov_b64 exec, %[saved]"
                             : [bin] "=&v"(bin), [saved] "=&s"(saved), [n_cold] "=&s"(n_cold)
                             : [hot] "s"(hot), [key] "v"(k[it][j]), [base] "s"(hist_at_mn), [one] "v"(one)
                             : "vcc", "scc", "memory");
                cold += n_cold;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the LDS adds above are not the compiler's to count)
        if (since != cold && lane == 0) atomicAdd(&hist[(uint32_t)hot - (uint32_t)mn], since - cold);
    } else {
#pragma unroll
    for (int it = 0; it < kQ5Iters; ++it) {
        uint64_t b0 = __ballot(k[it][0] == hot);
        if (__popcll((unsigned long long)b0) < kHotMin) {
            // the candidate went cold: park its count, then try this iteration's first two distinct keys
            if (hot_cnt) {
                if (lane == 0) atomicAdd(&hist[((uint32_t)hot - (uint32_t)mn) * rmul], hot_cnt);
                hot_cnt = 0;
            }
            const int32_t c1 = __builtin_amdgcn_readfirstlane(k[it][0]);
            const uint64_t m1 = __ballot(k[it][0] == c1);
            hot = c1;
            b0 = m1;
            if (__popcll((unsigned long long)m1) < kHotMin && ~m1) {
                const int l2 = __ffsll((unsigned long long)~m1) - 1;
                const int32_t c2 = __builtin_amdgcn_readlane(k[it][0], l2);
                const uint64_t m2 = __ballot(k[it][0] == c2);
                if (__popcll((unsigned long long)m2) > __popcll((unsigned long long)m1)) {
                    hot = c2;
                    b0 = m2;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool is_hot = k[it][j] == hot;
            const uint64_t b = (j == 0) ? b0 : __ballot(is_hot);
            hot_cnt += (uint32_t)__popcll((unsigned long long)b);
            if (kQ5Variant == 0) {
                // branch-free: lanes holding the hot key hit their private scratch bin instead
                const uint32_t bin = is_hot ? (uint32_t)(kHist + lane) : (uint32_t)k[it][j] - (uint32_t)mn;
                atomicAdd(&hist[bin], 1u);
            } else if (!is_hot) {
                atomicAdd(&hist[((uint32_t)k[it][j] - (uint32_t)mn) * rmul + radd], 1u);
            }
        }
    }
    if (hot_cnt && lane == 0) atomicAdd(&hist[((uint32_t)hot - (uint32_t)mn) * rmul], hot_cnt);
    }
    __syncthreads();
#if defined(FLOCKGPU_EXPERIMENTAL) && defined(FLOCKGPU_AB_Q5_ABLATE)   // (ablation builds, WRONG results by design: what the flush costs)
    if (hist[threadIdx.x] != 0x7fffffffu) return;
#endif
    if (!full_tile) {   // hand the copies of key0 back (block-uniform branch; with replicas the SUM over a bin's four comes out right)
        if (threadIdx.x == 0) hist[((uint32_t)key0 - (uint32_t)mn) * rmul] -= (uint32_t)(kQ5Tile - (tr.hi - tr.lo));
        __syncthreads();
    }
    if (c16 && kQ5Variant != 2) {   // (block-uniform) an even key and its successor: one atomic on the word they share
        const uint32_t d = (uint32_t)mn & 1u;   // bins are indexed from mn, pairs from the even key at or below it (offsets, so that a range across 0 needs no care)
        for (uint32_t s = threadIdx.x * 2; s <= span + d; s += kBlock * 2) {
            const uint32_t c0 = s >= d ? hist[s - d] : 0u, c1 = s + 1 - d <= span ? hist[s + 1 - d] : 0u;
            if (c0 | c1) emit_pair16((int32_t)((uint32_t)mn - d + s), c0, c1, f);
        }
        return;
    }
    for (uint32_t s = threadIdx.x; s <= span; s += kBlock) {
        uint32_t c;
        if (rep) {
            const uint4 r4 = *reinterpret_cast<const uint4 *>(&hist[s * 4]);
            c = r4.x + r4.y + r4.z + r4.w;
        } else {
            c = hist[s];
        }
        if (c) emit_pair((int32_t)((uint32_t)mn + s), c, f);
    }
}

template <bool kWeighted>
__global__ __launch_bounds__(kBlock) void q5_count_kernel(const int32_t *__restrict__ auction, const uint32_t *__restrict__ weight,
                                                          SegTiles st, const PaneDesc *__restrict__ panes,
                                                          const int32_t *__restrict__ pane_win_ptr,
                                                          const int32_t *__restrict__ pane_win_idx, uint32_t *counters,
                                                          uint64_t *tables, uint32_t cap, uint32_t *tab_used, uint32_t *err,
                                                          int32_t *slow_list, const uint64_t *__restrict__ spec_info,
                                                          unsigned long long *pane_wsum, int c16, unsigned long long *tab_rows) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[kWeighted ? 4 : kHist + kHistPad];
    __shared__ int32_t s_red[8];
    __shared__ unsigned long long s_w[kWavesPerBlock];
    if (spec_info && !spec_info[2]) return;  // the device layout declined this call
#if defined(FLOCKGPU_EXPERIMENTAL) && defined(FLOCKGPU_AB_Q5_PERSIST)   // (A/B builds only: num_cus x 8 workgroups walk the tiles)
    for (int32_t tile = (int32_t)blockIdx.x; tile < st.n_tiles; tile += (int32_t)gridDim.x) {
        q5_count_tile<kWeighted>(auction, weight, st, panes, pane_win_ptr, pane_win_idx, counters, tables, cap, tab_used, err, slow_list, pane_wsum, tile, hist, s_red, s_w, c16 != 0, tab_rows);
        __syncthreads();   // the next tile zeroes the histogram and rewrites the reduction slots
    }
#else
    q5_count_tile<kWeighted>(auction, weight, st, panes, pane_win_ptr, pane_win_idx, counters, tables, cap, tab_used, err, slow_list, pane_wsum, (int32_t)blockIdx.x, hist, s_red, s_w, c16 != 0, tab_rows);
#endif
}

#ifdef FLOCKGPU_EXPERIMENTAL   // (the count pass's other forms: measured in round 6, profiles/r06/q5_variants_ab.md; none beat the workgroup form)
// ---- count, wave-private form (round 6) ------------------------------------------------------------------------------------------
// The workgroup form above is co-bound by its LDS atomics (~53 % of the kernel's cycles) BECAUSE its phases are serial per workgroup: eight
// 16-byte loads per lane, a barrier (a workgroup-scope fence: `s_waitcnt vmcnt(0)` on gfx9), the 32 LDS adds per lane, a barrier, the
// flush -- while a workgroup counts it has nothing in flight, and the eight workgroups of a CU cover for each other only on average.
// Here a tile belongs to ONE wave with a private 2048-bin histogram (8 KB of LDS): no barrier anywhere (a wave's DS operations complete in
// order), so the wave streams its 8192 rows as four chunks of eight loads, the next chunk's loads in flight under the current chunk's
// LDS adds (`vmcnt(8)`, never 0), and nothing but the wave itself waits for its flush.  Bins are indexed by the key's low 11 bits, so no
// minimum has to be known before the first add; the tile's span (max - min, reduced over the wave at the end) says afterwards whether two
// keys shared a bin -- such a tile adds nothing and goes to the slow list, as the tiles wider than the workgroup form's histogram do.
constexpr int kWaveHist = 2048;
constexpr int kWaveChunks = 4, kWaveChunkIters = kQ5Iters * (kBlock / 64) / kWaveChunks;   // 4 chunks x 8 loads x 64 lanes x 4 keys = 8192 rows
static_assert(kWaveChunks * kWaveChunkIters * 64 * 4 == kQ5Tile, "a wave tile is a q5 tile");

__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

struct WaveCountState {
    int32_t hot;
    uint32_t hot_cnt;
    int32_t mn, mx;
};

// one chunk: eight iterations of the hot-key ballot + LDS adds of q5_count_tile, bins by the key's low bits
__device__ __forceinline__ void q5_wave_chunk(const int32_t (&k)[kWaveChunkIters][4], uint32_t *hist, WaveCountState &s, const int lane) {
#pragma unroll
    for (int it = 0; it < kWaveChunkIters; ++it) {
        s.mn = min(s.mn, min(min(k[it][0], k[it][1]), min(k[it][2], k[it][3])));
        s.mx = max(s.mx, max(max(k[it][0], k[it][1]), max(k[it][2], k[it][3])));
        uint64_t b0 = __ballot(k[it][0] == s.hot);
        if (__popcll((unsigned long long)b0) < kHotMin) {
            if (s.hot_cnt) {   // the candidate went cold: park its count
                if (lane == 0) atomicAdd(&hist[(uint32_t)s.hot & (kWaveHist - 1)], s.hot_cnt);
                s.hot_cnt = 0;
            }
            const int32_t c1 = __builtin_amdgcn_readfirstlane(k[it][0]);
            const uint64_t m1 = __ballot(k[it][0] == c1);
            s.hot = c1;
            b0 = m1;
            if (__popcll((unsigned long long)m1) < kHotMin && ~m1) {
                const int l2 = __ffsll((unsigned long long)~m1) - 1;
                const int32_t c2 = __builtin_amdgcn_readlane(k[it][0], l2);
                const uint64_t m2 = __ballot(k[it][0] == c2);
                if (__popcll((unsigned long long)m2) > __popcll((unsigned long long)m1)) {
                    s.hot = c2;
                    b0 = m2;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool is_hot = k[it][j] == s.hot;
            const uint64_t b = (j == 0) ? b0 : __ballot(is_hot);
            s.hot_cnt += (uint32_t)__popcll((unsigned long long)b);
            if (!is_hot) atomicAdd(&hist[(uint32_t)k[it][j] & (kWaveHist - 1)], 1u);
        }
    }
}

template <int kWaves>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(4, 5))) void q5_count_wave_kernel(
    const int32_t *__restrict__ auction, SegTiles st, const PaneDesc *__restrict__ panes, const int32_t *__restrict__ pane_win_ptr,
    const int32_t *__restrict__ pane_win_idx, uint32_t *counters, uint64_t *tables, uint32_t cap, uint32_t *tab_used, uint32_t *err,
    int32_t *slow_list, const uint64_t *__restrict__ spec_info) {
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[kWaves][kWaveHist];
    if (spec_info && !spec_info[2]) return;  // the device layout declined this call
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int32_t tile = (int32_t)blockIdx.x * kWaves + wave;
    if (tile >= st.n_tiles) return;
    uint32_t *hist = s_hist[wave];
    const TileRange tr = locate_tile(st, tile, kQ5Tile);
    const bool full_tile = tr.lo == tr.tile_begin && tr.hi == tr.tile_begin + kQ5Tile;
    // a uniform base (SGPRs) + one loop-invariant 32-bit lane offset: the loads then need no address VGPRs of their own
    const char *base = reinterpret_cast<const char *>(auction + tr.tile_begin);
    const uint32_t lane_off = (uint32_t)lane * 16u;
    int32_t ka[kWaveChunkIters][4], kb[kWaveChunkIters][4];
    if (!full_tile) {   // the first / last tile of a pane (~430 of 122 K at 1e9 bids): the general path
        if (lane == 0) slow_list[1 + atomicAdd(&slow_list[0], 1)] = tile;
        return;
    }
    // two loop-carried 32-bit lane offsets (the 13-bit immediate reaches 4 KB), advanced once per chunk, on ONE scalar base: no address
    // temporaries that the register allocator could place in a key register whose load the waitcnt pass still tracks
    uint32_t o0 = lane_off, o1 = lane_off + 4096u;
    auto load_chunk = [&](int32_t (&k)[kWaveChunkIters][4], int) {
        asm volatile("" : "+v"(o0), "+v"(o1));
#pragma unroll
        for (int it = 0; it < kWaveChunkIters; ++it) {
            const int4 t = stream_load4(reinterpret_cast<const int32_t *>(base + (it < 4 ? o0 : o1) + (it & 3) * 1024));
            k[it][0] = t.x; k[it][1] = t.y; k[it][2] = t.z; k[it][3] = t.w;
        }
        o0 += kWaveChunkIters * 1024;
        o1 += kWaveChunkIters * 1024;
    };
    load_chunk(ka, 0);
    {   // the histogram is zeroed under the first chunk's loads
        uint4 *z = reinterpret_cast<uint4 *>(hist);
#pragma unroll
        for (int i = 0; i < kWaveHist / 4 / 64; ++i) z[i * 64 + lane] = make_uint4(0, 0, 0, 0);
    }
    FlushArgs f;
    f.wp0 = pane_win_ptr[tr.seg];
    f.wp1 = pane_win_ptr[tr.seg + 1];
    if (f.wp0 == f.wp1) return;  // pane belongs to no (full) window
    f.pane = panes[tr.seg];
    f.pane_win_idx = pane_win_idx;
    f.counters = counters;
    f.tables = tables;
    f.cap = cap;
    f.tab_used = tab_used;
    f.err = err;
    WaveCountState s;
    s.hot_cnt = 0;
    s.mn = 0x7fffffff;
    s.mx = (int32_t)0x80000000;
    s.hot = 0;
    // a rolled loop over chunk pairs: ka / kb are loop-carried, so the register allocator keeps TWO chunks of keys (64 VGPRs), not four
#pragma unroll 1
    for (int c = 0;; c += 2) {
        load_chunk(kb, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        q5_wave_chunk(ka, hist, s, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 >= kWaveChunks) break;
        load_chunk(ka, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        q5_wave_chunk(kb, hist, s, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    q5_wave_chunk(kb, hist, s, lane);
    if (s.hot_cnt && lane == 0) atomicAdd(&hist[(uint32_t)s.hot & (kWaveHist - 1)], s.hot_cnt);
    const int32_t mn = wave_min_i32(s.mn), mx = wave_max_i32(s.mx);
    const uint32_t span = (uint32_t)mx - (uint32_t)mn;
    if (span >= (uint32_t)kWaveHist) {  // two keys may have shared a bin: nothing is flushed, the general path counts the tile
        if (lane == 0) slow_list[1 + atomicAdd(&slow_list[0], 1)] = (int32_t)((uint32_t)tile | kWideTile);
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the lanes' adds before the lanes' reads: one wave, DS operations in order)
    for (uint32_t o = lane; o <= span; o += 64) {
        const uint32_t key = (uint32_t)mn + o, c = hist[key & (kWaveHist - 1)];
        if (c) emit_pair((int32_t)key, c, f);
    }
}

// The same, PERSISTENT: wave g of G walks tiles g, g + G, ... and its loads never stop -- chunk 0 of the NEXT tile is requested before the
// last chunk of the current one is counted, so the flush (LDS reads, ~10 global atomics per lane) and the next tile's descriptor run under
// eight loads in flight.  The flush zeroes the bins it reads (only [min, max] can be non-zero), so the histogram is cleared once per wave.
// ONE rolled loop over chunk PAIRS across tiles (two copies of the chunk body, two key buffers = 64 VGPRs), the flush under a uniform branch.
template <int kWaves>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(4, 5))) void q5_count_wavep_kernel(
    const int32_t *__restrict__ auction, const TileRange *__restrict__ tiles, int32_t n_tiles, const PaneDesc *__restrict__ panes,
    const int32_t *__restrict__ pane_win_ptr, const int32_t *__restrict__ pane_win_idx, uint32_t *counters, uint64_t *tables, uint32_t cap,
    uint32_t *tab_used, uint32_t *err, int32_t *slow_list, const uint64_t *__restrict__ spec_info) {
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[kWaves][kWaveHist];
    if (spec_info && !spec_info[2]) return;  // the device layout declined this call
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int32_t stride = (int32_t)gridDim.x * kWaves;
    uint32_t *hist = s_hist[wave];
    {
        uint4 *z = reinterpret_cast<uint4 *>(hist);
#pragma unroll
        for (int i = 0; i < kWaveHist / 4 / 64; ++i) z[i * 64 + lane] = make_uint4(0, 0, 0, 0);
    }
    // two loop-invariant 32-bit lane offsets (the 13-bit immediate reaches 4 KB) on a scalar base per chunk, opaque to the compiler so that no
    // address arithmetic is re-materialised into a key register whose load the waitcnt pass still tracks
    uint32_t o0 = (uint32_t)lane * 16u, o1 = o0 + 4096u;
    asm volatile("" : "+v"(o0), "+v"(o1));
    // BUFFER loads: the tile's rows as a buffer resource in SGPRs, the chunk as the scalar offset, the lane as a 32-bit VGPR offset, the load
    // within the chunk as the immediate -- no address arithmetic in VGPRs at all
    auto load_chunk = [&](int32_t (&k)[kWaveChunkIters][4], __amdgpu_buffer_rsrc_t rs, int chunk) {
#pragma unroll
        for (int it = 0; it < kWaveChunkIters; ++it) {
            const flockgpu_v4u t = __builtin_amdgcn_raw_buffer_load_b128(rs, (it < 4 ? o0 : o1) + (uint32_t)((it & 3) * 1024), chunk * (kWaveChunkIters * 1024), 2 /* nt */);
            k[it][0] = (int32_t)t.x; k[it][1] = (int32_t)t.y; k[it][2] = (int32_t)t.z; k[it][3] = (int32_t)t.w;
        }
    };
    FlushArgs f;
    f.pane_win_idx = pane_win_idx;
    f.counters = counters;
    f.tables = tables;
    f.cap = cap;
    f.tab_used = tab_used;
    f.err = err;
    WaveCountState s;
    s.hot = 0;
    s.hot_cnt = 0;
    s.mn = 0x7fffffff;
    s.mx = (int32_t)0x80000000;
    int32_t ka[kWaveChunkIters][4], kb[kWaveChunkIters][4];
    // the tiles this wave streams: full tiles of panes that belong to a window; the others (a pane's first / last tile: the general path; panes
    // of no window: nothing) are settled on the way, without loads.  t: the candidate; result: tile (-1: none), its rows' address, its pane.
    // (`tiles` is a __restrict__ parameter of its own, not SegTiles' member: the descriptors then come through the scalar cache although
    // the loop stores in between)
    int32_t tile = -1, seg = 0;
    __amdgpu_buffer_rsrc_t tbase = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(auction), 0, 0, 0x00020000);   // the tile's rows as a buffer resource
    auto advance = [&](int32_t t) {
        tile = -1;
        for (;;) {
            t = __builtin_amdgcn_readfirstlane(t);   // (wave-uniform by construction)
            if (t >= n_tiles) return;
            const TileRange tr = tiles[t];
            if (pane_win_ptr[tr.seg] != pane_win_ptr[tr.seg + 1]) {
                if (tr.lo == tr.tile_begin && tr.hi == tr.tile_begin + kQ5Tile) {
                    tile = t;
                    seg = tr.seg;
                    tbase = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(auction + tr.tile_begin), 0, kQ5Tile * 4, 0x00020000);
                    return;
                }
                if (lane == 0) slow_list[1 + atomicAdd(&slow_list[0], 1)] = t;
            }
            t += stride;
        }
    };
    auto flush = [&](int32_t this_tile, int32_t this_seg) {
        if (s.hot_cnt) {   // the tile's counts are flushed: the hot key's share goes with them (the candidate itself stays)
            if (lane == 0) atomicAdd(&hist[(uint32_t)s.hot & (kWaveHist - 1)], s.hot_cnt);
            s.hot_cnt = 0;
        }
        const int32_t mn = wave_min_i32(s.mn), mx = wave_max_i32(s.mx);
        s.mn = 0x7fffffff;
        s.mx = (int32_t)0x80000000;
        const uint32_t span = (uint32_t)mx - (uint32_t)mn;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the lanes' adds before the lanes' reads: one wave, DS operations in order)
        if (span >= (uint32_t)kWaveHist) {  // two keys may have shared a bin: nothing is flushed, the general path counts the tile
            if (lane == 0) slow_list[1 + atomicAdd(&slow_list[0], 1)] = (int32_t)((uint32_t)this_tile | kWideTile);
            uint4 *z = reinterpret_cast<uint4 *>(hist);
#pragma unroll
            for (int i = 0; i < kWaveHist / 4 / 64; ++i) z[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        } else {
            f.wp0 = pane_win_ptr[this_seg];
            f.wp1 = pane_win_ptr[this_seg + 1];
            f.pane = panes[this_seg];
#pragma unroll 1
            for (uint32_t i = lane; i <= span; i += 64) {
                const uint32_t key = (uint32_t)mn + i, cnt = hist[key & (kWaveHist - 1)];
                if (cnt) {
                    hist[key & (kWaveHist - 1)] = 0;
                    emit_pair((int32_t)key, cnt, f);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    advance((int32_t)blockIdx.x * kWaves + wave);
    if (tile < 0) return;
    load_chunk(ka, tbase, 0);
    int c = 0;   // the chunk pair of the current tile: chunks (c, c + 1), c = 0 or 2
    // every trip issues exactly eight loads before each chunk body (so the waits stay `vmcnt(8 + n)`): the trip that finds no further tile
    // leaves the loop before its second half, and the tail below counts the last chunk
#pragma unroll 1
    for (;;) {
        load_chunk(kb, tbase, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        q5_wave_chunk(ka, hist, s, lane);
        __builtin_amdgcn_sched_barrier(0);
        const int32_t this_tile = tile, this_seg = seg;
        const bool last_pair = c != 0;
        if (last_pair) {
            advance(tile + stride);
            if (tile < 0) {
                tile = this_tile;
                seg = this_seg;
                break;
            }
        }
        c ^= 2;
        load_chunk(ka, tbase, c);
        __builtin_amdgcn_sched_barrier(0);
        q5_wave_chunk(kb, hist, s, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (last_pair) flush(this_tile, this_seg);
    }
    q5_wave_chunk(kb, hist, s, lane);
    flush(tile, seg);
}

// The workgroup form made PERSISTENT with the next tile's loads in flight under the LDS phase and the flush (the round-2 attempts at this died
// on `__syncthreads()`: a workgroup-scope fence, i.e. `s_waitcnt vmcnt(0)` -- the prefetch was waited for at the first barrier).  Here the
// barriers are raw: `s_waitcnt lgkmcnt(0); s_barrier` (the LDS traffic of a wave is what the other waves must see; nothing in flight from
// memory is anybody else's business).  Workgroup b walks tiles b, b + G, ...
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 6))) void q5_count_wgp_kernel(
    const int32_t *__restrict__ auction, const TileRange *__restrict__ tiles, int32_t n_tiles, const PaneDesc *__restrict__ panes,
    const int32_t *__restrict__ pane_win_ptr, const int32_t *__restrict__ pane_win_idx, uint32_t *counters, uint64_t *tables, uint32_t cap,
    uint32_t *tab_used, uint32_t *err, int32_t *slow_list, const uint64_t *__restrict__ spec_info) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[kHist + kHistPad];
    __shared__ int32_t s_red[2][8];
    if (spec_info && !spec_info[2]) return;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    FlushArgs f;
    f.pane_win_idx = pane_win_idx;
    f.counters = counters;
    f.tables = tables;
    f.cap = cap;
    f.tab_used = tab_used;
    f.err = err;
    // full tiles of counted panes are streamed; ragged ones go to the general path (as the wave forms do)
    auto next_streamed = [&](int32_t t, TileRange *tr) -> int32_t {
        for (; t < n_tiles; t += (int32_t)gridDim.x) {
            *tr = tiles[t];
            if (pane_win_ptr[tr->seg] == pane_win_ptr[tr->seg + 1]) continue;
            if (tr->lo == tr->tile_begin && tr->hi == tr->tile_begin + kQ5Tile) return t;
            if (threadIdx.x == 0) slow_list[1 + atomicAdd(&slow_list[0], 1)] = t;
        }
        return -1;
    };
    auto load_tile = [&](int32_t (&k)[kQ5Iters][4], const TileRange &tr) {
#pragma unroll
        for (int it = 0; it < kQ5Iters; ++it) {
            const int4 t = stream_load4(auction + tr.tile_begin + it * (kBlock * 4) + threadIdx.x * 4);
            k[it][0] = t.x; k[it][1] = t.y; k[it][2] = t.z; k[it][3] = t.w;
        }
    };
    TileRange tr;
    int32_t tile = next_streamed((int32_t)blockIdx.x, &tr);
    if (tile < 0) return;
    int32_t k[kQ5Iters][4], kn[kQ5Iters][4];
    load_tile(k, tr);
    int par = 0;
#pragma unroll 1
    for (;;) {
        {
            uint4 *z = reinterpret_cast<uint4 *>(hist);
            for (int s2 = threadIdx.x; s2 < (kHist + kHistPad) / 4; s2 += kBlock) z[s2] = make_uint4(0, 0, 0, 0);
        }
        int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
#pragma unroll
        for (int it = 0; it < kQ5Iters; ++it) {
            mn = min(mn, min(min(k[it][0], k[it][1]), min(k[it][2], k[it][3])));
            mx = max(mx, max(max(k[it][0], k[it][1]), max(k[it][2], k[it][3])));
        }
        // the next tile: its descriptor, then its rows -- in flight from here to the top of the next trip
        TileRange trn;
        const int32_t next = next_streamed(tile + (int32_t)gridDim.x, &trn);
        if (next >= 0) {
            load_tile(kn, trn);
            __builtin_amdgcn_sched_barrier(0);   // (the requests stay HERE: the scheduler would sink them below the LDS phase to save registers)
        }
        __builtin_amdgcn_sched_barrier(0);
        mn = wave_min_i32(mn);
        mx = wave_max_i32(mx);
        if (lane == 0) {
            s_red[par][wave] = mn;
            s_red[par][4 + wave] = mx;
        }
        lds_barrier();   // (histogram zeroed, extrema posted)
        mn = min(min(s_red[par][0], s_red[par][1]), min(s_red[par][2], s_red[par][3]));
        mx = max(max(s_red[par][4], s_red[par][5]), max(s_red[par][6], s_red[par][7]));
        par ^= 1;
        const uint32_t span = (uint32_t)mx - (uint32_t)mn;
        if (span >= (uint32_t)kHist) {   // (block-uniform)
            if (threadIdx.x == 0) slow_list[1 + atomicAdd(&slow_list[0], 1)] = (int32_t)((uint32_t)tile | kWideTile);
        } else {
            int32_t hot = __builtin_amdgcn_readfirstlane(k[0][0]);
            uint32_t hot_cnt = 0;
#pragma unroll
            for (int it = 0; it < kQ5Iters; ++it) {
                uint64_t b0 = __ballot(k[it][0] == hot);
                if (__popcll((unsigned long long)b0) < kHotMin) {
                    if (hot_cnt) {
                        if (lane == 0) atomicAdd(&hist[(uint32_t)hot - (uint32_t)mn], hot_cnt);
                        hot_cnt = 0;
                    }
                    const int32_t c1 = __builtin_amdgcn_readfirstlane(k[it][0]);
                    const uint64_t m1 = __ballot(k[it][0] == c1);
                    hot = c1;
                    b0 = m1;
                    if (__popcll((unsigned long long)m1) < kHotMin && ~m1) {
                        const int l2 = __ffsll((unsigned long long)~m1) - 1;
                        const int32_t c2 = __builtin_amdgcn_readlane(k[it][0], l2);
                        const uint64_t m2 = __ballot(k[it][0] == c2);
                        if (__popcll((unsigned long long)m2) > __popcll((unsigned long long)m1)) {
                            hot = c2;
                            b0 = m2;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool is_hot = k[it][j] == hot;
                    const uint64_t b = (j == 0) ? b0 : __ballot(is_hot);
                    hot_cnt += (uint32_t)__popcll((unsigned long long)b);
                    if (!is_hot) atomicAdd(&hist[(uint32_t)k[it][j] - (uint32_t)mn], 1u);
                }
            }
            if (hot_cnt && lane == 0) atomicAdd(&hist[(uint32_t)hot - (uint32_t)mn], hot_cnt);
            lds_barrier();   // (every wave's adds are in)
            f.wp0 = pane_win_ptr[tr.seg];
            f.wp1 = pane_win_ptr[tr.seg + 1];
            f.pane = panes[tr.seg];
            for (uint32_t s2 = threadIdx.x; s2 <= span; s2 += kBlock) {
                const uint32_t c = hist[s2];
                if (c) emit_pair((int32_t)((uint32_t)mn + s2), c, f);
            }
        }
        if (next < 0) break;
        lds_barrier();   // (the bins are read: the next trip may zero them)
        tile = next;
        tr = trn;
#pragma unroll
        for (int it = 0; it < kQ5Iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) k[it][j] = kn[it][j];
    }
}

#endif

// ---- Partial COUNT per tile (the exchange's stage 0): the histogram phase of q5_count_kernel, then the tile's bins are written as
// (key, count) pairs at a position claimed with ONE atomic per tile on the pane's cursor.  Ragged tiles and tiles whose keys spread wider than the histogram hand their rows out as
// (key, 1) pairs: exact, merely uncompressed (the FinalPartitioned side adds pairs up whatever their number).
__global__ __launch_bounds__(kBlock) void q5_partial_tile_kernel(const int32_t *__restrict__ auction, SegTiles st,
                                                                 const int64_t *__restrict__ region_start, uint32_t *cursor,
                                                                 int32_t *__restrict__ out_key, uint32_t *__restrict__ out_cnt) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[kHist + kHistPad];
    __shared__ int32_t s_red[8];
    __shared__ uint32_t s_base;
    // Workgroup b takes tile b / n_seg of pane b % n_seg: the ~2000 workgroups in flight then spread over all panes, and so do their
    // claims on the panes' cursors.  In tile order they would all sit in one or two panes: a returning atomic on one word completes
    // ~88 times per microsecond, i.e. 1.4 ms for the 122 K tiles of 1e9 bids (measured 1.28 ms; this order: see DESIGN.md section 8).
    const int32_t pane = (int32_t)(blockIdx.x % (uint32_t)st.n_seg);
    const int32_t tile = st.tile_first[pane] + (int32_t)(blockIdx.x / (uint32_t)st.n_seg);
    if (tile >= st.tile_first[pane + 1]) return;
    const TileRange tr = locate_tile(st, tile, kQ5Tile);
    const int64_t reg = region_start[tr.seg];
    if (reg < 0) return;  // pane of no window
    {
        uint4 *z = reinterpret_cast<uint4 *>(hist);
        for (int s = threadIdx.x; s < (kHist + kHistPad) / 4; s += kBlock) z[s] = make_uint4(0, 0, 0, 0);
    }
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const bool full_tile = tr.lo == tr.tile_begin && tr.hi == tr.tile_begin + kQ5Tile;
    int32_t k[kQ5Iters][4];
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
    if (full_tile) {
#pragma unroll
        for (int it = 0; it < kQ5Iters; ++it) {
            const int64_t r0 = tr.tile_begin + it * (kBlock * 4) + threadIdx.x * 4;
            const int4 t = stream_load4(auction + r0);
            k[it][0] = t.x; k[it][1] = t.y; k[it][2] = t.z; k[it][3] = t.w;
        }
#pragma unroll
        for (int it = 0; it < kQ5Iters; ++it) {
            mn = min(mn, min(min(k[it][0], k[it][1]), min(k[it][2], k[it][3])));
            mx = max(mx, max(max(k[it][0], k[it][1]), max(k[it][2], k[it][3])));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, __shfl_xor(mn, o, 64));
            mx = max(mx, __shfl_xor(mx, o, 64));
        }
        if (lane == 0) {
            s_red[wave] = mn;
            s_red[4 + wave] = mx;
        }
    }
    __syncthreads();  // also orders the zeroing of `hist`
    if (full_tile) {
        mn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
        mx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    }
    const uint32_t span = (uint32_t)mx - (uint32_t)mn;
    if (!full_tile || span >= (uint32_t)kHist) {   // rows as they are
        const uint32_t n = (uint32_t)(tr.hi - tr.lo);
        if (threadIdx.x == 0) s_base = atomicAdd(&cursor[tr.seg], n);
        __syncthreads();
        const int64_t o = reg + s_base;
        for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
            out_key[o + i] = auction[tr.lo + i];
            out_cnt[o + i] = 1u;
        }
        return;
    }
    // The tile hands out bins [0, span] of its histogram -- zero counts included -- as pairs in key order: span + 1 slots are claimed
    // NOW, before the histogram is built, so the claim's round trip (a returning atomic) is hidden behind the counting, and the write-out
    // is a plain coalesced copy of the bins: no compaction, no scan.  NEXMark's ids are dense (616 of a tile's 635 ids occur), so the
    // empty pairs are ~3 % of the output; for sparse keys they cost exchange bandwidth, never correctness (a zero count adds nothing),
    // and a tile never hands out more pairs than half its rows (span < kHist = 4096).
    if (threadIdx.x == 0) s_base = atomicAdd(&cursor[tr.seg], span + 1u);
    // hot key of this wave, kept in scalar registers across iterations (as in q5_count_kernel)
    int32_t hot = __builtin_amdgcn_readfirstlane(k[0][0]);
    uint32_t hot_cnt = 0;
#pragma unroll
    for (int it = 0; it < kQ5Iters; ++it) {
        uint64_t b0 = __ballot(k[it][0] == hot);
        if (__popcll((unsigned long long)b0) < kHotMin) {
            if (hot_cnt) {
                if (lane == 0) atomicAdd(&hist[(uint32_t)hot - (uint32_t)mn], hot_cnt);
                hot_cnt = 0;
            }
            const int32_t c1 = __builtin_amdgcn_readfirstlane(k[it][0]);
            const uint64_t m1 = __ballot(k[it][0] == c1);
            hot = c1;
            b0 = m1;
            if (__popcll((unsigned long long)m1) < kHotMin && ~m1) {
                const int l2 = __ffsll((unsigned long long)~m1) - 1;
                const int32_t c2 = __builtin_amdgcn_readlane(k[it][0], l2);
                const uint64_t m2 = __ballot(k[it][0] == c2);
                if (__popcll((unsigned long long)m2) > __popcll((unsigned long long)m1)) {
                    hot = c2;
                    b0 = m2;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool is_hot = k[it][j] == hot;
            const uint64_t b = (j == 0) ? b0 : __ballot(is_hot);
            hot_cnt += (uint32_t)__popcll((unsigned long long)b);
            if (!is_hot) atomicAdd(&hist[(uint32_t)k[it][j] - (uint32_t)mn], 1u);   // (hot lanes masked, not sent to a scratch bin: the count pass's variant 1)
        }
    }
    if (hot_cnt && lane == 0) atomicAdd(&hist[(uint32_t)hot - (uint32_t)mn], hot_cnt);
    __syncthreads();
    const int64_t o = reg + s_base;
    for (uint32_t b = threadIdx.x; b <= span; b += kBlock) {   // (read again only after the whole pass: kept out of the caches the bids stream through)
        __builtin_nontemporal_store((int32_t)((uint32_t)mn + b), out_key + o + b);
        __builtin_nontemporal_store(hist[b], out_cnt + o + b);
    }
}

// ---- count: general kernel for the tiles the fast kernel declined ------------------------------------------------
// num[i] = MAX of the window the i-th ordered winner belongs to
__global__ __launch_bounds__(kBlock) void q5_winner_num_kernel(const int32_t *__restrict__ win, const uint64_t *__restrict__ win_max,
                                                               int64_t n, uint64_t *__restrict__ num) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) num[i] = win_max[win[i]];
}

template <bool kWeighted>
__global__ __launch_bounds__(kBlock) void q5_count_slow_kernel(const int32_t *__restrict__ auction,
                                                               const uint32_t *__restrict__ weight, SegTiles st,
                                                               const PaneDesc *__restrict__ panes,
                                                               const int32_t *__restrict__ pane_win_ptr,
                                                               const int32_t *__restrict__ pane_win_idx, uint32_t *counters,
                                                               uint64_t *tables, uint32_t cap, uint32_t *tab_used,
                                                               uint32_t *err, const int32_t *__restrict__ slow_list,
                                                               const uint64_t *__restrict__ spec_info, int c16, unsigned long long *tab_rows) {
    __shared__ __attribute__((aligned(16))) uint64_t slots[kSlots];
    if (spec_info && !spec_info[2]) return;
    __shared__ uint32_t s_fill;  // slots claimed so far (approximate while lanes race: only steers the bypass)
    const int lane = lane_id();
    const int32_t n = slow_list[0];
    for (int32_t i = blockIdx.x; i < n; i += gridDim.x) {
        __syncthreads();  // previous tile's flush is done with `slots`
        for (int s = threadIdx.x; s < kSlots; s += kBlock) slots[s] = 0;
        if (threadIdx.x == 0) s_fill = 0;
        const uint32_t entry = (uint32_t)slow_list[1 + i];
        const TileRange tr = locate_tile(st, (int32_t)(entry & ~kWideTile), kQ5Tile);
        FlushArgs f;
        f.pane = panes[tr.seg];
        f.wp0 = pane_win_ptr[tr.seg];
        f.wp1 = pane_win_ptr[tr.seg + 1];
        f.pane_win_idx = pane_win_idx;
        f.counters = counters;
        f.tables = tables;
        f.cap = cap;
        f.tab_used = tab_used;
        f.err = err;
        f.c16 = c16 != 0;
        f.tab_rows = tab_rows ? tab_rows + tr.seg : nullptr;
        __syncthreads();
        if (kWeighted) {  // one (key, count) row per lane and trip: LDS hash while it has room, else straight out
#pragma unroll 1
            for (int64_t r0 = tr.lo; r0 < tr.hi; r0 += kBlock) {
                const int64_t r = r0 + threadIdx.x;
                if (r >= tr.hi) continue;
                const int32_t key = auction[r];
                const uint32_t w = weight[r];
                if (w == 0) continue;
                const bool full = *(volatile uint32_t *)&s_fill >= (uint32_t)(kSlots * 3 / 4);
                const int rc = (full || f.pane.range) ? 0 : lds_hash_insert(slots, (uint32_t)key, w);
                if (rc == 0) emit_pair(key, w, f);
                else if (rc == 2) atomicAdd(&s_fill, 1u);
            }
            __syncthreads();
#pragma unroll 1
            for (int s2 = threadIdx.x; s2 < kSlots; s2 += kBlock) {
                const uint64_t e = slots[s2];
                if (e) emit_pair((int32_t)(uint32_t)(e >> 32), (uint32_t)e, f);
            }
            continue;
        }
        if (f.pane.range && (entry & kWideTile)) {
            // Direct-address pane, tile spread wider than the LDS histogram (keys in no particular order): one global
            // atomic per row on the pane's counters; the lanes that share the first lane's key add once.  An LDS hash
            // in front only pays when a tile repeats keys, and it overflows first when it does not.  (Ragged tiles of
            // ordered data keep the LDS hash below: few distinct keys.)
            int32_t k[kQ5Iters][4];
#pragma unroll
            for (int it = 0; it < kQ5Iters; ++it) {
                const int64_t r0 = tr.tile_begin + it * (kBlock * 4) + threadIdx.x * 4;
                if (r0 >= tr.lo && r0 + 4 <= tr.hi) {
                    const int4 t = stream_load4(auction + r0);
                    k[it][0] = t.x; k[it][1] = t.y; k[it][2] = t.z; k[it][3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) k[it][j] = (r0 + j >= tr.lo && r0 + j < tr.hi) ? auction[r0 + j] : 0;
                }
            }
#pragma unroll
            for (int it = 0; it < kQ5Iters; ++it) {
                const int64_t r0 = tr.tile_begin + it * (kBlock * 4) + threadIdx.x * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool v = r0 + j >= tr.lo && r0 + j < tr.hi;
                    const uint64_t live = __ballot(v);
                    if (!live) continue;
                    const int32_t hot = __builtin_amdgcn_readlane(k[it][j], __ffsll((unsigned long long)live) - 1);
                    const uint64_t m = __ballot(v && k[it][j] == hot);
                    if (v && k[it][j] == hot) {
                        if (mbcnt(m) == 0) emit_pair(hot, (uint32_t)__popcll((unsigned long long)m), f);
                    } else if (v) {
                        emit_pair(k[it][j], 1u, f);
                    }
                }
            }
            continue;
        }
#pragma unroll 1
        for (int64_t r0 = tr.lo; r0 < tr.hi; r0 += kBlock) {
            const int64_t r = r0 + threadIdx.x;
            const bool v = r < tr.hi;
            const int32_t key = v ? auction[r] : 0;
            const uint64_t live = __ballot(v);
            if (!live) continue;
            // wave-level collapse of the wave's first key, the rest one LDS hash insert per lane
            const int src = __ffsll((unsigned long long)live) - 1;
            const int32_t hot = __builtin_amdgcn_readlane(key, src);
            const bool m = v && key == hot;
            const uint32_t cnt = (uint32_t)__popcll((unsigned long long)__ballot(m));
            // a tile of (nearly) distinct keys fills the LDS table after a quarter of its rows: from then on a probe
            // sequence only finds foreign keys, so the rest of the tile goes straight to the window tables
            const bool full = *(volatile uint32_t *)&s_fill >= (uint32_t)(kSlots * 3 / 4);
            if (lane == src) {
                const int r = full ? 0 : lds_hash_insert(slots, (uint32_t)hot, cnt);
                if (r == 0) emit_pair(hot, cnt, f);
                else if (r == 2) atomicAdd(&s_fill, 1u);
            } else if (v && !m) {
                const int r = full ? 0 : lds_hash_insert(slots, (uint32_t)key, 1u);
                if (r == 0) emit_pair(key, 1u, f);
                else if (r == 2) atomicAdd(&s_fill, 1u);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int s = threadIdx.x; s < kSlots; s += kBlock) {
            const uint64_t e = slots[s];
            if (e) emit_pair((int32_t)(uint32_t)(e >> 32), (uint32_t)e, f);
        }
    }
}

// ---- max + group count, then select ------------------------------------------------------------------------
// count(window, key) = sum over the window's panes of counters[key - pane.base] + the window's hash-table entry.
// Bases and ranges are multiples of 4, so an aligned 4-key group is entirely inside or outside a pane's range.
__device__ __forceinline__ uint4 window_counts4(const WinDesc &d, const PaneDesc *s_panes, int n_panes,
                                                const uint32_t *__restrict__ counters, int64_t k0) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int p = 0; p < n_panes; ++p) {
        const PaneDesc pd = s_panes[p];
        const uint64_t idx = (uint64_t)(k0 - pd.base);
        if (idx < (uint64_t)pd.range) {
            const uint4 c = *reinterpret_cast<const uint4 *>(counters + pd.cnt_off + idx);
            acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
        }
    }
    return acc;
}

template <bool SELECT>
__global__ __launch_bounds__(kBlock) void q5_scan_kernel(const WinDesc *__restrict__ wins, const PaneDesc *__restrict__ panes,
                                                         const uint32_t *__restrict__ counters,
                                                         const uint64_t *__restrict__ tables, uint32_t cap,
                                                         const uint32_t *__restrict__ tab_used, uint64_t *win_max,
                                                         uint64_t *win_groups, uint32_t *block_max, uint32_t *cursor,
                                                         uint32_t out_cap, int32_t *out_win, int32_t *out_key,
                                                         const uint64_t *__restrict__ spec_info) {
    __shared__ PaneDesc s_panes[kMaxWinPanes];
    if (spec_info && !spec_info[2]) return;
    const int32_t w = blockIdx.y;
    // select visits the same keys as the same block of the max pass did: a block whose maximum is not the window's
    // holds no winner and skips its share of the counters
    if (SELECT && block_max[(size_t)w * gridDim.x + blockIdx.x] != (uint32_t)win_max[w]) return;
    const WinDesc d = wins[w];
    const int n_panes = d.range ? d.pane_hi - d.pane_lo : 0;
    if ((int)threadIdx.x < n_panes) s_panes[threadIdx.x] = panes[d.pane_lo + threadIdx.x];
    __syncthreads();
    const uint64_t *tab = tables + (size_t)w * cap;
    const bool has_tab = tab_used[w] != 0;
    const uint32_t mx = SELECT ? (uint32_t)win_max[w] : 0;
    if (SELECT && mx == 0) return;  // empty window: MAX is NULL, the inner join emits nothing
    uint32_t best = 0, groups = 0;
    const uint32_t n4 = d.range / 4;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n4; i += gridDim.x * kBlock) {
        const int64_t k0 = d.base + (int64_t)i * 4;
        const uint4 c4 = window_counts4(d, s_panes, n_panes, counters, k0);
        uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
        if (has_tab) {  // keys of the union range that some pane sent to the hash table
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] += table_find(tab, cap, (uint32_t)(int32_t)(k0 + j));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (SELECT) {
                // one cursor bump per wave: when every group ties for the MAX each lane is a winner
                const bool hit = c[j] == mx;
                const uint64_t b = __ballot(hit);
                if (b) {
                    const int leader = __ffsll((unsigned long long)b) - 1;
                    uint32_t base = 0;
                    if (lane_id() == leader) base = atomicAdd(cursor, (uint32_t)__popcll((unsigned long long)b));
                    base = __builtin_amdgcn_readlane(base, leader);
                    const uint32_t p = base + mbcnt(b);
                    if (hit && p < out_cap) {
                        out_win[p] = w;
                        out_key[p] = (int32_t)(k0 + j);
                    }
                }
            } else {
                best = max(best, c[j]);
                groups += c[j] != 0;
            }
        }
    }
    if (has_tab) {  // table entries whose key lies outside the union range are complete on their own
        for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s < cap; s += gridDim.x * kBlock) {
            const uint64_t e = tab[s];
            if (!e) continue;
            const int64_t key = (int32_t)(uint32_t)(e >> 32);
            if ((uint64_t)(key - d.base) < (uint64_t)d.range) continue;  // already counted above
            if (SELECT) {
                if ((uint32_t)e == mx) {
                    const uint32_t p = atomicAdd(cursor, 1u);
                    if (p < out_cap) {
                        out_win[p] = w;
                        out_key[p] = (int32_t)key;
                    }
                }
            } else {
                best = max(best, (uint32_t)e);
                ++groups;
            }
        }
    }
    if (!SELECT) {
        // one update per WORKGROUP: every wave of every workgroup of a window bumping win_max[w] / win_groups[w] is a
        // same-address atomic chain (the pass took 0.10 / 0.18 / 0.75 ms with 8 / 64 / 256 workgroups per window)
        __shared__ uint32_t s_best[kWavesPerBlock];
        __shared__ uint64_t s_groups[kWavesPerBlock];
        best = wave_max_u32(best);
        const uint64_t g = wave_sum_u64(groups);
        if (lane_id() == 0) {
            s_best[threadIdx.x >> 6] = best;
            s_groups[threadIdx.x >> 6] = g;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t b = 0;
            uint64_t gs = 0;
#pragma unroll
            for (int v = 0; v < kWavesPerBlock; ++v) {
                b = max(b, s_best[v]);
                gs += s_groups[v];
            }
            if (b) block_max[(size_t)w * gridDim.x + blockIdx.x] = b;  // this workgroup's own slot
            if (b) atomicMax(reinterpret_cast<unsigned long long *>(&win_max[w]), (unsigned long long)b);
            if (gs) atomicAdd(reinterpret_cast<unsigned long long *>(&win_groups[w]), (unsigned long long)gs);
        }
    }
}

// ---- max + select for windows of one or two consecutive panes (ElementWise / Tumbling / Hopping(2h, h)), walking PANES --------------
// q5_scan_kernel walks windows: under Hopping(10 s, 5 s) every pane is in two windows and its counters are read twice per pass
// (0.57 GB per 1e9 bids at 5.4 TB/s = 0.105 ms: bandwidth, not instructions).  Here workgroup (x, p) reads its share of pane p's
// counters ONCE and serves both windows of the pane:
//   as the FIRST pane of window wb = (p, p + 1): count(k) = c_p[k] + c_{p+1}[k] where pane p + 1 covers k (the few auctions in flight
//     at the pane boundary), and every key of pane p is accounted to wb here;
//   as the SECOND pane of window wa = (p - 1, p): only the keys pane p - 1 does NOT cover (the others were accounted by its sweep).
// A generic version of this (any number of panes per window and windows per pane, descriptors in LDS) measured 0.134 + 0.048 ms
// against 0.105 + 0.015 ms of the window walk: instruction overhead ate the saved traffic.  This one is specialised to the two roles.
// k16: 16-bit counters, two per word (see kQ5Counters16) -- a lane's 16-byte group holds EIGHT keys; the max pass then also adds up
// every counter of the pane it reads (pane_sum: the overflow check of q5_finish_kernel).
template <bool SELECT, bool k16>
__global__ __launch_bounds__(kBlock) void q5_hop2_scan_kernel(const PaneDesc *__restrict__ panes, const int32_t *__restrict__ pane_wa,
                                                              const int32_t *__restrict__ pane_wb, const uint32_t *__restrict__ counters,
                                                              const uint64_t *__restrict__ tables, uint32_t cap, const uint32_t *__restrict__ tab_used,
                                                              uint64_t *win_max, uint64_t *win_groups, uint32_t *block_max, uint32_t *cursor,
                                                              uint32_t out_cap, int32_t *out_win, int32_t *out_key, const uint64_t *__restrict__ spec_info,
                                                              unsigned long long *pane_sum) {
    constexpr int G = k16 ? 8 : 4;   // keys per 16-byte group of counters
    __shared__ uint32_t s_best[2][kWavesPerBlock];
    __shared__ uint64_t s_groups[3][kWavesPerBlock];
    if (spec_info && !spec_info[2]) return;
    const int32_t p = blockIdx.y;
    const int32_t wa = pane_wa[p], wb = pane_wb[p];
    if (wa < 0 && wb < 0) return;
    const PaneDesc pd = panes[p];
    const PaneDesc prev = wa >= 0 ? panes[p - 1] : PaneDesc{0, 0, 0, 0};
    const bool two_b = wb >= 0 && pane_wa[p + 1] == wb;   // wb has a second pane (p + 1) -- pane_wa has n_panes + 1 entries, the last one -1
    const PaneDesc next = two_b ? panes[p + 1] : PaneDesc{0, 0, 0, 0};
    const uint32_t mx_a = SELECT && wa >= 0 ? (uint32_t)win_max[wa] : 0u, mx_b = SELECT && wb >= 0 ? (uint32_t)win_max[wb] : 0u;
    // select revisits the keys the same workgroup saw in the max pass: no winner of a window where its maximum was lower
    bool act_a = wa >= 0, act_b = wb >= 0;
    if (SELECT) {
        act_a = act_a && mx_a != 0 && block_max[((size_t)wa * 2 + 1) * gridDim.x + blockIdx.x] == mx_a;
        act_b = act_b && mx_b != 0 && block_max[((size_t)wb * 2 + 0) * gridDim.x + blockIdx.x] == mx_b;
        if (!act_a && !act_b) return;
    }
    const bool tab_a = wa >= 0 && tab_used[wa] != 0, tab_b = wb >= 0 && tab_used[wb] != 0;
    const uint64_t *ta = tables + (size_t)(wa >= 0 ? wa : 0) * cap, *tb = tables + (size_t)(wb >= 0 ? wb : 0) * cap;
    uint32_t best_a = 0, best_b = 0, groups_a = 0, groups_b = 0;
    uint64_t sum = 0;
    auto take = [&](uint32_t c, int32_t key, int32_t w, uint32_t mx, uint32_t &best, uint32_t &groups) {
        if (SELECT) {
            const bool hit = c == mx;
            const uint64_t b = __ballot(hit);
            if (b) {   // one cursor bump per wave
                const int leader = __ffsll((unsigned long long)b) - 1;
                uint32_t base = 0;
                if (lane_id() == leader) base = atomicAdd(cursor, (uint32_t)__popcll((unsigned long long)b));
                base = __builtin_amdgcn_readlane(base, leader);
                const uint32_t pos = base + mbcnt(b);
                if (hit && pos < out_cap) {
                    out_win[pos] = w;
                    out_key[pos] = key;
                }
            }
        } else {
            best = max(best, c);
            groups += c != 0;
        }
    };
    auto unpack = [&](const uint4 v, uint32_t (&c)[G]) {
        if constexpr (k16) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[2 * j] = w[j] & 0xFFFFu;
                c[2 * j + 1] = w[j] >> 16;
            }
        } else {
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        }
    };
    const uint32_t ng = pd.range / G;   // (ranges are multiples of 8)
    const uint32_t *mine = counters + (k16 ? pd.cnt_off / 2 : pd.cnt_off), *theirs = counters + (k16 ? next.cnt_off / 2 : next.cnt_off);
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < ng; i += gridDim.x * kBlock) {
        const int64_t k0 = pd.base + (int64_t)i * G;
        uint32_t me[G];
        unpack(*reinterpret_cast<const uint4 *>(mine + (uint64_t)i * 4), me);
        if (k16 && !SELECT) {
#pragma unroll
            for (int j = 0; j < G; ++j) sum += me[j];
        }
        if (act_b) {   // (block-uniform)
            uint32_t c[G];
#pragma unroll
            for (int j = 0; j < G; ++j) c[j] = me[j];
            const uint64_t idx = (uint64_t)(k0 - next.base);
            if (idx < (uint64_t)next.range) {   // (bases and ranges are multiples of 8: the aligned group is covered as a whole or not at all)
                uint32_t o[G];
                unpack(*reinterpret_cast<const uint4 *>(theirs + (k16 ? idx / 2 : idx)), o);
#pragma unroll
                for (int j = 0; j < G; ++j) c[j] += o[j];
            }
            if (tab_b) {
#pragma unroll
                for (int j = 0; j < G; ++j) c[j] += table_find(tb, cap, (uint32_t)(int32_t)(k0 + j));
            }
#pragma unroll
            for (int j = 0; j < G; ++j) take(c[j], (int32_t)(k0 + j), wb, mx_b, best_b, groups_b);
        }
        if (act_a && !((uint64_t)(k0 - prev.base) < (uint64_t)prev.range)) {
            uint32_t c[G];
#pragma unroll
            for (int j = 0; j < G; ++j) c[j] = me[j];
            if (tab_a) {
#pragma unroll
                for (int j = 0; j < G; ++j) c[j] += table_find(ta, cap, (uint32_t)(int32_t)(k0 + j));
            }
#pragma unroll
            for (int j = 0; j < G; ++j) take(c[j], (int32_t)(k0 + j), wa, mx_a, best_a, groups_a);
        }
    }
    if (tab_b && act_b) {   // straggler-table entries of wb that neither of its panes covers: complete on their own
        for (uint32_t sl = blockIdx.x * kBlock + threadIdx.x; sl < cap; sl += gridDim.x * kBlock) {
            const uint64_t e = tb[sl];
            if (!e) continue;
            const int64_t key = (int32_t)(uint32_t)(e >> 32);
            if ((uint64_t)(key - pd.base) < (uint64_t)pd.range || (uint64_t)(key - next.base) < (uint64_t)next.range) continue;
            if (SELECT) {
                if ((uint32_t)e == mx_b) {
                    const uint32_t pos = atomicAdd(cursor, 1u);
                    if (pos < out_cap) {
                        out_win[pos] = wb;
                        out_key[pos] = (int32_t)key;
                    }
                }
            } else {
                best_b = max(best_b, (uint32_t)e);
                groups_b += 1;
            }
        }
    }
    if (!SELECT) {   // one update per workgroup and window
        const uint32_t ba = wave_max_u32(best_a), bb = wave_max_u32(best_b);
        const uint64_t ga = wave_sum_u64(groups_a), gb = wave_sum_u64(groups_b), gs = k16 ? wave_sum_u64(sum) : 0;
        if (lane_id() == 0) {
            s_best[0][threadIdx.x >> 6] = ba;
            s_best[1][threadIdx.x >> 6] = bb;
            s_groups[0][threadIdx.x >> 6] = ga;
            s_groups[1][threadIdx.x >> 6] = gb;
            s_groups[2][threadIdx.x >> 6] = gs;
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            const int r = threadIdx.x;   // 0: role a (second pane of wa), 1: role b (first pane of wb)
            const int32_t w = r == 0 ? wa : wb;
            uint32_t b = 0;
            uint64_t g = 0;
#pragma unroll
            for (int v = 0; v < kWavesPerBlock; ++v) {
                b = max(b, s_best[r][v]);
                g += s_groups[r][v];
            }
            if (w >= 0) {
                if (b) block_max[((size_t)w * 2 + (r == 0 ? 1 : 0)) * gridDim.x + blockIdx.x] = b;
                if (b) atomicMax(reinterpret_cast<unsigned long long *>(&win_max[w]), (unsigned long long)b);
                if (g) atomicAdd(reinterpret_cast<unsigned long long *>(&win_groups[w]), (unsigned long long)g);
            }
        }
        if (k16 && threadIdx.x == 2) {   // this workgroup's share of the pane's counter sum
            uint64_t t = 0;
#pragma unroll
            for (int v = 0; v < kWavesPerBlock; ++v) t += s_groups[2][v];
            if (t) atomicAdd(&pane_sum[p], (unsigned long long)t);
        }
    }
}

// ---- the call's read-back in ONE kernel -----------------------------------------------------------------------------------------
// A call used to end with four device-to-host copy nodes (scalars, the first winners' windows and keys, the device layout's verdict)
// and, after the host had ordered the winners, two host-to-device copies of the result columns: nine copy nodes of ~4 us each per
// call with the schedule upload and the sample, 4 % of the 1e9-bid step (rocprofv3: 54 `copyBuffer` for 6 calls).  One workgroup does
// it all: it copies the scalars into ONE pinned block with plain stores, orders up to kFinishMax winners by (window, auction) with a
// bitonic sort in LDS, writes the result columns where they stay (device) and the windows' output offsets into the pinned block.
// More winners than that (mass ties) or a failed attempt: sorted = 0 and the host takes the old road.
constexpr int kFinishThreads = 1024;
constexpr uint32_t kFinishMax = 1024;
__global__ __launch_bounds__(kFinishThreads) void q5_finish_kernel(const uint64_t *__restrict__ meta, uint32_t n_meta, int32_t n_win, const uint64_t *__restrict__ info,
                                                                   const int32_t *__restrict__ slow_list, const int32_t *__restrict__ sel_win,
                                                                   const int32_t *__restrict__ sel_key, uint32_t out_cap, int32_t *__restrict__ out_auction,
                                                                   uint64_t *__restrict__ out_num, uint64_t *__restrict__ h_fin,
                                                                   const unsigned long long *__restrict__ pane_sum, const unsigned long long *__restrict__ tab_rows,
                                                                   const int64_t *__restrict__ seg_off, const int32_t *__restrict__ pane_win_ptr, int32_t n_panes) {
    __shared__ uint64_t s_k[kFinishMax];
    __shared__ uint32_t s_bad;
    const uint32_t *tail = reinterpret_cast<const uint32_t *>(meta + 2 * (size_t)n_win);
    const uint32_t n_sel = tail[0], err = tail[1];
    // 16-bit counters: every counted pane's counters (added up by the max pass) + the rows it sent to the straggler tables = its rows, unless a
    // count outgrew its 16 bits (kQ5Counters16).  (A call the device layout declined counted nothing: the host repeats it anyway.)
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    if (pane_sum && (!info || info[2]))
        for (int32_t p = threadIdx.x; p < n_panes; p += kFinishThreads)
            if (pane_win_ptr[p + 1] != pane_win_ptr[p] && pane_sum[p] + tab_rows[p] != (unsigned long long)(seg_off[2 * p + 1] - seg_off[2 * p])) s_bad = 1;
    __syncthreads();
    const uint32_t bad16 = s_bad;
    for (uint32_t i = threadIdx.x; i < n_meta; i += kFinishThreads) h_fin[i] = meta[i];
    if (threadIdx.x < 3) h_fin[n_meta + threadIdx.x] = info ? info[threadIdx.x] : 0;
    if (threadIdx.x == 3) h_fin[n_meta + 3] = (uint64_t)(uint32_t)slow_list[0] | ((uint64_t)bad16 << 32);
    const bool sortable = !err && !bad16 && n_sel <= kFinishMax && n_sel <= out_cap && (!info || info[2]);
    if (threadIdx.x == 4) h_fin[n_meta + 4] = sortable ? 1 : 0;
    int64_t *h_off = reinterpret_cast<int64_t *>(h_fin + n_meta + 5);
    if (!sortable) return;   // (block-uniform)
    uint32_t p2 = 1;
    while (p2 < n_sel) p2 <<= 1;
    if (threadIdx.x < p2)
        s_k[threadIdx.x] = threadIdx.x < n_sel ? ((uint64_t)(uint32_t)sel_win[threadIdx.x] << 32) | ((uint32_t)sel_key[threadIdx.x] ^ 0x80000000u) : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= p2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t i = threadIdx.x, l = i ^ j;
            if (i < p2 && l > i) {
                const uint64_t a = s_k[i], b = s_k[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    s_k[i] = b;
                    s_k[l] = a;
                }
            }
            __syncthreads();
        }
    if (threadIdx.x < n_sel) {
        const uint64_t e = s_k[threadIdx.x];
        out_auction[threadIdx.x] = (int32_t)((uint32_t)e ^ 0x80000000u);
        out_num[threadIdx.x] = meta[(uint32_t)(e >> 32)];   // the window's MAX
    }
    for (int32_t w = threadIdx.x; w <= n_win; w += kFinishThreads) {   // offsets[w] = first winner of a window >= w
        uint32_t lo = 0, hi = n_sel;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((int32_t)(s_k[mid] >> 32) < w) lo = mid + 1;
            else hi = mid;
        }
        h_off[w] = (int64_t)lo;
    }
}

// ---- keys in no particular order: partition by key range, then count in LDS ("wide" mode) -----------------------------------------
// The kernels above are built for keys that sweep their range with time (a tile of 8192 bids names ~600 consecutive auctions).  Bids
// whose keys are spread over the whole pane range (3e5 ids per 5-s pane; `also.q5_uniform`: the generator's bids shuffled inside
// their panes) make every tile "wide": the LDS histogram declines it and q5_count_slow_kernel ends up issuing one global atomic per
// ROW on a random counter -- 2.5e10 atomics / s is what the memory side gives (39 ms per 1e9 bids).  General hash aggregation the
// LDS-staged way instead:
//   partition : one radix pass over the pane's rows on the HIGH bits of (key - pane base): digit = (key - base) >> kPartShift, one more
//               digit for the keys outside the pane's estimated range; count -> scan -> emit as in sort.hip (lanes holding one digit
//               find each other by ballots, rows leave the tile regrouped by digit as runs of consecutive addresses).  The histogram
//               matrix is laid out (pane, digit, tile), so ONE scan yields every (pane, digit) bucket as a contiguous run of keys.
//   count     : one workgroup per (pane, digit) bucket streams its keys into an LDS histogram over the bucket's 2^kPartShift counters
//               (the first lane's key counted once per instruction: hot keys) and STORES the bins -- it owns them: no global atomics.
//               The out-of-range bucket goes row by row through emit_pair (the windows' straggler tables), as before.
// 16 B of traffic per row instead of 4, and ~10x faster than a random global atomic per row.  Chosen from what the previous call of the
// ctx saw (most tiles wide) and left again when a sample of the partition's tiles turns out narrow.
constexpr int kPartItems = 16;
constexpr int kPartTile = kBlock * kPartItems;           // 4096 rows
constexpr int kPartShift = 13;                           // 8192 counters = 32 KB of LDS per bucket: four workgroups per CU (64 KB: two, 1.78 vs 0.97 ms)
constexpr int kPartMaxDigits = 256;

__device__ __forceinline__ uint32_t part_digit(int32_t key, const PaneDesc &pd, uint32_t straggler) {
    const uint64_t idx = (uint64_t)((int64_t)key - pd.base);
    return idx < (uint64_t)pd.range ? (uint32_t)(idx >> kPartShift) : straggler;
}

// Round 4: the partition keeps every tile's rows in the tile's OWN region of the side buffer -- grouped by digit there, with the digits'
// start offsets in a small matrix  off[(tile_first[pane] * (nd + 1)) + digit * tiles_of_pane + tile_in_pane]  (row nd: the tile's rows) --
// so no output position depends on another tile: the count pass over the column (0.93 ms per 1e9 bids), the scan of its histogram matrix
// and the ballot matching that kept rows in order are gone (COUNT does not care in which order a bucket's rows arrive: a row's rank in
// its digit is what one returning LDS add says).  A bucket's workgroup reads its digit's row of the matrix and the next one -- both
// contiguous -- and streams the runs they delimit.
__global__ __launch_bounds__(kBlock) void q5_part_tile_kernel(const int32_t *__restrict__ auction, SegTiles st, const PaneDesc *__restrict__ panes,
                                                              const int32_t *__restrict__ pane_win_ptr, uint32_t nd, int32_t *__restrict__ off_mat,
                                                              int32_t *__restrict__ keys_out, uint16_t *__restrict__ low_out,
                                                              uint32_t *__restrict__ sample) {
    __shared__ uint32_t s_cnt[kPartMaxDigits];
    __shared__ __attribute__((aligned(16))) uint16_t s_low[kPartTile];
    __shared__ uint32_t s_wave_total[kWavesPerBlock];
    __shared__ int32_t s_red[2 * kWavesPerBlock];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    s_cnt[threadIdx.x] = 0;
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kPartTile);
    if (pane_win_ptr[tr.seg] == pane_win_ptr[tr.seg + 1]) return;   // (block-uniform) the pane is in no window: its buckets are not counted
    const int32_t t0 = st.tile_first[tr.seg], tiles_p = st.tile_first[tr.seg + 1] - t0;
    int32_t *off = off_mat + (size_t)t0 * (nd + 1) + (tile - t0);
    const PaneDesc pd = panes[tr.seg];
    int32_t k[kPartItems];
#pragma unroll
    for (int it = 0; it < kPartItems / 4; ++it) {
        const int64_t r0 = tr.tile_begin + (int64_t)(it * kBlock + threadIdx.x) * 4;
        if (r0 >= tr.lo && r0 + 4 <= tr.hi) {
            const int4 t = stream_load4(auction + r0);
            k[it * 4 + 0] = t.x; k[it * 4 + 1] = t.y; k[it * 4 + 2] = t.z; k[it * 4 + 3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) k[it * 4 + j] = (r0 + j >= tr.lo && r0 + j < tr.hi) ? auction[r0 + j] : 0;
        }
    }
    __syncthreads();
    uint16_t rank[kPartItems];
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
#pragma unroll
    for (int i = 0; i < kPartItems; ++i) {
        const int64_t r = tr.tile_begin + (int64_t)((i >> 2) * kBlock + threadIdx.x) * 4 + (i & 3);
        rank[i] = 0;
        if (r >= tr.lo && r < tr.hi) {
            rank[i] = (uint16_t)atomicAdd(&s_cnt[part_digit(k[i], pd, nd - 1)], 1u);
            mn = min(mn, k[i]);
            mx = max(mx, k[i]);
        }
    }
    __syncthreads();
    {   // thread d: where digit d's rows start in the tile
        const uint32_t c = s_cnt[threadIdx.x];
        const uint32_t in = wave_incl_scan_u32(c);
        if (lane == 63) s_wave_total[wave] = in;
        __syncthreads();
        uint32_t o = in - c;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) o += w < wave ? s_wave_total[w] : 0u;
        s_cnt[threadIdx.x] = o;
        if (threadIdx.x < nd) off[(size_t)threadIdx.x * tiles_p] = (int32_t)o;
        if (threadIdx.x == 0) off[(size_t)nd * tiles_p] = (int32_t)(tr.hi - tr.lo);
    }
    __syncthreads();
    // Inside a bucket only the key's low kPartShift bits are news (the digit IS the bucket), so a partitioned key travels as 16 bits.  The
    // straggler bucket -- keys outside the pane's estimated range -- keeps whole keys, in the int32 array at the same positions (touched
    // only where stragglers sit).
#pragma unroll
    for (int i = 0; i < kPartItems; ++i) {
        const int64_t r = tr.tile_begin + (int64_t)((i >> 2) * kBlock + threadIdx.x) * 4 + (i & 3);
        if (r >= tr.lo && r < tr.hi) {
            const uint32_t d = part_digit(k[i], pd, nd - 1), pos = s_cnt[d] + rank[i];
            if (d == nd - 1) keys_out[(size_t)tile * kPartTile + pos] = k[i];
            else s_low[pos] = (uint16_t)(((uint32_t)k[i] - (uint32_t)pd.base) & ((1u << kPartShift) - 1));
        }
    }
    __syncthreads();
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(s_low);
        uint4 *dst = reinterpret_cast<uint4 *>(low_out + (size_t)tile * kPartTile);
        const uint32_t n16 = ((uint32_t)(tr.hi - tr.lo) + 7u) >> 3;
        for (uint32_t i = threadIdx.x; i < n16; i += kBlock) stream_store4(dst + i, src[i]);
    }
    if ((blockIdx.x & 63u) == 0) {   // a sample of the tiles reports whether the fast kernel would have taken it (span below its histogram)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn = min(mn, __shfl_xor(mn, o, 64));
            mx = max(mx, __shfl_xor(mx, o, 64));
        }
        if (lane == 0) {
            s_red[wave] = mn;
            s_red[kWavesPerBlock + wave] = mx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kWavesPerBlock; ++w) {
                mn = min(mn, s_red[w]);
                mx = max(mx, s_red[kWavesPerBlock + w]);
            }
            atomicAdd(&sample[0], 1u);
            if (mx >= mn && (uint32_t)mx - (uint32_t)mn < (uint32_t)kHist / 2) atomicAdd(&sample[1], 1u);
        }
    }
}

// One workgroup per (digit, pane) bucket: the runs [off[d][t], off[d + 1][t]) of the pane's tile regions, all keys inside
// [base + digit << shift, + 2^shift).
__global__ __launch_bounds__(kBlock) void q5_bucket_count_kernel(const int32_t *__restrict__ keys, const uint16_t *__restrict__ low, SegTiles st,
                                                                 const PaneDesc *__restrict__ panes,
                                                                 const int32_t *__restrict__ pane_win_ptr, const int32_t *__restrict__ pane_win_idx,
                                                                 uint32_t nd, const int32_t *__restrict__ off_mat, uint32_t *counters, uint64_t *tables,
                                                                 uint32_t cap, uint32_t *tab_used, uint32_t *err) {
    __shared__ uint32_t s_hist[1 << kPartShift];
    const uint32_t d = blockIdx.x;
    const int32_t p = blockIdx.y;
    const int32_t t0 = st.tile_first[p], tiles_p = st.tile_first[p + 1] - t0;
    if (tiles_p == 0 || pane_win_ptr[p] == pane_win_ptr[p + 1]) return;
    const int32_t *lo_row = off_mat + (size_t)t0 * (nd + 1) + (size_t)d * tiles_p, *hi_row = lo_row + tiles_p;
    const PaneDesc pd = panes[p];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (d == nd - 1) {   // keys outside the pane's estimated range: the windows' straggler tables, row by row (a wave per tile)
        FlushArgs f;
        f.wp0 = pane_win_ptr[p];
        f.wp1 = pane_win_ptr[p + 1];
        f.pane = pd;
        f.pane_win_idx = pane_win_idx;
        f.counters = counters;
        f.tables = tables;
        f.cap = cap;
        f.tab_used = tab_used;
        f.err = err;
        for (int32_t t = wave; t < tiles_p; t += kWavesPerBlock) {
            const int32_t a = lo_row[t], b = hi_row[t];
            for (int32_t i = a + lane; i < b; i += 64) emit_pair(keys[(size_t)(t0 + t) * kPartTile + i], 1u, f);
        }
        return;
    }
    {
        uint4 *z = reinterpret_cast<uint4 *>(s_hist);
        for (int i = threadIdx.x; i < (1 << kPartShift) / 4; i += kBlock) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    auto add = [&](uint32_t bin, bool valid) {
        // the first lane's key once per instruction: half of NEXMark's bids name one auction, wherever they sit
        const uint32_t hot = __builtin_amdgcn_readfirstlane(bin);
        const uint64_t b = __ballot(valid && bin == hot);
        if (valid && bin == hot) {
            if (mbcnt(b) == 0) atomicAdd(&s_hist[hot], (uint32_t)__popcll((unsigned long long)b));
        } else if (valid) {
            atomicAdd(&s_hist[bin], 1u);
        }
    };
    auto add4 = [&](uint2 v, int32_t q, int32_t a, int32_t b, bool act) {   // bins 4q .. 4q + 3 of a tile region, those inside [a, b)
        add(v.x & 0xFFFFu, act && 4 * q >= a && 4 * q < b);
        add(v.x >> 16, act && 4 * q + 1 >= a && 4 * q + 1 < b);
        add(v.y & 0xFFFFu, act && 4 * q + 2 >= a && 4 * q + 2 < b);
        add(v.y >> 16, act && 4 * q + 3 >= a && 4 * q + 3 < b);
    };
    // A wave takes 64 tiles at a time: lane l fetches tile l's run bounds (the next 64 tiles' while these are streamed), then the runs
    // are streamed two at a time (one per half-wave: a run of ~4096 / nd rows is a few dozen 8-byte quads) through kDepth slots: a slot
    // is counted and at once requested again for the pair kDepth further on, so kDepth - 1 loads are in flight under every count.
    constexpr int kDepth = 8;
    const int half = lane >> 5, hl = lane & 31;
    int32_t c0 = wave * 64;
    int32_t ra = 0, rb = 0;
    if (c0 + lane < tiles_p) {
        ra = lo_row[c0 + lane];
        rb = hi_row[c0 + lane];
    }
    for (; c0 < tiles_p; c0 += kWavesPerBlock * 64) {
        const int32_t cn = c0 + kWavesPerBlock * 64;
        int32_t na = 0, nb2 = 0;
        if (cn + lane < tiles_p) {
            na = lo_row[cn + lane];
            nb2 = hi_row[cn + lane];
        }
        int32_t a[kDepth], b[kDepth], q[kDepth];
        const uint16_t *base[kDepth];
        uint2 v[kDepth];
        bool act[kDepth];
        auto request = [&](int g, int pair) {   // pair = 0 .. 31 of this chunk; beyond: an empty run
            const int rr = 2 * pair + half;
            a[g] = __shfl(ra, rr & 63, 64);
            b[g] = __shfl(rb, rr & 63, 64);
            if (pair >= 32) a[g] = b[g] = 0;
            base[g] = low + (size_t)(t0 + c0 + (rr & 63)) * kPartTile;
            q[g] = (a[g] >> 2) + hl;
            act[g] = 4 * q[g] < b[g];
            v[g] = act[g] ? *reinterpret_cast<const uint2 *>(base[g] + 4 * q[g]) : make_uint2(0u, 0u);
        };
#pragma unroll
        for (int g = 0; g < kDepth; ++g) request(g, g);
        for (int p0 = 0; p0 < 32; p0 += kDepth) {
#pragma unroll
            for (int g = 0; g < kDepth; ++g) {
                add4(v[g], q[g], a[g], b[g], act[g]);
                for (;;) {   // runs of more than 128 - 3 rows: on, 32 quads at a time (the same trips for the whole wave)
                    q[g] += 32;
                    const bool more = 4 * q[g] < b[g];
                    if (!__ballot(more)) break;
                    const uint2 w = more ? *reinterpret_cast<const uint2 *>(base[g] + 4 * q[g]) : make_uint2(0u, 0u);
                    add4(w, q[g], a[g], b[g], more);
                }
                request(g, p0 + kDepth + g);
            }
        }
        ra = na;
        rb = nb2;
    }
    __syncthreads();
    const uint64_t idx0 = (uint64_t)d << kPartShift;
    for (uint32_t i = threadIdx.x; i < (1u << kPartShift); i += kBlock) {
        const uint32_t c = s_hist[i];
        if (c && idx0 + i < (uint64_t)pd.range) counters[pd.cnt_off + idx0 + i] = c;   // this workgroup owns the bucket's counters (cleared before)
    }
}

// ---- partial aggregation (q5.dag: HashAggregateExec mode=Partial, the stage BEFORE the hash repartition) ---------------
// After the count pass with "window = pane", the groups of pane p are its non-zero direct-address counters plus the
// live slots of its straggler table.  Both are compacted with the flag-tile machinery over ONE flat index space:
// [0, cnt_total) = the counter arena, [tab0, tab0 + n_panes * cap) = the tables; segment 2p = pane p's counters,
// segment 2p + 1 = its table, so a pane's groups come out contiguous.
__global__ __launch_bounds__(kBlock) void q5_partial_flag_kernel(SegTiles sx, const uint32_t *__restrict__ counters,
                                                                 const uint64_t *__restrict__ tables, int64_t tab0,
                                                                 uint32_t *__restrict__ flag_words, uint32_t *__restrict__ counts) {
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(sx, tile, kFlagTile);
    const int32_t rel0 = flag_rel0();
    uint32_t flags = 0;
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = tr.tile_begin + rel0 + it * 256 + j;
            bool live = false;
            if (i >= tr.lo && i < tr.hi) live = (tr.seg & 1) ? tables[i - tab0] != 0 : counters[i] != 0;
            flags |= (uint32_t)live << (it * 4 + j);
        }
    store_flags_and_counts(flags, tile, flag_words, counts);
}

__global__ __launch_bounds__(kBlock) void q5_partial_emit_kernel(SegTiles sx, const uint32_t *__restrict__ flag_words,
                                                                 const uint32_t *__restrict__ counts,
                                                                 const uint64_t *__restrict__ tile_base,
                                                                 const uint32_t *__restrict__ counters,
                                                                 const uint64_t *__restrict__ tables, int64_t tab0,
                                                                 const PaneDesc *__restrict__ panes, int32_t *__restrict__ out_key,
                                                                 uint32_t *__restrict__ out_count) {
    __shared__ uint16_t s_list[kFlagTile];
    const int32_t tile = (int32_t)blockIdx.x;
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)tile * kWavesPerBlock);
    if (wc.x + wc.y + wc.z + wc.w == 0) return;
    const uint32_t total = build_flag_list(flag_words[(size_t)tile * kBlock + threadIdx.x], wc, s_list);
    __syncthreads();
    const TileRange tr = locate_tile(sx, tile, kFlagTile);
    const uint64_t base = tile_base[tile];
    if (tr.seg & 1) {
        for (uint32_t i = threadIdx.x; i < total; i += kBlock) {
            const uint64_t e = tables[tr.tile_begin + s_list[i] - tab0];
            out_key[base + i] = (int32_t)(uint32_t)(e >> 32);
            out_count[base + i] = (uint32_t)e;
        }
    } else {
        const PaneDesc pd = panes[tr.seg >> 1];
        for (uint32_t i = threadIdx.x; i < total; i += kBlock) {
            const int64_t idx = tr.tile_begin + s_list[i];
            out_key[base + i] = (int32_t)(pd.base + (idx - (int64_t)pd.cnt_off));
            out_count[base + i] = counters[idx];
        }
    }
}

}  // namespace



// The three entry points share one driver:
//   hot items          : rows = bids, weight = nullptr, `out` set
//   weighted hot items : rows = partial groups (auction, count) received from the other partitions, `out` set
//   partial counts     : window = pane, `part` set: count pass, then the groups of every pane are compacted out
static int q5_run(flockgpu_ctx *ctx, const int32_t *auction, const uint32_t *weight, int64_t rows, const flockgpu_windows *win,
                  flockgpu_q5_result *out, flockgpu_q5_partial_result *part) {
    FG_TRY(check_windows(ctx, win, rows, "q5"));
    if (rows > 0 && !auction) return fail(ctx, FLOCKGPU_ERR_INVALID, "q5: null auction column");
    if ((reinterpret_cast<uintptr_t>(auction) & 15) || (reinterpret_cast<uintptr_t>(weight) & 15))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q5: auction / count columns must be 16-byte aligned");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_win = win->n_windows, n_panes = win->n_panes;
    static const bool plain_clear = exp_env("FLOCKGPU_Q5_PLAIN_CLEAR") != nullptr;   // (experiment knob: cached stores in the clear, as in round 1)

    // pane -> windows CSR, window row counts
    std::vector<int32_t> ptr(n_panes + 1, 0), idx;
    int64_t max_win_rows = 0, covered_rows = 0;
    int max_win_panes = 0;
    for (int w = 0; w < n_win; ++w) {
        for (int p = win->win_pane_lo[w]; p < win->win_pane_hi[w]; ++p) ++ptr[p + 1];
        const int64_t rows = win->pane_row_offsets[win->win_pane_hi[w]] - win->pane_row_offsets[win->win_pane_lo[w]];
        max_win_rows = std::max(max_win_rows, rows);
        max_win_panes = std::max(max_win_panes, win->win_pane_hi[w] - win->win_pane_lo[w]);
    }
    if (max_win_rows >= (int64_t(1) << 32))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q5: a window holds >= 2^32 rows (32-bit counters)");
    for (int p = 0; p < n_panes; ++p) {
        if (ptr[p + 1]) covered_rows += win->pane_row_offsets[p + 1] - win->pane_row_offsets[p];
        ptr[p + 1] += ptr[p];
    }
    // roles of a pane under windows of one or two consecutive panes: wa = the window it is the SECOND pane of, wb = the window it is the
    // FIRST pane of (-1: none).  Any other schedule (longer windows, two windows in one role) keeps the window-walking passes.
    std::vector<int32_t> role_a((size_t)n_panes + 1, -1), role_b((size_t)n_panes + 1, -1);
    bool hop2 = n_win > 0 && n_panes > 0;
    for (int w = 0; w < n_win && hop2; ++w) {
        const int lo = win->win_pane_lo[w], np = win->win_pane_hi[w] - lo;
        if (np < 1 || np > 2 || role_b[(size_t)lo] >= 0 || (np == 2 && role_a[(size_t)lo + 1] >= 0)) hop2 = false;
        else {
            role_b[(size_t)lo] = w;
            if (np == 2) role_a[(size_t)lo + 1] = w;
        }
    }
    idx.resize(ptr[n_panes]);
    {
        std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
        for (int w = 0; w < n_win; ++w)
            for (int p = win->win_pane_lo[w]; p < win->win_pane_hi[w]; ++p) idx[fill[p]++] = w;
    }
    std::vector<int64_t> sb(n_panes), se(n_panes);
    for (int p = 0; p < n_panes; ++p) {
        sb[p] = win->pane_row_offsets[p];
        se[p] = win->pane_row_offsets[p + 1];
    }
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, "q5", sb.data(), se.data(), n_panes, kQ5Tile, &st));

    int32_t *d_ptr = nullptr, *d_idx = nullptr, *h_ptr = nullptr, *h_idx = nullptr;
    FG_TRY(arena_get_t(ctx, "q5.pane_win_ptr", ptr.size(), &d_ptr));
    FG_TRY(arena_get_t(ctx, "q5.pane_win_idx", idx.size() + 1, &d_idx));
    FG_TRY(pinned_get_t(ctx, "q5.pane_win_ptr", ptr.size(), &h_ptr));
    FG_TRY(pinned_get_t(ctx, "q5.pane_win_idx", idx.size() + 1, &h_idx));
    int32_t *d_roles = nullptr, *h_roles = nullptr;   // role_a (n_panes + 1) | role_b (n_panes + 1)
    FG_TRY(arena_get_t(ctx, "q5.pane_roles", 2 * role_a.size(), &d_roles));
    FG_TRY(pinned_get_t(ctx, "q5.pane_roles", 2 * role_a.size(), &h_roles));
    {   // the pane -> windows CSR is uploaded only when it (or its buffers) changed: a stream of equal batches re-submits the same schedule
        std::vector<int64_t> &sig = ctx->host_i64["q5.csr_sig"];
        std::vector<int64_t> now;
        now.reserve(ptr.size() + idx.size() + 3);
        now.push_back((int64_t)reinterpret_cast<uintptr_t>(d_ptr));
        now.push_back((int64_t)reinterpret_cast<uintptr_t>(d_idx));
        now.push_back((int64_t)reinterpret_cast<uintptr_t>(d_roles));
        now.insert(now.end(), ptr.begin(), ptr.end());
        now.insert(now.end(), idx.begin(), idx.end());
        if (sig != now) {
            std::copy(ptr.begin(), ptr.end(), h_ptr);
            std::copy(idx.begin(), idx.end(), h_idx);
            std::copy(role_a.begin(), role_a.end(), h_roles);
            std::copy(role_b.begin(), role_b.end(), h_roles + role_a.size());
            FG_HIP(ctx, hipMemcpyAsync(d_ptr, h_ptr, ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            if (!idx.empty())
                FG_HIP(ctx, hipMemcpyAsync(d_idx, h_idx, idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            FG_HIP(ctx, hipMemcpyAsync(d_roles, h_roles, 2 * role_a.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            sig = now;
        }
    }

    // ---- key-range estimate per pane -> direct-address range per pane, union range per window
    int32_t *d_rng = nullptr, *h_rng = nullptr;  // per (pane, sampling block): {min, max}
    const size_t n_rng = (size_t)2 * kRangeBlocks * std::max(n_panes, 1);
    FG_TRY(arena_get_t(ctx, "q5.pane_range", n_rng + 2, &d_rng));
    FG_TRY(pinned_get_t(ctx, "q5.pane_range", n_rng + 2, &h_rng));
    std::vector<PaneDesc> panes((size_t)std::max(n_panes, 1));
    std::vector<WinDesc> wins((size_t)std::max(n_win, 1));
    for (auto &p : panes) p = PaneDesc{0, 0, 0, 0};
    for (int w = 0; w < n_win; ++w) wins[w] = WinDesc{0, 0, win->win_pane_lo[w], win->win_pane_hi[w], 0};
    uint64_t cnt_total = 0, scan_total = 0;
    bool dense = st.n_tiles > 0 && n_win > 0 && max_win_panes <= kMaxWinPanes;
    // sizes of the previous call of this ctx: {counters, window-range total, usable}.  A stream of equal batches needs about as
    // many counters as the batch before, so their arena is sized from it and the layout itself is made on the device.
    std::vector<int64_t> &hint = ctx->host_i64[weight ? "q5.layout_hint.weighted" : "q5.layout_hint"];
    if (hint.size() != 3) hint.assign(3, 0);
    // affordable = counters + their scans cost no more than a few passes over the input
    const uint64_t budget = std::max<uint64_t>(uint64_t(256) << 20, (uint64_t)covered_rows * 4 * 2);
    if (dense) {
        LaunchScope ls(ctx, "q5_range_kernel");
        hipLaunchKernelGGL(q5_range_kernel, dim3(kRangeBlocks, (unsigned)n_panes), dim3(kBlock), 0, ctx->stream, auction, rows, st.seg_off, d_rng);
    }
    FG_TRY(check_launch(ctx, "q5_range_kernel"));
    // "wide" mode (partition by key range, then count in LDS -- see q5_part_count_kernel): entered when the previous call of this ctx
    // found most tiles wider than the fast kernel's histogram, left when a sample of the partition's tiles is narrow again
    std::vector<int64_t> &wide_hint = ctx->host_i64["q5.wide_hint"];
    if (wide_hint.size() != 1) wide_hint.assign(1, 0);
    static const bool no_wide = exp_env("FLOCKGPU_Q5_NO_WIDE") != nullptr;   // (A/B knob)
    bool wide_mode = wide_hint[0] != 0 && !weight && !part && !no_wide && dense;
    bool wide_ran = false;   // wide mode was asked for AND feasible for this input (digit count within limits)
    bool speculate = dense && !part && hint[2] && hint[0] > 0 && !wide_mode;   // (wide mode sizes its digit count from the host-side layout)
    auto host_layout = [&]() -> int {   // the same rules on the host: first call of a ctx, the Partial stage, a declined speculation
        FG_HIP(ctx, hipMemcpyAsync(h_rng, d_rng, sizeof(int32_t) * n_rng, hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        cnt_total = scan_total = 0;
        for (int p = 0; p < n_panes && dense; ++p) {
            if (ptr[p + 1] == ptr[p] || se[p] <= sb[p]) continue;  // unused / empty pane
            int64_t lo = INT64_MAX, hi = INT64_MIN;
            for (int b = 0; b < kRangeBlocks; ++b) {
                lo = std::min<int64_t>(lo, h_rng[((size_t)p * kRangeBlocks + b) * 2]);
                hi = std::max<int64_t>(hi, h_rng[((size_t)p * kRangeBlocks + b) * 2 + 1]);
            }
            if (lo > hi) continue;
            int64_t base = 0, range = 0;
            if (!q5_pane_layout(lo, hi, &base, &range)) { dense = false; break; }
            panes[p] = PaneDesc{base, cnt_total, (uint32_t)range, 0};
            cnt_total += (uint64_t)range;
        }
        for (int w = 0; w < n_win && dense; ++w) {
            int64_t lo = INT64_MAX, hi = INT64_MIN;
            for (int p = win->win_pane_lo[w]; p < win->win_pane_hi[w]; ++p)
                if (panes[p].range) {
                    lo = std::min(lo, panes[p].base);
                    hi = std::max(hi, panes[p].base + (int64_t)panes[p].range);
                }
            if (lo >= hi) continue;
            if (hi - lo >= (int64_t(1) << 31)) { dense = false; break; }
            wins[w].base = lo;
            wins[w].range = (uint32_t)(hi - lo);
            scan_total += (uint64_t)(hi - lo);
        }
        if (dense && (cnt_total * 4 > budget || scan_total * 4 > 2 * budget)) dense = false;
        return FLOCKGPU_OK;
    };
    if (dense && !speculate) FG_TRY(host_layout());
    PaneDesc *d_panes = nullptr, *h_panes = nullptr;
    WinDesc *d_wins = nullptr, *h_wins = nullptr;
    FG_TRY(arena_get_t(ctx, "q5.panes", panes.size(), &d_panes));
    FG_TRY(pinned_get_t(ctx, "q5.panes", panes.size(), &h_panes));
    FG_TRY(arena_get_t(ctx, "q5.wins", wins.size(), &d_wins));
    FG_TRY(pinned_get_t(ctx, "q5.wins", wins.size(), &h_wins));
    uint64_t *d_info = nullptr, *h_info = nullptr;  // the device layout's {counters, window-range total, verdict}
    FG_TRY(arena_get_t(ctx, "q5.layout_info", 4, &d_info));
    FG_TRY(pinned_get_t(ctx, "q5.layout_info", 4, &h_info));
    auto upload_layout = [&]() -> int {
        if (!dense) {
            cnt_total = scan_total = 0;
            for (auto &p : panes) p = PaneDesc{0, 0, 0, 0};
            for (int w = 0; w < n_win; ++w) wins[w] = WinDesc{0, 0, win->win_pane_lo[w], win->win_pane_hi[w], 0};
        }
        std::copy(panes.begin(), panes.end(), h_panes);
        std::copy(wins.begin(), wins.end(), h_wins);
        ctx->host_i64["q5.wins_sig"].clear();   // the windows' buffer is rewritten here: the speculating path uploads its ranges again
        FG_HIP(ctx, hipMemcpyAsync(d_panes, h_panes, panes.size() * sizeof(PaneDesc), hipMemcpyHostToDevice, ctx->stream));
        FG_HIP(ctx, hipMemcpyAsync(d_wins, h_wins, wins.size() * sizeof(WinDesc), hipMemcpyHostToDevice, ctx->stream));
        return FLOCKGPU_OK;
    };
    uint32_t *counters = nullptr;
    uint64_t capacity = 0;
    // Clean-up after use: a speculating call zeroes the counters it dirtied AFTER its results are back (the kernel runs while the host
    // hands the results on and comes back with the next batch: ~0.1 ms of turnaround per call in which the GPU has nothing else to do),
    // so the next call's clear pass skips them -- 0.045 ms of the 1.1 ms of kernels per 1e9 bids.  The record {arena pointer, words
    // known zero} is taken (and voided) here and written again only once the clean-up has been queued: a call that fails in between,
    // a call on the host-layout path or a regrown arena leave it void, and the next call clears everything itself.
    std::vector<int64_t> &preclean = ctx->host_i64["q5.preclean"];
    if (preclean.size() != 2) preclean.assign(2, 0);
    LayoutArgs lay{};
    uint64_t clean_upto = 0;   // (in 32-bit WORDS of the arena, as preclean[1]: whatever the counters' width was when they were zeroed)
    // 16-bit counters (kQ5Counters16): the bid path over windows of one or two panes, until a call's sum check fails on this ctx
    static const bool no_hop2 = exp_env("FLOCKGPU_Q5_WINDOW_SCAN") != nullptr;   // (A/B knob: the window-walking passes)
    static const bool no_c16 = exp_env("FLOCKGPU_Q5_NO_C16") != nullptr;         // (A/B knob: 32-bit counters, as in rounds 1-5)
    const bool c16 = kQ5Counters16 && kQ5Variant != 2 && !no_c16 && !weight && !part && !wide_mode && hop2 && !no_hop2 && ctx->host_i64["q5.no_c16"].empty() &&
                     !exp_env("FLOCKGPU_Q5_COUNT");   // (the experimental count forms keep 32-bit counters)
    auto words_of = [&](uint64_t n_counters) -> uint64_t { return c16 ? (n_counters + 1) / 2 : n_counters; };
    if (speculate) {
        // window pane ranges for the device pass (bases / ranges are filled in there); counters for the previous call's size + 1/8
        FG_TRY(arena_get_t(ctx, "q5.counters", (size_t)words_of((uint64_t)hint[0] + (uint64_t)hint[0] / 8) + 8, &counters));
        capacity = (ctx->arena["q5.counters"].cap / sizeof(uint32_t) - 8) * (c16 ? 2 : 1);
        if (preclean[0] == (int64_t)reinterpret_cast<uintptr_t>(counters)) clean_upto = (uint64_t)preclean[1];
        {   // the windows' pane ranges: uploaded when the schedule or the buffer changed (the layout kernel rewrites base / range, never lo / hi)
            std::vector<int64_t> &wsig = ctx->host_i64["q5.wins_sig"];   // (one buffer, one record of what it holds)
            std::vector<int64_t> wnow;
            wnow.reserve((size_t)2 * n_win + 2);
            wnow.push_back((int64_t)reinterpret_cast<uintptr_t>(d_wins));
            for (int w = 0; w < n_win; ++w) {
                wnow.push_back(win->win_pane_lo[w]);
                wnow.push_back(win->win_pane_hi[w]);
            }
            if (wsig != wnow) {
                std::copy(wins.begin(), wins.end(), h_wins);
                FG_HIP(ctx, hipMemcpyAsync(d_wins, h_wins, wins.size() * sizeof(WinDesc), hipMemcpyHostToDevice, ctx->stream));
                wsig = wnow;
            }
        }
        lay = LayoutArgs{d_rng, d_ptr, st.seg_off, n_panes, n_win, capacity, budget, d_panes, d_wins, d_info};   // (runs as workgroup 0 of the clear below)
        cnt_total = (uint64_t)hint[0];   // (provisional: sizes launches; the true values come back with the results)
        scan_total = (uint64_t)hint[1];
    } else {
        FG_TRY(upload_layout());
        FG_TRY(arena_get_t(ctx, "q5.counters", (size_t)words_of(cnt_total) + 8, &counters));
    }
    preclean[0] = preclean[1] = 0;

    // device scalars: [0, n_win) win_max, [n_win, 2 n_win) win_groups, then cursor + err (2 x u32), then tab_used (u32 x n_win)
    const size_t n_meta = (size_t)2 * n_win + 1 + ((size_t)n_win + 1) / 2 + 1;
    const size_t n_meta_all = n_meta + (size_t)2 * std::max(n_panes, 0);   // ... then, for 16-bit counters, per pane: counter sum, rows sent to the tables (u64 each)
    uint64_t *d_meta = nullptr, *h_meta = nullptr;
    FG_TRY(arena_get_t(ctx, "q5.meta", n_meta_all, &d_meta));
    unsigned long long *d_pane_sum = reinterpret_cast<unsigned long long *>(d_meta + n_meta), *d_tab_rows = d_pane_sum + std::max(n_panes, 0);
    FG_TRY(pinned_get_t(ctx, "q5.meta", n_meta, &h_meta));
    uint32_t *d_cursor = reinterpret_cast<uint32_t *>(d_meta + 2 * n_win), *d_err = d_cursor + 1, *d_used = d_cursor + 2;

    unsigned long long *d_wsum = nullptr, *h_wsum = nullptr;  // weighted input: per-pane weight totals
    if (weight && n_panes > 0) {
        FG_TRY(arena_get_t(ctx, "q5.pane_weight", (size_t)n_panes + 1, &d_wsum));
        FG_TRY(pinned_get_t(ctx, "q5.pane_weight", (size_t)n_panes + 1, &h_wsum));
    }
    // hash tables: only stragglers in dense mode; every group otherwise (sized from the density seen last call)
    double rpg = ctx->q5_rows_per_group < 1.0 ? 1.0 : ctx->q5_rows_per_group;
    uint64_t cap64 = dense ? 1024 : std::max<uint64_t>(1024, (uint64_t)((double)max_win_rows / rpg * 2.0) + 64);
    uint32_t out_cap = 1u << 16;
    std::vector<int32_t> h_win, h_key;
    uint32_t n_sel = 0;
    const int32_t *sel_win = nullptr, *sel_key = nullptr;  // set when the winners stay on the device
    bool fin_sorted = false;                               // q5_finish_kernel ordered the winners: result columns and offsets are in place
    const int64_t *fin_off = nullptr;
    const int32_t *fin_auction = nullptr;
    const uint64_t *fin_num = nullptr;
    uint32_t h_slow_count = 0;
    for (int attempt = 0;; ++attempt) {
        if (attempt > 8 || cap64 >= (uint64_t(1) << 31))
            return fail(ctx, FLOCKGPU_ERR_CAPACITY, "q5: hash table capacity %llu still overflows", (unsigned long long)cap64);
        const uint32_t cap = (uint32_t)cap64;
        const uint64_t *spec_info = speculate ? d_info : nullptr;
        uint64_t *tables = nullptr;
        FG_TRY(arena_get_t(ctx, "q5.tables", (size_t)cap * std::max(n_win, 1), &tables));
        int32_t *o_win = nullptr, *o_key = nullptr, *slow_list = nullptr;
        FG_TRY(arena_get_t(ctx, "q5.sel_win", out_cap, &o_win));
        FG_TRY(arena_get_t(ctx, "q5.sel_key", out_cap, &o_key));
        FG_TRY(arena_get_t(ctx, "q5.slow_list", (size_t)st.n_tiles + 2, &slow_list));
        // 64 workgroups per window: max + select measured 0.158 / 0.122 / 0.119 / 0.145 ms with 8 / 32 / 64 / 128
        // (fewer: select cannot skip finely; more: per-workgroup prologue and the per-window atomics)
        const uint64_t per_win = n_win > 0 ? std::max<uint64_t>(cap, scan_total / n_win / 4) : cap;
        const bool pane_walk = hop2 && !no_hop2;
        const uint64_t per_pane = n_panes > 0 ? std::max<uint64_t>(cap, cnt_total / (uint64_t)n_panes / 4) : cap;
        // (32-bit counters: max pass 0.062 / 0.073 / 0.108 ms with 32 / 64 / 128 workgroups per pane; 16-bit: max + select 0.052 / 0.050 / 0.051 / 0.056 / 0.066 ms with 8 / 16 / 24 / 32 / 48)
        static const int hop2_blocks_env = exp_env("FLOCKGPU_Q5_HOP2_BLOCKS") ? atoi(exp_env("FLOCKGPU_Q5_HOP2_BLOCKS")) : 0;
        const int hop2_blocks = hop2_blocks_env ? hop2_blocks_env : c16 ? 16 : 32;
        const unsigned gx = (unsigned)std::min<int64_t>(std::max<int64_t>(div_up((int64_t)(pane_walk ? per_pane : per_win), kBlock * 2), 1), pane_walk ? hop2_blocks : 64);
        uint32_t *block_max = nullptr;
        const uint64_t n_block_max = (uint64_t)gx * std::max(n_win, 1) * (pane_walk ? 2 : 1);
        FG_TRY(arena_get_t(ctx, "q5.block_max", (size_t)n_block_max, &block_max));
        {   // one clear for everything this attempt writes into
            const uint64_t clear_words = std::max<uint64_t>({words_of(speculate ? capacity : cnt_total), (uint64_t)cap * n_win, (uint64_t)n_meta_all, n_block_max});
            const unsigned cg = (unsigned)std::max<int64_t>(1, std::min<int64_t>(div_up((int64_t)clear_words / 4 + 1, kBlock), (int64_t)ctx->num_cus * 16));
            {
                LaunchScope ls(ctx, "q5_clear_kernel");
                hipLaunchKernelGGL(q5_clear_kernel, dim3(cg), dim3(kBlock), 0, ctx->stream, counters, speculate ? capacity : cnt_total,
                                   (attempt == 0 && speculate) ? (clean_upto & ~uint64_t(3)) : uint64_t(0), tables, (uint64_t)cap * n_win, d_meta, (uint64_t)n_meta_all,
                                   slow_list, block_max, n_block_max, plain_clear ? 1 : 0, c16 ? 1 : 0, speculate ? lay : LayoutArgs{});
                if (speculate) clean_upto = std::max(clean_upto, words_of(capacity));   // (the arena is zero up to there now; only [0, counters in use) gets dirty)
            }
            FG_TRY(check_launch(ctx, "q5_clear_kernel"));
        }
        if (d_wsum) FG_HIP(ctx, hipMemsetAsync(d_wsum, 0, sizeof(unsigned long long) * (size_t)n_panes, ctx->stream));
        uint32_t nd = 0;
        if (wide_mode && dense) {
            uint32_t max_range = 0;
            for (int p = 0; p < n_panes; ++p) max_range = std::max(max_range, panes[(size_t)p].range);
            nd = (uint32_t)div_up((int64_t)max_range, int64_t(1) << kPartShift) + 1;
        }
        const bool wide = wide_mode && dense && nd >= 2 && nd <= (uint32_t)kPartMaxDigits && st.n_tiles > 0 && n_win > 0;
        wide_ran = wide;
        uint32_t *d_sample = nullptr, *h_sample = nullptr;
        FG_TRY(arena_get_t(ctx, "q5.part_sample", 4, &d_sample));
        FG_TRY(pinned_get_t(ctx, "q5.part_sample", 4, &h_sample));
        h_sample[0] = h_sample[1] = 0;
        if (wide) {
            SegTiles st4;
            FG_TRY(build_seg_tiles(ctx, "q5.part", sb.data(), se.data(), n_panes, kPartTile, &st4));
            const int64_t slots = (int64_t)st4.n_tiles * (nd + 1);
            if (slots >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q5: too many (pane, digit, tile) slots for the partition pass");
            int32_t *off_mat = nullptr, *part_keys = nullptr;
            uint16_t *part_low = nullptr;
            FG_TRY(arena_get_t(ctx, "q5.part_hist", (size_t)slots + 4, &off_mat));
            FG_TRY(arena_get_t(ctx, "q5.part_keys", (size_t)st4.n_tiles * kPartTile + 4, &part_keys));   // (whole keys of the straggler buckets only: sparse use)
            FG_TRY(arena_get_t(ctx, "q5.part_low", (size_t)st4.n_tiles * kPartTile + 8, &part_low));
            FG_HIP(ctx, hipMemsetAsync(d_sample, 0, 2 * sizeof(uint32_t), ctx->stream));
            {
                LaunchScope ls(ctx, "q5_part_tile_kernel");
                hipLaunchKernelGGL(q5_part_tile_kernel, dim3((unsigned)st4.n_tiles), dim3(kBlock), 0, ctx->stream, auction, st4, d_panes, d_ptr, nd, off_mat, part_keys,
                                   part_low, d_sample);
            }
            FG_TRY(check_launch(ctx, "q5_part_tile_kernel"));
            {
                LaunchScope ls(ctx, "q5_bucket_count_kernel");
                hipLaunchKernelGGL(q5_bucket_count_kernel, dim3(nd, (unsigned)n_panes), dim3(kBlock), 0, ctx->stream, part_keys, part_low, st4, d_panes, d_ptr, d_idx, nd,
                                   off_mat, counters, tables, cap, d_used, d_err);
            }
            FG_TRY(check_launch(ctx, "q5_bucket_count_kernel"));
            FG_HIP(ctx, hipMemcpyAsync(h_sample, d_sample, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        } else if (st.n_tiles > 0 && n_win > 0) {
#ifdef FLOCKGPU_EXPERIMENTAL
            // (A/B knob, experimental builds: FLOCKGPU_Q5_COUNT = wg | wave1 | wave2 | wave4 | wavep1 | wavep4 | wgp, FLOCKGPU_Q5_WAVES_PER_CU)
            static const char *count_form_env = exp_env("FLOCKGPU_Q5_COUNT");
            const int wave_form = weight ? 0 : count_form_env ? (!strcmp(count_form_env, "wave1") ? 1 : !strcmp(count_form_env, "wave2") ? 2 : !strcmp(count_form_env, "wave4") ? 4 :
                                                                   !strcmp(count_form_env, "wavep1") ? 101 : !strcmp(count_form_env, "wavep4") ? 104 : !strcmp(count_form_env, "wgp") ? 200 : 0) : kQ5WaveForm;
            if (wave_form == 200) {
                LaunchScope ls(ctx, "q5_count_kernel");
                static const int per_cu = exp_env("FLOCKGPU_Q5_WAVES_PER_CU") ? atoi(exp_env("FLOCKGPU_Q5_WAVES_PER_CU")) : 5;   // (workgroups per CU here)
                const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>(st.n_tiles, (int64_t)ctx->num_cus * per_cu));
                hipLaunchKernelGGL(q5_count_wgp_kernel, dim3(g), dim3(kBlock), 0, ctx->stream, auction, st.tiles, st.n_tiles, d_panes, d_ptr, d_idx, counters, tables, cap, d_used, d_err, slow_list, spec_info);
            } else if (wave_form > 100) {
                LaunchScope ls(ctx, "q5_count_kernel");
                const int kw = wave_form - 100;
                static const int per_cu = exp_env("FLOCKGPU_Q5_WAVES_PER_CU") ? atoi(exp_env("FLOCKGPU_Q5_WAVES_PER_CU")) : kQ5WavesPerCu;
                const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>(div_up((int64_t)st.n_tiles, kw), (int64_t)ctx->num_cus * per_cu / kw));
                if (kw == 1)
                    hipLaunchKernelGGL(q5_count_wavep_kernel<1>, dim3(g), dim3(64), 0, ctx->stream, auction, st.tiles, st.n_tiles, d_panes, d_ptr, d_idx, counters, tables, cap, d_used, d_err, slow_list, spec_info);
                else
                    hipLaunchKernelGGL(q5_count_wavep_kernel<4>, dim3(g), dim3(256), 0, ctx->stream, auction, st.tiles, st.n_tiles, d_panes, d_ptr, d_idx, counters, tables, cap, d_used, d_err, slow_list, spec_info);
            } else if (wave_form) {
                LaunchScope ls(ctx, "q5_count_kernel");
                const unsigned g = (unsigned)div_up((int64_t)st.n_tiles, wave_form);
                if (wave_form == 1)
                    hipLaunchKernelGGL(q5_count_wave_kernel<1>, dim3(g), dim3(64), 0, ctx->stream, auction, st, d_panes, d_ptr, d_idx, counters, tables, cap, d_used, d_err, slow_list, spec_info);
                else if (wave_form == 2)
                    hipLaunchKernelGGL(q5_count_wave_kernel<2>, dim3(g), dim3(128), 0, ctx->stream, auction, st, d_panes, d_ptr, d_idx, counters, tables, cap, d_used, d_err, slow_list, spec_info);
                else
                    hipLaunchKernelGGL(q5_count_wave_kernel<4>, dim3(g), dim3(256), 0, ctx->stream, auction, st, d_panes, d_ptr, d_idx, counters, tables, cap, d_used, d_err, slow_list, spec_info);
            } else
#endif
            {
                LaunchScope ls(ctx, "q5_count_kernel");
#if defined(FLOCKGPU_EXPERIMENTAL) && defined(FLOCKGPU_AB_Q5_PERSIST)
                const unsigned count_grid = (unsigned)std::min<int64_t>(st.n_tiles, (int64_t)ctx->num_cus * (exp_env("FLOCKGPU_Q5_PERSIST_PER_CU") ? atoi(exp_env("FLOCKGPU_Q5_PERSIST_PER_CU")) : 8));
#else
                const unsigned count_grid = (unsigned)st.n_tiles;
#endif
                hipLaunchKernelGGL(weight ? q5_count_kernel<true> : q5_count_kernel<false>, dim3(count_grid), dim3(kBlock), 0,
                                   ctx->stream, auction, weight, st, d_panes, d_ptr, d_idx, counters, tables, cap, d_used, d_err,
                                   slow_list, spec_info, d_wsum, c16 ? 1 : 0, c16 ? d_tab_rows : (unsigned long long *)nullptr);
            }
            FG_TRY(check_launch(ctx, "q5_count_kernel"));
            if (d_wsum) FG_HIP(ctx, hipMemcpyAsync(h_wsum, d_wsum, sizeof(unsigned long long) * (size_t)n_panes, hipMemcpyDeviceToHost, ctx->stream));
            {
                LaunchScope ls(ctx, "q5_count_slow_kernel");
                const unsigned gs = (unsigned)std::min<int64_t>(st.n_tiles, (int64_t)ctx->num_cus * 8);
                hipLaunchKernelGGL(weight ? q5_count_slow_kernel<true> : q5_count_slow_kernel<false>, dim3(gs), dim3(kBlock), 0,
                                   ctx->stream, auction, weight, st, d_panes, d_ptr, d_idx, counters, tables, cap, d_used, d_err,
                                   slow_list, spec_info, c16 ? 1 : 0, c16 ? d_tab_rows : (unsigned long long *)nullptr);
            }
            FG_TRY(check_launch(ctx, "q5_count_slow_kernel"));
        }
        if (part) {  // Partial stage: hand out the groups of every pane (window w == pane w)
            const int64_t tab0 = ((int64_t)cnt_total + 3) & ~int64_t(3);
            std::vector<int64_t> xb((size_t)2 * n_win), xe((size_t)2 * n_win);
            for (int w = 0; w < n_win; ++w) {
                xb[2 * w] = (int64_t)panes[w].cnt_off;
                xe[2 * w] = (int64_t)panes[w].cnt_off + (int64_t)panes[w].range;
                xb[2 * w + 1] = tab0 + (int64_t)w * cap;
                xe[2 * w + 1] = xb[2 * w + 1] + cap;
            }
            SegTiles sx;
            FG_TRY(build_seg_tiles(ctx, "q5.partial", xb.data(), xe.data(), 2 * n_win, kFlagTile, &sx));
            uint32_t *x_flags = nullptr, *x_counts = nullptr;
            uint64_t *x_base = nullptr;
            int64_t *d_xoff = nullptr, *h_xoff = nullptr;
            FG_TRY(arena_get_t(ctx, "q5.partial_flags", (size_t)sx.n_tiles * kBlock + 4, &x_flags));
            FG_TRY(arena_get_t(ctx, "q5.partial_counts", (size_t)sx.n_tiles * kWavesPerBlock + 4, &x_counts));
            FG_TRY(arena_get_t(ctx, "q5.partial_base", (size_t)sx.n_tiles + 1, &x_base));
            FG_TRY(arena_get_t(ctx, "q5.partial_off", (size_t)2 * n_win + 1, &d_xoff));
            FG_TRY(pinned_get_t(ctx, "q5.partial_off", (size_t)2 * n_win + 1, &h_xoff));
            if (sx.n_tiles > 0) {
                LaunchScope ls(ctx, "q5_partial_flag_kernel");
                hipLaunchKernelGGL(q5_partial_flag_kernel, dim3((unsigned)sx.n_tiles), dim3(kBlock), 0, ctx->stream, sx, counters, tables,
                                   tab0, x_flags, x_counts);
            }
            FG_TRY(check_launch(ctx, "q5_partial_flag_kernel"));
            FG_TRY(launch_tile_scan(ctx, x_counts, sx.n_tiles, x_base, sx.tile_first, sx.n_seg, d_xoff));
            FG_HIP(ctx, hipMemcpyAsync(h_xoff, d_xoff, sizeof(int64_t) * ((size_t)2 * n_win + 1), hipMemcpyDeviceToHost, ctx->stream));
            FG_HIP(ctx, hipMemcpyAsync(h_meta, d_meta, sizeof(uint64_t) * n_meta, hipMemcpyDeviceToHost, ctx->stream));
            FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (reinterpret_cast<const uint32_t *>(h_meta + 2 * n_win)[1]) {  // a pane's straggler table filled up
                cap64 *= 4;
                continue;
            }
            const int64_t n_out = h_xoff[2 * n_win];
            if (n_out >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q5 partial: more than 2^31 groups");
            int32_t *p_key = nullptr;
            uint32_t *p_cnt = nullptr;
            FG_TRY(arena_get_t(ctx, "q5.partial_key", (size_t)n_out + 4, &p_key));
            FG_TRY(arena_get_t(ctx, "q5.partial_cnt", (size_t)n_out + 4, &p_cnt));
            if (sx.n_tiles > 0 && n_out > 0) {
                LaunchScope ls(ctx, "q5_partial_emit_kernel");
                hipLaunchKernelGGL(q5_partial_emit_kernel, dim3((unsigned)sx.n_tiles), dim3(kBlock), 0, ctx->stream, sx, x_flags, x_counts,
                                   x_base, counters, tables, tab0, d_panes, p_key, p_cnt);
            }
            FG_TRY(check_launch(ctx, "q5_partial_emit_kernel"));
            std::vector<int64_t> &poffs = ctx->host_i64["q5.partial_pane_offsets"];
            poffs.resize((size_t)n_win + 1);
            for (int w = 0; w <= n_win; ++w) poffs[w] = h_xoff[2 * w];
            part->auction = p_key;
            part->count = p_cnt;
            part->pane_out_offsets = poffs.data();
            part->rows = n_out;
            return FLOCKGPU_OK;
        }
        if (n_win > 0 && pane_walk) {
            {
                LaunchScope ls(ctx, "q5_max_kernel");
                hipLaunchKernelGGL((c16 ? q5_hop2_scan_kernel<false, true> : q5_hop2_scan_kernel<false, false>), dim3(gx, (unsigned)n_panes), dim3(kBlock), 0, ctx->stream, d_panes, d_roles,
                                   d_roles + role_a.size(), counters, tables, cap, d_used, d_meta, d_meta + n_win, block_max, d_cursor, out_cap, o_win, o_key, spec_info, d_pane_sum);
            }
            FG_TRY(check_launch(ctx, "q5_max_kernel"));
            {
                LaunchScope ls(ctx, "q5_select_kernel");
                hipLaunchKernelGGL((c16 ? q5_hop2_scan_kernel<true, true> : q5_hop2_scan_kernel<true, false>), dim3(gx, (unsigned)n_panes), dim3(kBlock), 0, ctx->stream, d_panes, d_roles,
                                   d_roles + role_a.size(), counters, tables, cap, d_used, d_meta, d_meta + n_win, block_max, d_cursor, out_cap, o_win, o_key, spec_info, d_pane_sum);
            }
            FG_TRY(check_launch(ctx, "q5_select_kernel"));
        } else if (n_win > 0) {
            {
                LaunchScope ls(ctx, "q5_max_kernel");
                hipLaunchKernelGGL(q5_scan_kernel<false>, dim3(gx, (unsigned)n_win), dim3(kBlock), 0, ctx->stream, d_wins,
                                   d_panes, counters, tables, cap, d_used, d_meta, d_meta + n_win, block_max, d_cursor, out_cap,
                                   o_win, o_key, spec_info);
            }
            FG_TRY(check_launch(ctx, "q5_max_kernel"));
            {
                LaunchScope ls(ctx, "q5_select_kernel");
                hipLaunchKernelGGL(q5_scan_kernel<true>, dim3(gx, (unsigned)n_win), dim3(kBlock), 0, ctx->stream, d_wins,
                                   d_panes, counters, tables, cap, d_used, d_meta, d_meta + n_win, block_max, d_cursor, out_cap,
                                   o_win, o_key, spec_info);
            }
            FG_TRY(check_launch(ctx, "q5_select_kernel"));
        }
        // one synchronisation for the scalars AND the winners, and one kernel instead of copy nodes: q5_finish_kernel leaves the
        // scalars, the layout's verdict, the ordered winners' offsets in ONE pinned block and the result columns on the device
        constexpr uint32_t kSpecWinners = 4096;
        uint64_t *h_fin = nullptr;
        int32_t *fin_a = nullptr;
        uint64_t *fin_n = nullptr;
        FG_TRY(pinned_get_t(ctx, "q5.finish", n_meta + 8 + (size_t)n_win + 2, &h_fin));
        FG_TRY(arena_get_t(ctx, "q5.out_auction", (size_t)kFinishMax + 1, &fin_a));
        FG_TRY(arena_get_t(ctx, "q5.out_num", (size_t)kFinishMax + 1, &fin_n));
        {
            LaunchScope ls(ctx, "q5_finish_kernel");
            hipLaunchKernelGGL(q5_finish_kernel, dim3(1), dim3(kFinishThreads), 0, ctx->stream, d_meta, (uint32_t)n_meta, n_win, spec_info, slow_list, o_win, o_key, out_cap,
                               fin_a, fin_n, h_fin, c16 && pane_walk ? d_pane_sum : (const unsigned long long *)nullptr, d_tab_rows, st.seg_off, d_ptr, n_panes);
        }
        FG_TRY(check_launch(ctx, "q5_finish_kernel"));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::copy(h_fin, h_fin + n_meta, h_meta);
        if (speculate) std::copy(h_fin + n_meta, h_fin + n_meta + 3, h_info);
        h_slow_count = (uint32_t)h_fin[n_meta + 3];
        if (h_fin[n_meta + 3] >> 32) {   // a pane's 16-bit counters do not add up to its rows: some count outgrew 65535 -- 32-bit counters for this ctx from here on
            ctx->host_i64["q5.no_c16"].assign(1, 1);
            return q5_run(ctx, auction, weight, rows, win, out, part);   // (preclean is void by now: the repeat clears everything itself)
        }
        fin_sorted = h_fin[n_meta + 4] != 0;
        fin_off = reinterpret_cast<const int64_t *>(h_fin + n_meta + 5);
        if (speculate) {
            if (!h_info[2]) {   // declined (ranges moved beyond the counters the previous call sized, or not dense any more): the slow way
                speculate = false;
                hint[2] = 0;
                FG_TRY(host_layout());
                FG_TRY(upload_layout());
                FG_TRY(arena_get_t(ctx, "q5.counters", (size_t)words_of(cnt_total) + 8, &counters));
                cap64 = dense ? 1024 : std::max<uint64_t>(1024, (uint64_t)((double)max_win_rows / rpg * 2.0) + 64);
                --attempt;
                continue;
            }
            cnt_total = h_info[0];
            scan_total = h_info[1];
        }
        if (h_wsum) {   // per-pane weight totals, summed by the count kernel: a window's counts must stay below 2^32 (uint32 counters)
            for (int w = 0; w < n_win; ++w) {
                unsigned long long tot = 0;
                for (int p = win->win_pane_lo[w]; p < win->win_pane_hi[w]; ++p) tot += h_wsum[p];
                if (tot >= (1ull << 32))
                    return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q5 weighted: the counts of window %d add up to %llu >= 2^32 (32-bit counters)", w, tot);
            }
        }
        const uint32_t *tail = reinterpret_cast<const uint32_t *>(h_meta + 2 * n_win);
        n_sel = tail[0];
        if (tail[1]) {  // a window table filled up: the group-count hint was too optimistic
            cap64 *= 4;
            continue;
        }
        if (n_sel > out_cap) {  // many ties: grow the winner buffer and redo (rare)
            out_cap = n_sel + 1024;
            continue;
        }
        h_win.resize(n_sel);
        h_key.resize(n_sel);
        if (fin_sorted) {
            fin_auction = fin_a;
            fin_num = fin_n;
        } else if (n_sel <= kSpecWinners) {   // more winners than the finish kernel orders: copy them back, order them here (below)
            FG_HIP(ctx, hipMemcpyAsync(h_win.data(), o_win, sizeof(int32_t) * n_sel, hipMemcpyDeviceToHost, ctx->stream));
            FG_HIP(ctx, hipMemcpyAsync(h_key.data(), o_key, sizeof(int32_t) * n_sel, hipMemcpyDeviceToHost, ctx->stream));
            FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        } else {  // many ties: ordered on the device below
            sel_win = o_win;
            sel_key = o_key;
        }
        break;
    }
    hint[0] = (int64_t)cnt_total;   // the next call of this kind sizes its counters from this one and lays them out on the device
    hint[1] = (int64_t)scan_total;
    hint[2] = dense ? 1 : 0;
    if (!weight && !part) {
        uint32_t *h_sample = nullptr;
        FG_TRY(pinned_get_t(ctx, "q5.part_sample", 4, &h_sample));
        // wide mode asked for but not feasible for this input (a pane range beyond the digit limit, or under two digits): nothing was
        // sampled, so the leave condition below could never fire and the ctx would stay without its device-side layout speculation
        // for good -- back to the speculating path
        if (wide_mode && !wide_ran) wide_hint[0] = 0;
        else if (wide_mode && h_sample[0]) wide_hint[0] = (uint64_t)h_sample[1] * 4 >= (uint64_t)h_sample[0] * 3 ? 0 : 1;   // three quarters of the sample narrow: back to the fast kernel
        else if (!wide_mode && dense && st.n_tiles > 64) wide_hint[0] = (int64_t)h_slow_count * 4 > (int64_t)st.n_tiles ? 1 : 0;   // tiles the fast kernel declined
    }
    // remember how dense the groups were so the next sparse call sizes its tables right away
    {
        double best = 1e30;
        for (int w = 0; w < n_win; ++w) {
            const int64_t rows = win->pane_row_offsets[win->win_pane_hi[w]] - win->pane_row_offsets[win->win_pane_lo[w]];
            const uint64_t g = h_meta[n_win + w];
            if (g) best = std::min(best, (double)rows / (double)g);
        }
        if (best < 1e30) ctx->q5_rows_per_group = best;
    }
    std::vector<int64_t> &offs = ctx->host_i64["q5.win_out_offsets"];
    std::vector<uint64_t> &wmax = ctx->host_u64["q5.win_max"], &wgrp = ctx->host_u64["q5.win_groups"];
    offs.assign((size_t)n_win + 1, 0);
    wmax.assign(h_meta, h_meta + n_win);
    wgrp.assign(h_meta + n_win, h_meta + 2 * n_win);
    if (fin_sorted) {   // the usual case: the finish kernel left the ordered result columns on the device and the offsets in pinned memory
        offs.assign(fin_off, fin_off + n_win + 1);
        out->auction = fin_auction;
        out->num = fin_num;
        out->win_out_offsets = offs.data();
        out->win_max = wmax.data();
        out->win_groups = wgrp.data();
        out->rows = n_sel;
    } else
    if (sel_key) {
        // Every group of a window can tie for its MAX (equal counts): then the result is as large as the group set, and
        // (window, auction) order comes from two stable radix sorts on the device -- by auction, then by window --
        // instead of a host sort of millions of pairs.
        int32_t *d_mm = nullptr, *h_mm = nullptr, *k1 = nullptr, *w_in = nullptr, *w2 = nullptr, *d_oa = nullptr;
        uint32_t *v1 = nullptr, *v2 = nullptr;
        uint64_t *d_on = nullptr;
        int64_t *d_off = nullptr, *h_off = nullptr;
        FG_TRY(arena_get_t(ctx, "q5.ord_minmax", 4, &d_mm));
        FG_TRY(pinned_get_t(ctx, "q5.ord_minmax", 4, &h_mm));
        FG_TRY(key_min_max(ctx, sel_key, n_sel, d_mm));
        FG_HIP(ctx, hipMemcpyAsync(h_mm, d_mm, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int bits = 1, wbits = 1;
        while (bits < 32 && (((uint64_t)((int64_t)h_mm[1] - (int64_t)h_mm[0])) >> bits)) ++bits;
        while (wbits < 31 && ((uint32_t)n_win >> wbits)) ++wbits;
        FG_TRY(radix_sort_pairs(ctx, "q5.ord_key", sel_key, nullptr, n_sel, h_mm[0], bits, &k1, &v1));
        FG_TRY(arena_get_t(ctx, "q5.ord_win_in", (size_t)n_sel + 4, &w_in));
        FG_TRY(gather_i32(ctx, sel_win, reinterpret_cast<const int32_t *>(v1), n_sel, w_in));
        FG_TRY(radix_sort_pairs(ctx, "q5.ord_win", w_in, v1, n_sel, 0, wbits, &w2, &v2));
        FG_TRY(arena_get_t(ctx, "q5.out_auction", (size_t)n_sel + 1, &d_oa));
        FG_TRY(arena_get_t(ctx, "q5.out_num", (size_t)n_sel + 1, &d_on));
        FG_TRY(arena_get_t(ctx, "q5.ord_off", (size_t)n_win + 2, &d_off));
        FG_TRY(pinned_get_t(ctx, "q5.ord_off", (size_t)n_win + 2, &h_off));
        FG_TRY(gather_i32(ctx, sel_key, reinterpret_cast<const int32_t *>(v2), n_sel, d_oa));
        hipLaunchKernelGGL(q5_winner_num_kernel, dim3((unsigned)div_up((int64_t)n_sel, kBlock)), dim3(kBlock), 0, ctx->stream, w2,
                           d_meta, (int64_t)n_sel, d_on);
        FG_TRY(check_launch(ctx, "q5_winner_num_kernel"));
        FG_TRY(sorted_key_offsets(ctx, w2, n_sel, n_win, d_off));
        FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * ((size_t)n_win + 1), hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        offs.assign(h_off, h_off + n_win + 1);
        out->auction = d_oa;
        out->num = d_on;
        out->win_out_offsets = offs.data();
        out->win_max = wmax.data();
        out->win_groups = wgrp.data();
        out->rows = n_sel;
        return FLOCKGPU_OK;   // (the rare many-ties path leaves the clean-up to the next call)
    }
    if (!fin_sorted) {
    // more winners than the finish kernel orders, fewer than the device sorts pay for: ordered by (window, auction) on the host
    std::vector<uint32_t> order(n_sel);
    for (uint32_t i = 0; i < n_sel; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return h_win[a] != h_win[b] ? h_win[a] < h_win[b] : h_key[a] < h_key[b];
    });
    int32_t *h_oa = nullptr;
    uint64_t *h_on = nullptr;
    FG_TRY(pinned_get_t(ctx, "q5.out_auction", (size_t)n_sel + 1, &h_oa));
    FG_TRY(pinned_get_t(ctx, "q5.out_num", (size_t)n_sel + 1, &h_on));
    for (uint32_t i = 0; i < n_sel; ++i) {
        const uint32_t s = order[i];
        h_oa[i] = h_key[s];
        h_on[i] = wmax[h_win[s]];
        offs[h_win[s] + 1] += 1;
    }
    for (int w = 0; w < n_win; ++w) offs[w + 1] += offs[w];
    int32_t *d_oa = nullptr;
    uint64_t *d_on = nullptr;
    FG_TRY(arena_get_t(ctx, "q5.out_auction", (size_t)n_sel + 1, &d_oa));
    FG_TRY(arena_get_t(ctx, "q5.out_num", (size_t)n_sel + 1, &d_on));
    if (n_sel) {  // stream-ordered uploads from pinned staging (rewritten only after the next call's first synchronisation)
        FG_HIP(ctx, hipMemcpyAsync(d_oa, h_oa, sizeof(int32_t) * n_sel, hipMemcpyHostToDevice, ctx->stream));
        FG_HIP(ctx, hipMemcpyAsync(d_on, h_on, sizeof(uint64_t) * n_sel, hipMemcpyHostToDevice, ctx->stream));
    }
    out->auction = d_oa;
    out->num = d_on;
    out->win_out_offsets = offs.data();
    out->win_max = wmax.data();
    out->win_groups = wgrp.data();
    out->rows = n_sel;
    }
    static const bool no_preclean = exp_env("FLOCKGPU_Q5_NO_PRECLEAN") != nullptr || exp_env("FLOCKGPU_Q5_PLAIN_CLEAR") != nullptr;   // (A/B knobs)
    if (speculate && dense && cnt_total > 0 && !no_preclean) {   // clean up after use (see `preclean` above); the results above are already on their way
        int32_t *slow_list = nullptr;
        FG_TRY(arena_get_t(ctx, "q5.slow_list", (size_t)st.n_tiles + 2, &slow_list));
        {
            LaunchScope ls(ctx, "q5_clear_kernel");
            const unsigned cg = (unsigned)std::max<int64_t>(1, std::min<int64_t>(div_up((int64_t)words_of(cnt_total) / 4 + 1, kBlock), (int64_t)ctx->num_cus * 16));
            hipLaunchKernelGGL(q5_clear_kernel, dim3(cg), dim3(kBlock), 0, ctx->stream, counters, cnt_total, uint64_t(0), (uint64_t *)nullptr,
                               uint64_t(0), (uint64_t *)nullptr, uint64_t(0), slow_list, (uint32_t *)nullptr, uint64_t(0), plain_clear ? 1 : 0, c16 ? 1 : 0, LayoutArgs{});
        }
        FG_TRY(check_launch(ctx, "q5_clear_kernel"));
        preclean[0] = (int64_t)reinterpret_cast<uintptr_t>(counters);
        preclean[1] = (int64_t)std::max<uint64_t>(clean_upto, words_of(cnt_total));
    }
    return FLOCKGPU_OK;
}

namespace flockgpu {

int q5_partial_by_tile(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win, Q5TilePartial *out) {
    *out = Q5TilePartial{};
    FG_TRY(check_windows(ctx, win, bid->rows, "q5 partial"));
    if (bid->rows > 0 && !bid->auction) return fail(ctx, FLOCKGPU_ERR_INVALID, "q5 partial: null auction column");
    if (reinterpret_cast<uintptr_t>(bid->auction) & 15) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q5: auction column must be 16-byte aligned");
    if (bid->rows >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q5 partial: more than 2^31 rows per call");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_panes = win->n_panes;
    std::vector<char> used((size_t)std::max(n_panes, 1), 0);
    for (int w = 0; w < win->n_windows; ++w)
        for (int p = win->win_pane_lo[w]; p < win->win_pane_hi[w]; ++p) used[(size_t)p] = 1;
    std::vector<int64_t> sb((size_t)n_panes), se((size_t)n_panes);
    for (int p = 0; p < n_panes; ++p) {
        sb[(size_t)p] = win->pane_row_offsets[p];
        se[(size_t)p] = win->pane_row_offsets[p + 1];
    }
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, "q5.tile_panes", sb.data(), se.data(), n_panes, kQ5Tile, &st));  // (own name: "q5" holds the FinalPartitioned side's schedule)
    int64_t *d_reg = nullptr, *h_reg = nullptr;
    uint32_t *d_cur = nullptr, *h_cur = nullptr;
    int32_t *o_key = nullptr;
    uint32_t *o_cnt = nullptr;
    FG_TRY(arena_get_t(ctx, "q5.tile_region", (size_t)n_panes + 1, &d_reg));
    FG_TRY(pinned_get_t(ctx, "q5.tile_region", (size_t)n_panes + 1, &h_reg));
    FG_TRY(arena_get_t(ctx, "q5.tile_cursor", (size_t)n_panes + 1, &d_cur));
    FG_TRY(pinned_get_t(ctx, "q5.tile_cursor", (size_t)n_panes + 1, &h_cur));
    FG_TRY(arena_get_t(ctx, "q5.tile_key", (size_t)bid->rows + 4, &o_key));
    FG_TRY(arena_get_t(ctx, "q5.tile_cnt", (size_t)bid->rows + 4, &o_cnt));
    // (the pinned staging was last read under the synchronisation that ended the previous call)
    for (int p = 0; p < n_panes; ++p) h_reg[p] = used[(size_t)p] ? sb[(size_t)p] : -1;
    FG_HIP(ctx, hipMemcpyAsync(d_reg, h_reg, sizeof(int64_t) * (size_t)std::max(n_panes, 1), hipMemcpyHostToDevice, ctx->stream));
    FG_HIP(ctx, hipMemsetAsync(d_cur, 0, sizeof(uint32_t) * ((size_t)n_panes + 1), ctx->stream));
    int64_t max_tiles = 0;
    for (int p = 0; p < n_panes; ++p)
        if (se[(size_t)p] > sb[(size_t)p]) max_tiles = std::max(max_tiles, div_up(se[(size_t)p] - (sb[(size_t)p] & ~int64_t(3)), (int64_t)kQ5Tile));
    if (max_tiles * n_panes >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q5 partial: too many (pane, tile) pairs");
    if (st.n_tiles > 0) {
        LaunchScope ls(ctx, "q5_partial_tile_kernel");
        hipLaunchKernelGGL(q5_partial_tile_kernel, dim3((unsigned)(max_tiles * n_panes)), dim3(kBlock), 0, ctx->stream, bid->auction, st, d_reg, d_cur, o_key,
                           o_cnt);
    }
    FG_TRY(check_launch(ctx, "q5_partial_tile_kernel"));
    FG_HIP(ctx, hipMemcpyAsync(h_cur, d_cur, sizeof(uint32_t) * (size_t)std::max(n_panes, 1), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->auction = o_key;
    out->count = o_cnt;
    out->capacity = bid->rows;
    out->offsets.resize((size_t)2 * n_panes + 1);
    for (int p = 0; p < n_panes; ++p) {
        const int64_t n = used[(size_t)p] ? (int64_t)h_cur[p] : 0;
        if (n > se[(size_t)p] - sb[(size_t)p]) return fail(ctx, FLOCKGPU_ERR_HIP, "q5 partial: pane %d claims %lld pairs for %lld rows", p, (long long)n, (long long)(se[(size_t)p] - sb[(size_t)p]));
        out->offsets[(size_t)2 * p] = sb[(size_t)p];
        out->offsets[(size_t)2 * p + 1] = sb[(size_t)p] + n;
        out->pairs += n;
    }
    out->offsets[(size_t)2 * n_panes] = n_panes ? se[(size_t)n_panes - 1] : 0;
    return FLOCKGPU_OK;
}

}  // namespace flockgpu

extern "C" {

int flockgpu_q5_hot_items(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                          flockgpu_q5_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!bid || !out || bid->rows < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "q5: null argument");
    return q5_run(ctx, bid->auction, nullptr, bid->rows, win, out, nullptr);
}

int flockgpu_q5_hot_items_weighted(flockgpu_ctx *ctx, const int32_t *auction, const uint32_t *count, int64_t rows,
                                   const flockgpu_windows *win, flockgpu_q5_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!out || rows < 0 || (rows > 0 && (!auction || !count))) return fail(ctx, FLOCKGPU_ERR_INVALID, "q5 weighted: null argument");
    return q5_run(ctx, auction, count, rows, win, out, nullptr);  // (rows == 0 with null columns: the plain path over no rows)
}

int flockgpu_q5_partial_counts(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                               flockgpu_q5_partial_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!bid || !out || !win || bid->rows < 0 || win->n_panes < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "q5 partial: null argument");
    std::vector<int32_t> lo((size_t)std::max(win->n_panes, 1)), hi(lo.size());
    for (int p = 0; p < win->n_panes; ++p) {
        lo[p] = p;
        hi[p] = p + 1;
    }
    const flockgpu_windows panes{win->pane_row_offsets, win->n_panes, lo.data(), hi.data(), win->n_panes};
    *out = flockgpu_q5_partial_result{};
    return q5_run(ctx, bid->auction, nullptr, bid->rows, &panes, nullptr, out);
}

}  // extern "C"
