// Device-side building blocks shared by the operator kernels (gfx950, wave64):
//   * segment -> tile mapping (a window never shares a tile with another window)
//   * single-pass chained scan (decoupled look-back) for order-preserving compaction
//   * wave-level ballot / mbcnt ranks
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"

namespace flockgpu {

constexpr int kBlock = 256;           // 4 waves of 64
constexpr int kWavesPerBlock = kBlock / 64;

// ---- segment tiles ---------------------------------------------------------------------------
// Rows of segment s are [seg_off[2s], seg_off[2s+1]).  Its tiles start at the 4-row-aligned row at or
// below the segment start so that every lane's 16-byte vector load stays naturally aligned.
struct SegTiles {
    const int64_t *seg_off;     // device, 2 * n_seg  (begin, end) pairs
    const int32_t *tile_first;  // device, n_seg + 1  (exclusive prefix of tiles per segment)
    int32_t n_seg;
    int32_t n_tiles;
};

struct TileRange {
    int32_t seg;
    int64_t tile_begin;  // aligned row of the tile's first lane group
    int64_t lo, hi;      // rows of the segment covered by this tile: [lo, hi)
};

__device__ __forceinline__ TileRange locate_tile(const SegTiles &st, int32_t tile, int32_t tile_rows) {
    // upper_bound(tile_first, tile) - 1.  Windows of a schedule hold (nearly) equal row counts, so an
    // interpolated guess is right or off by one: two dependent scalar loads instead of log2(n_seg) -- the
    // block cannot issue its first column load before it knows its rows.
    int32_t lo = (int32_t)(((int64_t)tile * st.n_seg) / st.n_tiles);
    if (st.tile_first[lo] <= tile) {
        int32_t step = 1;  // gallop right
        int32_t hi = lo + 1;
        while (hi < st.n_seg && st.tile_first[hi] <= tile) { lo = hi; hi = hi + step > st.n_seg ? st.n_seg : hi + step; step <<= 1; }
        while (hi - lo > 1) {
            const int32_t mid = (lo + hi) >> 1;
            if (st.tile_first[mid] <= tile) lo = mid; else hi = mid;
        }
    } else {
        int32_t step = 1;  // gallop left
        int32_t hi = lo;
        lo = lo - 1;
        while (lo > 0 && st.tile_first[lo] > tile) { hi = lo; lo = lo - step < 0 ? 0 : lo - step; step <<= 1; }
        while (hi - lo > 1) {
            const int32_t mid = (lo + hi) >> 1;
            if (st.tile_first[mid] <= tile) lo = mid; else hi = mid;
        }
    }
    // empty segments share their tile_first with the next one: take the last segment that starts at or before
    while (lo + 1 < st.n_seg && st.tile_first[lo + 1] <= tile) ++lo;
    TileRange r;
    r.seg = lo;
    const int64_t sb = st.seg_off[2 * lo], se = st.seg_off[2 * lo + 1];
    const int64_t a0 = sb & ~int64_t(3);
    r.tile_begin = a0 + int64_t(tile - st.tile_first[lo]) * tile_rows;
    r.lo = r.tile_begin > sb ? r.tile_begin : sb;
    const int64_t te = r.tile_begin + tile_rows;
    r.hi = te < se ? te : se;
    return r;
}

// Host: builds the (begin,end) pairs + tile prefix for `n_seg` segments and uploads them (async on the ctx
// stream through pinned staging).  `name` keys the arena buffers.
inline int build_seg_tiles(flockgpu_ctx *ctx, const char *name, const int64_t *seg_begin, const int64_t *seg_end,
                           int32_t n_seg, int32_t tile_rows, SegTiles *out) {
    std::string k_off = std::string(name) + ".seg_off", k_tf = std::string(name) + ".tile_first";
    int64_t *h_off = nullptr;
    int32_t *h_tf = nullptr;
    FG_TRY(pinned_get_t(ctx, k_off.c_str(), size_t(2) * (n_seg + 1), &h_off));
    FG_TRY(pinned_get_t(ctx, k_tf.c_str(), size_t(n_seg) + 1, &h_tf));
    // the pinned staging buffers may still be in flight from the previous call on this stream
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int64_t tiles = 0;
    for (int32_t s = 0; s < n_seg; ++s) {
        h_off[2 * s] = seg_begin[s];
        h_off[2 * s + 1] = seg_end[s];
        h_tf[s] = (int32_t)tiles;
        if (seg_end[s] > seg_begin[s]) tiles += div_up(seg_end[s] - (seg_begin[s] & ~int64_t(3)), tile_rows);
        if (tiles > 0x7fffffff / 2) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: too many tiles", name);
    }
    h_tf[n_seg] = (int32_t)tiles;
    int64_t *d_off = nullptr;
    int32_t *d_tf = nullptr;
    FG_TRY(arena_get_t(ctx, k_off.c_str(), size_t(2) * (n_seg + 1), &d_off));
    FG_TRY(arena_get_t(ctx, k_tf.c_str(), size_t(n_seg) + 1, &d_tf));
    if (n_seg > 0) FG_HIP(ctx, hipMemcpyAsync(d_off, h_off, sizeof(int64_t) * 2 * n_seg, hipMemcpyHostToDevice, ctx->stream));
    FG_HIP(ctx, hipMemcpyAsync(d_tf, h_tf, sizeof(int32_t) * (n_seg + 1), hipMemcpyHostToDevice, ctx->stream));
    out->seg_off = d_off;
    out->tile_first = d_tf;
    out->n_seg = n_seg;
    out->n_tiles = (int32_t)tiles;
    return FLOCKGPU_OK;
}

// ---- wave helpers ------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Inclusive prefix sum over the 64 lanes with DPP row shifts / row broadcasts (gfx9 family, wave64): 8 VALU
// instructions, no LDS crossbar traffic (a __shfl_up ladder is 6 ds_bpermute + index math).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
    const int xi = (int)x;
    int v = xi;
    v += __builtin_amdgcn_update_dpp(0, xi, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, xi, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, xi, 0x113, 0xf, 0xf, false);  // row_shr:3
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, false);   // row_shr:4, banks 1-3
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, false);   // row_shr:8, banks 2-3
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return (uint32_t)v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// ---- striped single-pass scan --------------------------------------------------------------------------
// Order-preserving compaction needs the number of selected rows before every tile.  The kernels are
// PERSISTENT: the grid G is no larger than what is resident at once (persistent_grid below) and block b walks
// tiles b, b + G, b + 2G, ... in increasing order.  Every tile publishes its aggregate
//   status[tile] = kStValid | count      (one naturally aligned 8-byte relaxed agent-scope store: the data is
//                                         the flag, CDNA guide G16 form R2 -- no fences)
// as soon as it is known, and the block keeps its own running prefix in registers:
//   prefix(tile) = prefix(prev) + count(prev) + sum of count(t) for prev < t < tile,
// i.e. it reads the G - 1 aggregates BETWEEN its previous tile and this one, all with independent loads spread
// over the 256 lanes (one memory round trip), instead of walking a chain of predecessors.  Nothing waits on a
// prefix, only on aggregates, and an aggregate depends on nothing but the tile's own rows -- so there is no
// serial propagation (the classic decoupled look-back advanced 64 tiles per ~2 us round trip here and capped q2
// at 1.4 TB/s; a global atomic ticket per tile tops out at ~88/us on one word, MI355X_MICROARCH.md "dequeue").
// Deadlock freedom: a tile only waits on lower-numbered tiles; the lowest unpublished tile's block is either not
// yet started (it will be: the grid is resident) or finishing an earlier tile whose predecessors are all
// published.  Dispatch order and placement do not matter.
constexpr uint64_t kStValid = 1ull << 63, kStValueMask = kStValid - 1;
constexpr uint32_t kScanSpinLimit = 1u << 22;  // a few seconds; a dead predecessor becomes an error, not a hang

__device__ __forceinline__ uint64_t ld_status(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Per-block scan state of a persistent tile loop (uniform across the block).
struct StripedScan {
    int32_t prev_tile = -1;
    uint64_t running = 0;  // prefix(prev_tile) + count(prev_tile)
};

// Block-wide, once per tile of the persistent loop.  Every wave contributes `wave_total`; returns the
// exclusive offset of this wave's first element in the global output, the tile's global base in *tile_base and
// its count in *tile_total.  `smem` must hold 2 * kWavesPerBlock uint64; the two barriers inside also order
// its reuse by the next iteration.  `err` gets bit 1 when a predecessor never shows up.
__device__ __forceinline__ uint64_t block_striped_offset(uint64_t *status, StripedScan &sc, int32_t tile,
                                                         uint64_t wave_total, uint64_t *smem, uint64_t *tile_base,
                                                         uint64_t *tile_total, uint32_t *err) {
    const int wave = threadIdx.x >> 6, lane = lane_id();
    if (lane == 0) smem[wave] = wave_total;
    __syncthreads();
    uint64_t total = 0, mine = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
        const uint64_t v = smem[w];
        if (w < wave) mine += v;
        total += v;
    }
    if (threadIdx.x == 0) st_status(&status[tile], kStValid | total);  // publish before waiting on anyone
    uint64_t acc = 0;
    for (int32_t t = sc.prev_tile + 1 + (int32_t)threadIdx.x; t < tile; t += kBlock) {
        uint64_t s = ld_status(&status[t]);
        uint32_t spins = 0;
        while (!(s & kStValid)) {
            __builtin_amdgcn_s_sleep(1);
            s = ld_status(&status[t]);
            if (++spins > kScanSpinLimit) {
                atomicOr(err, 2u);
                s = kStValid;
            }
        }
        acc += s & kStValueMask;
    }
    acc = wave_sum_u64(acc);
    if (lane == 0) smem[kWavesPerBlock + wave] = acc;
    __syncthreads();
    uint64_t between = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) between += smem[kWavesPerBlock + w];
    const uint64_t base = sc.running + between;
    sc.running = base + total;
    sc.prev_tile = tile;
    *tile_base = base;
    *tile_total = total;
    return base + mine;
}

// ---- count -> scan -> emit --------------------------------------------------------------------------------
// The order-preserving operators that replaced the striped scan run as three launches with NO dependence between
// workgroups inside a launch (nothing to deadlock, no residency assumption, every launch a plain streaming grid):
//   count : one workgroup per tile writes one count per wave            counts[tile * kWavesPerBlock + wave]
//   scan  : ONE workgroup turns the counts into exclusive tile bases     tile_base[n_tiles + 1]
//           and the per-segment (window) output offsets                  seg_out_off[n_seg + 1]
//   emit  : one workgroup per tile writes its rows at tile_base[tile] + (counts of its lower waves) + rank
// The scan kernel touches 16 B per tile (2 MB for 1e9 rows) and costs a few microseconds.
constexpr int kScanBlock = 1024;

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// Host: launches the scan (defined in gather.hip).  counts: n_tiles * kWavesPerBlock uint32 (16-byte aligned);
// tile_base: n_tiles + 1; seg_out_off (may be null): n_seg + 1.
int launch_tile_scan(flockgpu_ctx *ctx, const uint32_t *counts, int32_t n_tiles, uint64_t *tile_base,
                     const int32_t *tile_first, int32_t n_seg, int64_t *seg_out_off);

// Host: grid of a persistent striped-scan kernel = min(n_tiles, blocks that are certainly co-resident).
// The occupancy API over-reports by one block per CU only where SGPRs are the limit, at 7-8 blocks per CU
// (MI355X_MICROARCH.md "Residency"); at most 4 blocks per CU are used here, where the answer (VGPR / LDS
// limited) is exact.  16 waves per CU with >= 8 x 16-byte loads in flight per lane is well past what it takes
// to stream HBM at full rate.  Should a block nevertheless not be resident, kScanSpinLimit turns the wait into
// an error status instead of a hang.
template <typename Kernel>
inline int persistent_grid(flockgpu_ctx *ctx, Kernel kernel, const char *name, int64_t n_tiles, unsigned *grid) {
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0);
    if (e != hipSuccess) return fail(ctx, FLOCKGPU_ERR_HIP, "occupancy query of %s failed: %s", name, hipGetErrorString(e));
    per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    const int64_t cap = (int64_t)per_cu * ctx->num_cus;
    *grid = (unsigned)(n_tiles < cap ? (n_tiles > 0 ? n_tiles : 1) : cap);
    return FLOCKGPU_OK;
}

}  // namespace flockgpu
