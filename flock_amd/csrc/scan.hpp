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

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// ---- chained scan (single pass, decoupled look-back) ---------------------------------------------
// status[t]: bits 63..62 = state, low 62 bits = value.  The word IS the flag (one naturally aligned 8-byte
// relaxed agent-scope store / load, CDNA guide G16 form R2), so no release/acquire fences are needed.
// Tiles take their index from an atomic ticket, so every predecessor of a running tile is itself running
// (or done) and publishes its aggregate before it waits on anything: no dependence on dispatch order.
constexpr uint64_t kStInvalid = 0ull, kStAggregate = 1ull << 62, kStPrefix = 2ull << 62, kStMask = 3ull << 62;

__device__ __forceinline__ uint64_t ld_status(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_status(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Executed by ONE full wave of the block.  Returns the exclusive prefix (sum of the aggregates of tiles
// 0..tile-1) and publishes this tile's inclusive prefix.
__device__ __forceinline__ uint64_t chained_scan_lookback(uint64_t *status, int32_t tile, uint64_t aggregate) {
    const int lane = lane_id();
    if (tile == 0) {
        if (lane == 0) st_status(&status[0], kStPrefix | aggregate);
        return 0;
    }
    if (lane == 0) st_status(&status[tile], kStAggregate | aggregate);
    uint64_t excl = 0;
    int32_t base = tile - 1;
    for (;;) {
        const int32_t t = base - lane;
        uint64_t s = kStPrefix;  // virtual tiles below 0: prefix 0
        if (t >= 0) {
            s = ld_status(&status[t]);
            while ((s & kStMask) == kStInvalid) {
                __builtin_amdgcn_s_sleep(1);
                s = ld_status(&status[t]);
            }
        }
        const uint64_t is_prefix = __ballot((s & kStMask) == kStPrefix);
        const uint64_t val = s & ~kStMask;
        if (is_prefix) {
            const int first = __ffsll((unsigned long long)is_prefix) - 1;  // nearest predecessor holding a prefix
            excl += wave_sum_u64(lane <= first ? val : 0ull);
            break;
        }
        excl += wave_sum_u64(val);  // 64 aggregates, keep looking further back
        base -= 64;
    }
    if (lane == 0) st_status(&status[tile], kStPrefix | (excl + aggregate));
    return excl;
}

// Block-wide: every wave contributes `wave_total`; returns the exclusive offset of this wave's first
// element in the global output and leaves the tile's global base in *tile_base.
// `smem` must hold kWavesPerBlock + 1 uint64.
__device__ __forceinline__ uint64_t block_chained_offset(uint64_t *status, int32_t tile, uint64_t wave_total,
                                                         uint64_t *smem, uint64_t *tile_base, uint64_t *tile_total) {
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
    if (wave == 0) {
        const uint64_t excl = chained_scan_lookback(status, tile, total);
        if (lane == 0) smem[kWavesPerBlock] = excl;
    }
    __syncthreads();
    const uint64_t base = smem[kWavesPerBlock];
    *tile_base = base;
    *tile_total = total;
    return base + mine;
}

// one ticket per block, broadcast through LDS
__device__ __forceinline__ int32_t take_ticket(uint32_t *counter, int32_t *smem_slot) {
    if (threadIdx.x == 0) *smem_slot = (int32_t)atomicAdd(counter, 1u);
    __syncthreads();
    return *smem_slot;
}

}  // namespace flockgpu
