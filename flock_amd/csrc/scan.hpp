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
// The host lays out one 32-byte descriptor per tile: a workgroup cannot issue its first column load before it
// knows its rows, and searching the per-segment tile prefix on the device cost 3-4 dependent scalar round trips
// (~2 us of a one-tile workgroup's ~15 us life); the table is ONE s_load_dwordx8.
struct TileRange {
    int64_t tile_begin;  // aligned row of the tile's first lane group
    int64_t lo, hi;      // rows of the segment covered by this tile: [lo, hi)
    int32_t seg;
    int32_t pad;
};

struct SegTiles {
    const int64_t *seg_off;     // device, 2 * n_seg  (begin, end) pairs
    const int32_t *tile_first;  // device, n_seg + 1  (exclusive prefix of tiles per segment)
    const TileRange *tiles;     // device, n_tiles
    int32_t n_seg;
    int32_t n_tiles;
};

__device__ __forceinline__ TileRange locate_tile(const SegTiles &st, int32_t tile, int32_t /*tile_rows*/) {
    return st.tiles[tile];
}

// Grid of the streaming kernels that walk tiles b, b + G, b + 2G, ... (independent tiles: nothing waits on another
// workgroup) and request the NEXT tile's descriptor before they reduce the current one, so a tile's column loads
// never queue behind a dependent descriptor load and no workgroup launch sits between two tiles.  Measured on the
// q7 max pass over 1e9 bids: one tile per workgroup 0.92 ms, 4 / 8 / 12 / 16 workgroups per CU 0.87 / 0.82 / 0.78 / 0.80 ms.
constexpr int kStreamBlocksPerCu = 12;

// Host: builds the (begin,end) pairs, the tile prefix and the tile descriptors for `n_seg` segments and uploads
// them (async on the ctx stream through pinned staging).  `name` keys the arena buffers.  A schedule identical to
// the one uploaded by the previous call under the same name is reused as is (no upload, no synchronisation): a
// streaming host re-submits the same window layout for every batch of equal shape.
inline int build_seg_tiles(flockgpu_ctx *ctx, const char *name, const int64_t *seg_begin, const int64_t *seg_end,
                           int32_t n_seg, int32_t tile_rows, SegTiles *out) {
    std::string k_off = std::string(name) + ".seg_off", k_tf = std::string(name) + ".tile_first",
                k_td = std::string(name) + ".tile_desc";
    // begin/end pairs + {tile_rows, n_tiles, the three device buffers} of the last upload: the schedule on the device is
    // reused only while it sits in the very buffers it was uploaded to (an arena entry can be freed and re-created --
    // flockgpu_plan_destroy does -- or regrown in between)
    std::vector<int64_t> &cached = ctx->host_i64[k_off];
    bool same = cached.size() == size_t(2) * n_seg + 5 && cached[size_t(2) * n_seg] == tile_rows;
    for (int32_t s = 0; same && s < n_seg; ++s) same = cached[2 * s] == seg_begin[s] && cached[2 * s + 1] == seg_end[s];
    int64_t tiles = 0;
    if (same) {
        tiles = cached[size_t(2) * n_seg + 1];
    } else {
        for (int32_t s = 0; s < n_seg; ++s)
            if (seg_end[s] > seg_begin[s]) tiles += div_up(seg_end[s] - (seg_begin[s] & ~int64_t(3)), tile_rows);
        if (tiles > 0x7fffffff / 2) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: too many tiles", name);
    }
    const std::vector<int64_t> before = cached;
    cached.clear();  // invalid until this call has the buffers (an allocation below may fail half-way) and the upload is queued
    int64_t *d_off = nullptr;
    int32_t *d_tf = nullptr;
    TileRange *d_td = nullptr;
    FG_TRY(arena_get_t(ctx, k_off.c_str(), size_t(2) * (n_seg + 1), &d_off));
    FG_TRY(arena_get_t(ctx, k_tf.c_str(), size_t(n_seg) + 1, &d_tf));
    FG_TRY(arena_get_t(ctx, k_td.c_str(), size_t(tiles) + 1, &d_td));
    out->seg_off = d_off;
    out->tile_first = d_tf;
    out->tiles = d_td;
    out->n_seg = n_seg;
    out->n_tiles = (int32_t)tiles;
    const int64_t where[3] = {(int64_t)reinterpret_cast<uintptr_t>(d_off), (int64_t)reinterpret_cast<uintptr_t>(d_tf),
                              (int64_t)reinterpret_cast<uintptr_t>(d_td)};
    if (same && before[size_t(2) * n_seg + 2] == where[0] && before[size_t(2) * n_seg + 3] == where[1] &&
        before[size_t(2) * n_seg + 4] == where[2]) {
        cached = before;
        return FLOCKGPU_OK;
    }
    int64_t *h_off = nullptr;
    int32_t *h_tf = nullptr;
    TileRange *h_td = nullptr;
    FG_TRY(pinned_get_t(ctx, k_off.c_str(), size_t(2) * (n_seg + 1), &h_off));
    FG_TRY(pinned_get_t(ctx, k_tf.c_str(), size_t(n_seg) + 1, &h_tf));
    FG_TRY(pinned_get_t(ctx, k_td.c_str(), size_t(tiles) + 1, &h_td));
    // the pinned staging buffers may still be in flight from the previous upload on this stream
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int64_t t = 0;
    for (int32_t s = 0; s < n_seg; ++s) {
        h_off[2 * s] = seg_begin[s];
        h_off[2 * s + 1] = seg_end[s];
        h_tf[s] = (int32_t)t;
        if (seg_end[s] <= seg_begin[s]) continue;
        for (int64_t b = seg_begin[s] & ~int64_t(3); b < seg_end[s]; b += tile_rows, ++t)
            h_td[t] = TileRange{b, b > seg_begin[s] ? b : seg_begin[s], b + tile_rows < seg_end[s] ? b + tile_rows : seg_end[s], s, 0};
    }
    h_tf[n_seg] = (int32_t)t;
    if (n_seg > 0) FG_HIP(ctx, hipMemcpyAsync(d_off, h_off, sizeof(int64_t) * 2 * n_seg, hipMemcpyHostToDevice, ctx->stream));
    FG_HIP(ctx, hipMemcpyAsync(d_tf, h_tf, sizeof(int32_t) * (n_seg + 1), hipMemcpyHostToDevice, ctx->stream));
    if (tiles > 0) FG_HIP(ctx, hipMemcpyAsync(d_td, h_td, sizeof(TileRange) * tiles, hipMemcpyHostToDevice, ctx->stream));
    cached.assign(h_off, h_off + size_t(2) * n_seg);
    cached.push_back(tile_rows);
    cached.push_back(tiles);
    cached.insert(cached.end(), where, where + 3);
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

// ---- count -> scan -> emit --------------------------------------------------------------------------------
// The order-preserving operators that replaced the striped scan run as three launches with NO dependence between
// workgroups inside a launch (nothing to deadlock, no residency assumption, every launch a plain streaming grid):
//   count : one workgroup per tile writes one count per wave            counts[tile * kWavesPerBlock + wave]
//   scan  : ONE workgroup turns the counts into exclusive tile bases     tile_base[n_tiles + 1]
//           and the per-segment (window) output offsets                  seg_out_off[n_seg + 1]
//   emit  : one workgroup per tile writes its rows at tile_base[tile] + (counts of its lower waves) + rank
// The scan kernel touches 16 B per tile (2 MB for 1e9 rows) and costs a few microseconds.
constexpr int kScanBlock = 1024;

// 64-bit inclusive prefix sum (values < 2^63) as three 21-bit limbs through the 32-bit DPP scan: a limb's sum over
// 64 lanes stays below 2^27.
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
    const uint32_t s0 = wave_incl_scan_u32((uint32_t)v & 0x1FFFFFu);
    const uint32_t s1 = wave_incl_scan_u32((uint32_t)(v >> 21) & 0x1FFFFFu);
    const uint32_t s2 = wave_incl_scan_u32((uint32_t)(v >> 42) & 0x1FFFFFu);
    return (uint64_t)s0 + ((uint64_t)s1 << 21) + ((uint64_t)s2 << 42);
}

// Self-scan: the exclusive base of tile `tile` = sum of counts[0 .. tile * kWavesPerBlock), summed by the emitting workgroup itself --
// every count was written by the PREVIOUS launch, so no scan launch (and no boundary either side of it) sits between count and emit.
// 16 B per lower tile from L2: 733 tiles (q3 at 1e8 events) read 6 KB each on average, 7324 tiles 58 KB; relations of more than
// kSelfScanMaxTiles tiles keep the scan kernel.  All threads of the workgroup get the sum; s_red: kWavesPerBlock words of LDS.
constexpr int32_t kSelfScanMaxTiles = 16384;
__device__ __forceinline__ uint64_t block_base_of_tile(const uint32_t *__restrict__ counts, int32_t tile, uint64_t *s_red) {
    uint64_t sum = 0;
    for (int32_t t = (int32_t)threadIdx.x; t < tile; t += kBlock) {
        const uint4 w = *reinterpret_cast<const uint4 *>(counts + (size_t)t * kWavesPerBlock);
        sum += (uint64_t)w.x + w.y + w.z + w.w;
    }
    sum = wave_sum_u64(sum);
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = sum;
    __syncthreads();
    uint64_t tot = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) tot += s_red[w];
    __syncthreads();   // (s_red may be reused)
    return tot;
}

// Host: launches the scan (defined in gather.hip).  counts: n_tiles * kWavesPerBlock uint32 (16-byte aligned);
// tile_base: n_tiles + 1; seg_out_off (may be null): n_seg + 1.
int launch_tile_scan(flockgpu_ctx *ctx, const uint32_t *counts, int32_t n_tiles, uint64_t *tile_base,
                     const int32_t *tile_first, int32_t n_seg, int64_t *seg_out_off);

// ---- flag tiles ------------------------------------------------------------------------------------------------
// Common geometry of the "count" kernels that select rows (q2 filter, q3 probe, q8 persons): 8192-row tiles,
// lane l of wave w holds rows  w*2048 + it*256 + 4l .. 4l+3  relative to tile_begin (it = 0..7), so every 16-byte
// load instruction of a wave covers 1 KiB of consecutive bytes, and a lane's 32 row flags fill ONE 32-bit word
// (bit it*4 + j) that is stored coalesced: 1 KiB of flag words per tile.
constexpr int kFlagIters = 8;
constexpr int kFlagTile = kBlock * 4 * kFlagIters;        // 8192 rows
constexpr int kFlagWaveRows = kFlagTile / kWavesPerBlock;  // 2048

__device__ __forceinline__ int32_t flag_rel0() { return (int32_t)(threadIdx.x >> 6) * kFlagWaveRows + lane_id() * 4; }

__device__ __forceinline__ void load4_i32(const int32_t *__restrict__ col, int64_t r0, int64_t n_rows, int32_t (&v)[4]) {
    if (r0 >= 0 && r0 + 4 <= n_rows) {
        const int4 t = *reinterpret_cast<const int4 *>(col + r0);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (r0 + j >= 0 && r0 + j < n_rows) ? col[r0 + j] : 0;
    }
}

// Loads the tile's values of an int32 column (16-byte aligned base) in the flag-tile layout.
__device__ __forceinline__ void load_flag_tile(const int32_t *__restrict__ col, int64_t n_rows, const TileRange &tr,
                                               int32_t (&a)[kFlagIters][4]) {
    const int64_t wbase = tr.tile_begin + flag_rel0();
    if (tr.tile_begin + kFlagTile <= n_rows) {  // block-uniform: no row of the tile is past the column
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it) {
#if defined(FLOCKGPU_EXPERIMENTAL) && defined(FLOCKGPU_AB_PLAIN_TILE_LOADS)   // (A/B builds only: tools/gpu_ab_stream_loads.sh)
            const int4 t = *reinterpret_cast<const int4 *>(col + wbase + it * 256);
#else
            const int4 t = stream_load4(col + wbase + it * 256);   // (read once per pass: non-temporal, common.hpp)
#endif
            a[it][0] = t.x; a[it][1] = t.y; a[it][2] = t.z; a[it][3] = t.w;
        }
    } else {
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it) load4_i32(col, wbase + it * 256, n_rows, a[it]);
    }
}

// The same with ordinary (cached) loads, for a column a second pass reads again right away.
__device__ __forceinline__ void load_flag_tile_cached(const int32_t *__restrict__ col, int64_t n_rows, const TileRange &tr,
                                                      int32_t (&a)[kFlagIters][4]) {
    const int64_t wbase = tr.tile_begin + flag_rel0();
    if (tr.tile_begin + kFlagTile <= n_rows) {
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it) {
            const int4 t = *reinterpret_cast<const int4 *>(col + wbase + it * 256);
            a[it][0] = t.x; a[it][1] = t.y; a[it][2] = t.z; a[it][3] = t.w;
        }
    } else {
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it) load4_i32(col, wbase + it * 256, n_rows, a[it]);
    }
}

__device__ __forceinline__ void store_flags_and_counts(uint32_t flags, int32_t tile, uint32_t *__restrict__ flag_words,
                                                       uint32_t *__restrict__ counts) {
    flag_words[(size_t)tile * kBlock + threadIdx.x] = flags;
    const uint32_t incl = wave_incl_scan_u32((uint32_t)__popc(flags));
    if (lane_id() == 63) counts[(size_t)tile * kWavesPerBlock + (threadIdx.x >> 6)] = incl;
}

// Emit side: turns this lane's flag word into the tile-relative row numbers of the flagged rows, written in row
// order into `s_list` (kFlagTile entries).  `wc` = the tile's four wave counts.  Returns the tile's total; the
// caller must __syncthreads() before reading the list.
__device__ __forceinline__ uint32_t build_flag_list(uint32_t flags, const uint4 &wc, uint16_t *s_list) {
    const int wave = threadIdx.x >> 6;
    uint32_t p = (wave > 0 ? wc.x : 0u) + (wave > 1 ? wc.y : 0u) + (wave > 2 ? wc.z : 0u);
    const int32_t rel0 = flag_rel0();
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) {
        const uint32_t f4 = (flags >> (it * 4)) & 15u;
        if (!__ballot(f4 != 0)) continue;  // wave-uniform
        const uint32_t c = (uint32_t)__popc(f4);
        const uint32_t incl = wave_incl_scan_u32(c);
        uint32_t q = p + incl - c;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (f4 & (1u << j)) s_list[q++] = (uint16_t)(rel0 + it * 256 + j);
        p += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    return wc.x + wc.y + wc.z + wc.w;
}

}  // namespace flockgpu
