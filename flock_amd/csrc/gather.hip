// Shared device-side primitives (gather.hpp): tile scan, flag words -> rows, per-window key statistics,
// take() for fixed-width and Utf8 columns, in-place prefix sums.  Every multi-tile operation is
// count -> scan -> emit (scan.hpp): no workgroup ever waits on another one inside a launch.
#include "gather.hpp"

#include <algorithm>

using namespace flockgpu;

namespace {

// ---- tile scan: ONE workgroup, every access coalesced, all loads of a pass in flight together -----------------
// Pass = kScanRounds x 1024 tiles: thread t holds tiles  k*1024 + t  (k = 0..15).  Per round a wave scan; the
// 16 x 16 wave totals are scanned by wave 0; three barriers per pass.
constexpr int kScanRounds = 16;
__global__ __launch_bounds__(kScanBlock) void tile_scan_kernel(const uint32_t *__restrict__ counts, int32_t n_tiles,
                                                              uint64_t *__restrict__ tile_base,
                                                              const int32_t *__restrict__ tile_first, int32_t n_seg,
                                                              int64_t *__restrict__ seg_out_off) {
    constexpr int kWaves = kScanBlock / 64;
    __shared__ uint64_t s_tot[kScanRounds * kWaves];  // [round][wave] totals, then their exclusive prefix
    __shared__ uint64_t s_carry, s_next;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    for (int32_t p0 = 0; p0 < n_tiles; p0 += kScanRounds * kScanBlock) {
        uint64_t c[kScanRounds], incl[kScanRounds];
#pragma unroll
        for (int k = 0; k < kScanRounds; ++k) {
            const int32_t t = p0 + k * kScanBlock + (int32_t)threadIdx.x;
            c[k] = 0;
            if (t < n_tiles) {
                const uint4 w = *reinterpret_cast<const uint4 *>(counts + (size_t)t * kWavesPerBlock);
                c[k] = (uint64_t)w.x + w.y + w.z + w.w;
            }
        }
#pragma unroll
        for (int k = 0; k < kScanRounds; ++k) {
            incl[k] = 0;
            if (p0 + k * kScanBlock < n_tiles) incl[k] = wave_incl_scan_u64(c[k]);  // block-uniform
            if (lane == 63) s_tot[k * kWaves + wave] = incl[k];
        }
        __syncthreads();  // (also publishes s_carry)
        const uint64_t carry = s_carry;
        if (wave == 0) {  // exclusive scan of the 256 totals: 4 per lane
            uint64_t v[4], sum = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = s_tot[lane * 4 + i];
                sum += v[i];
            }
            uint64_t run = wave_incl_scan_u64(sum) - sum;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s_tot[lane * 4 + i] = run;
                run += v[i];
            }
            if (lane == 63) s_next = carry + run;  // not s_carry: the other waves may not have read it yet
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScanRounds; ++k) {
            const int32_t t = p0 + k * kScanBlock + (int32_t)threadIdx.x;
            if (t < n_tiles) tile_base[t] = carry + s_tot[k * kWaves + wave] + incl[k] - c[k];
        }
        __syncthreads();  // s_tot is rewritten by the next pass; everyone has read s_carry
        if (threadIdx.x == 0) s_carry = s_next;  // published by the next barrier
    }
    __syncthreads();
    const uint64_t total = s_carry;
    if (threadIdx.x == 0) tile_base[n_tiles] = total;
    if (!seg_out_off) return;
    // the workgroup reads back its own global stores (made before the barriers above) with L1-bypassing loads
    for (int32_t s = threadIdx.x; s <= n_seg; s += kScanBlock) {
        const int32_t t = tile_first[s];  // an empty segment shares the next segment's first tile
        seg_out_off[s] = (int64_t)(t >= n_tiles ? total
                                                : __hip_atomic_load(&tile_base[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
}

// ---- tile scan over more than one pass: one workgroup per 16384-tile chunk ------------------------------------------
// The single workgroup above walks its passes one after the other (122 K tiles of a 1e9-row column: 0.10 ms, as much as
// the emit kernel it serves).  Beyond one pass the chunks are scanned side by side: chunk-local exclusive bases + chunk
// totals, a single-workgroup scan of the totals, and a fix-up pass that adds every chunk's base.
__global__ __launch_bounds__(kScanBlock) void tile_scan_chunk_kernel(const uint32_t *__restrict__ counts, int32_t n_tiles,
                                                                    uint64_t *__restrict__ tile_base,
                                                                    uint64_t *__restrict__ chunk_total) {
    constexpr int kWaves = kScanBlock / 64;
    __shared__ uint64_t s_tot[kScanRounds * kWaves];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int32_t p0 = (int32_t)blockIdx.x * kScanRounds * kScanBlock;
    uint64_t c[kScanRounds], incl[kScanRounds];
#pragma unroll
    for (int k = 0; k < kScanRounds; ++k) {
        const int32_t t = p0 + k * kScanBlock + (int32_t)threadIdx.x;
        c[k] = 0;
        if (t < n_tiles) {
            const uint4 w = *reinterpret_cast<const uint4 *>(counts + (size_t)t * kWavesPerBlock);
            c[k] = (uint64_t)w.x + w.y + w.z + w.w;
        }
    }
#pragma unroll
    for (int k = 0; k < kScanRounds; ++k) {
        incl[k] = wave_incl_scan_u64(c[k]);
        if (lane == 63) s_tot[k * kWaves + wave] = incl[k];
    }
    __syncthreads();
    if (wave == 0) {  // exclusive scan of the 256 (round, wave) totals: 4 per lane
        uint64_t v[4], sum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = s_tot[lane * 4 + i];
            sum += v[i];
        }
        uint64_t run = wave_incl_scan_u64(sum) - sum;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s_tot[lane * 4 + i] = run;
            run += v[i];
        }
        if (lane == 63) chunk_total[blockIdx.x] = run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kScanRounds; ++k) {
        const int32_t t = p0 + k * kScanBlock + (int32_t)threadIdx.x;
        if (t < n_tiles) tile_base[t] = s_tot[k * kWaves + wave] + incl[k] - c[k];
    }
}

// chunk_base[i] = sum of chunk_total[0 .. i), i = 0 .. n_chunks (one workgroup; n_chunks is at most 2^16)
__global__ __launch_bounds__(kScanBlock) void chunk_scan_kernel(const uint64_t *__restrict__ chunk_total, int32_t n_chunks,
                                                               uint64_t *__restrict__ chunk_base) {
    constexpr int kWaves = kScanBlock / 64;
    __shared__ uint64_t s_wave[kWaves];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int32_t i0 = 0; i0 < n_chunks; i0 += kScanBlock) {
        const int32_t i = i0 + (int32_t)threadIdx.x;
        const uint64_t v = i < n_chunks ? chunk_total[i] : 0;
        const uint64_t incl = wave_incl_scan_u64(v);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint64_t base = s_carry + incl - v;
        for (int w = 0; w < wave; ++w) base += s_wave[w];
        if (i < n_chunks) chunk_base[i] = base;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) s_carry = base + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) chunk_base[n_chunks] = s_carry;
}

__global__ __launch_bounds__(kBlock) void tile_scan_fix_kernel(uint64_t *__restrict__ tile_base, int32_t n_tiles,
                                                               const uint64_t *__restrict__ chunk_base, int32_t n_chunks) {
    const int32_t t = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (t < n_tiles) tile_base[t] += chunk_base[t / (kScanRounds * kScanBlock)];
    if (t == n_tiles) tile_base[n_tiles] = chunk_base[n_chunks];
}

// seg_out_off[s] = output offset of segment s = base of its first tile (an empty segment shares the next one's)
__global__ __launch_bounds__(kBlock) void seg_offsets_kernel(const uint64_t *__restrict__ tile_base, int32_t n_tiles,
                                                             const int32_t *__restrict__ tile_first, int32_t n_seg,
                                                             int64_t *__restrict__ seg_out_off) {
    const int32_t s = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (s <= n_seg) {
        const int32_t t = tile_first[s];
        seg_out_off[s] = (int64_t)tile_base[t >= n_tiles ? n_tiles : t];
    }
}

// ---- flag words -> global row numbers ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void emit_rows_kernel(SegTiles st, const uint32_t *__restrict__ flag_words,
                                                           const uint32_t *__restrict__ counts,
                                                           const uint64_t *__restrict__ tile_base,
                                                           int32_t *__restrict__ out_rows) {
    __shared__ uint16_t s_list[kFlagTile];
    const int32_t tile = (int32_t)blockIdx.x;
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)tile * kWavesPerBlock);
    if (wc.x + wc.y + wc.z + wc.w == 0) return;
    const uint32_t total = build_flag_list(flag_words[(size_t)tile * kBlock + threadIdx.x], wc, s_list);
    __syncthreads();
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const uint64_t base = tile_base[tile];
    for (uint32_t i = threadIdx.x; i < total; i += kBlock) out_rows[base + i] = (int32_t)(tr.tile_begin + s_list[i]);
}

// The same without a scan launch in front (ONE segment, a relation of up to kSelfScanMaxTiles tiles): the workgroup sums the counts of the lower
// tiles itself (block_base_of_tile); the last tile reports the total where the host reads it (pinned: off[0] = 0, off[1] = selected rows).
__global__ __launch_bounds__(kBlock) void emit_rows_self_kernel(SegTiles st, const uint32_t *__restrict__ flag_words, const uint32_t *__restrict__ counts,
                                                                int32_t *__restrict__ out_rows, int64_t *__restrict__ h_off) {
    __shared__ uint16_t s_list[kFlagTile];
    __shared__ uint64_t s_red[kWavesPerBlock];
    const int32_t tile = (int32_t)blockIdx.x;
    const uint64_t base = block_base_of_tile(counts, tile, s_red);
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)tile * kWavesPerBlock);
    if (tile == (int32_t)gridDim.x - 1 && threadIdx.x == 0) {
        h_off[0] = 0;
        h_off[1] = (int64_t)(base + wc.x + wc.y + wc.z + wc.w);
    }
    if (wc.x + wc.y + wc.z + wc.w == 0) return;
    const uint32_t total = build_flag_list(flag_words[(size_t)tile * kBlock + threadIdx.x], wc, s_list);
    __syncthreads();
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    for (uint32_t i = threadIdx.x; i < total; i += kBlock) out_rows[base + i] = (int32_t)(tr.tile_begin + s_list[i]);
}

__global__ __launch_bounds__(kBlock) void emit_bids_kernel(const int32_t *__restrict__ auction, const int32_t *__restrict__ price,
                                                           const int32_t *__restrict__ bidder,
                                                           const int64_t *__restrict__ b_date_time, SegTiles st,
                                                           const uint32_t *__restrict__ flag_words,
                                                           const uint32_t *__restrict__ counts,
                                                           const uint64_t *__restrict__ tile_base, int32_t *__restrict__ o_auction,
                                                           int32_t *__restrict__ o_price, int32_t *__restrict__ o_bidder,
                                                           int64_t *__restrict__ o_time) {
    __shared__ uint16_t s_list[kFlagTile];
    const int32_t tile = (int32_t)blockIdx.x;
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)tile * kWavesPerBlock);
    if (wc.x + wc.y + wc.z + wc.w == 0) return;
    const uint32_t total = build_flag_list(flag_words[(size_t)tile * kBlock + threadIdx.x], wc, s_list);
    __syncthreads();
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const uint64_t base = tile_base[tile];
    for (uint32_t i = threadIdx.x; i < total; i += kBlock) {
        const int64_t r = tr.tile_begin + s_list[i];
        stream_store(&o_auction[base + i], auction[r]);   // (result columns: not read again on the device)
        stream_store(&o_price[base + i], price[r]);
        stream_store(&o_bidder[base + i], bidder[r]);
        stream_store(&o_time[base + i], b_date_time[r]);
    }
}

// ---- exact per-segment min / max + "strictly increasing" ---------------------------------------------------------
__global__ __launch_bounds__(kBlock) void segment_stats_kernel(const int32_t *__restrict__ col, int64_t n_rows, SegTiles st,
                                                               int32_t *seg_min, int32_t *seg_max, int32_t *seg_sorted) {
    __shared__ int32_t s_red[3 * kWavesPerBlock];
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    int32_t a[kFlagIters][4];
    load_flag_tile(col, n_rows, tr, a);
    const int32_t rel_lo = (int32_t)(tr.lo - tr.tile_begin), rel_hi = (int32_t)(tr.hi - tr.tile_begin);
    const int32_t rel0 = flag_rel0();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
    bool sorted = true;
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) {
        // predecessor of this lane's first row: the previous lane's last row, or (lane 0) the row before in memory
        int32_t prev = __shfl_up(a[it][3], 1, 64);
        const int32_t rel = rel0 + it * 256;
        if (lane == 0) {
            const int64_t r = tr.tile_begin + rel - 1;
            prev = (rel > rel_lo && r >= 0 && r < n_rows) ? col[r] : 0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (rel + j >= rel_lo && rel + j < rel_hi) {
                mn = min(mn, a[it][j]);
                mx = max(mx, a[it][j]);
                if (rel + j > rel_lo && a[it][j] <= prev) sorted = false;  // the segment's first row has no predecessor
            }
            prev = a[it][j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    const bool wave_sorted = __ballot(!sorted) == 0;
    if (lane == 0) {
        s_red[wave] = mn;
        s_red[kWavesPerBlock + wave] = mx;
        s_red[2 * kWavesPerBlock + wave] = wave_sorted ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0 && tr.hi > tr.lo) {
        mn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
        mx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
        atomicMin(&seg_min[tr.seg], mn);
        atomicMax(&seg_max[tr.seg], mx);
        if (!(s_red[8] & s_red[9] & s_red[10] & s_red[11])) atomicAnd(&seg_sorted[tr.seg], 0);
    }
}

__global__ __launch_bounds__(kBlock) void fill_stats_kernel(int32_t *seg_min, int32_t *seg_max, int32_t *seg_sorted, int32_t n) {
    const int32_t i = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (i < n) {
        seg_min[i] = 0x7fffffff;
        seg_max[i] = (int32_t)0x80000000;
        seg_sorted[i] = 1;
    }
}

// ---- fixed-width take ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gather_i32_kernel(const int32_t *__restrict__ src,
                                                            const int32_t *__restrict__ rows, int64_t n,
                                                            int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = src[rows[i]];
}

__global__ __launch_bounds__(kBlock) void gather_i64_kernel(const int64_t *__restrict__ src, const int32_t *__restrict__ rows,
                                                            int64_t n, int64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = src[rows[i]];
}

__global__ __launch_bounds__(kBlock) void gather_multi_kernel(GatherCols cols, const int32_t *__restrict__ rows, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n4; q += (int64_t)gridDim.x * kBlock) {
        const int4 r = *reinterpret_cast<const int4 *>(rows + q * 4);   // (the list is an arena buffer: 16-byte aligned)
        for (int c = 0; c < cols.n; ++c) {   // (uniform trip count; every column's loads are independent of the others')
            if (cols.width[c] == 4) {
                const int32_t *s = static_cast<const int32_t *>(cols.src[c]);
                const int4 v = make_int4(s[r.x], s[r.y], s[r.z], s[r.w]);
                *reinterpret_cast<int4 *>(static_cast<int32_t *>(cols.out[c]) + q * 4) = v;
            } else {
                const int64_t *s = static_cast<const int64_t *>(cols.src[c]);
                const int64_t a = s[r.x], b = s[r.y], d = s[r.z], e = s[r.w];
                int64_t *o = static_cast<int64_t *>(cols.out[c]) + q * 4;
                *reinterpret_cast<longlong2 *>(o) = make_longlong2(a, b);
                *reinterpret_cast<longlong2 *>(o + 2) = make_longlong2(d, e);
            }
        }
    }
    // the up to three rows behind the last whole group of four
    const int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (blockIdx.x == 0 && i < n)
        for (int c = 0; c < cols.n; ++c) {
            if (cols.width[c] == 4) static_cast<int32_t *>(cols.out[c])[i] = static_cast<const int32_t *>(cols.src[c])[rows[i]];
            else static_cast<int64_t *>(cols.out[c])[i] = static_cast<const int64_t *>(cols.src[c])[rows[i]];
        }
}

// ---- Utf8 take: lengths (count) -> tile scan -> offsets + bytes (emit) --------------------------------------------
// Tile = 1024 values; value  it*256 + tid  of the tile belongs to thread tid (it = 0..3): the row list and the
// source offsets are read coalesced.  counts[tile*4 + wave] = bytes of the wave's values.
constexpr int kLenItems = 4;
constexpr int kLenTile = kBlock * kLenItems;
constexpr int kMaxUtf8Multi = 4;
constexpr int kStageBytes = 18 * 1024;  // LDS staging buffer of the emit kernel: 18 B per value on average, 8 workgroups
                                        // per CU (tiles beyond it copy directly, byte by byte)

// src_off[r] and src_off[r + 1] through ONE 8-byte load (dword-aligned, which is all the hardware asks of a global load): with one row per
// lane every load instruction of a wave touches up to 64 cache lines, and the texture path takes them one line per cycle -- the Utf8 take is
// bound by its count of divergent load instructions (six per value before: two offsets, four dwords of bytes), not by bytes.
__device__ __forceinline__ int2 load_off_pair(const int32_t *__restrict__ src_off, int32_t r) {
    int2 v;
    __builtin_memcpy(&v, src_off + r, 8);
    return v;
}

// d_n (may be null): the row list's length when only the device knows it yet; n is then an upper bound.
__global__ __launch_bounds__(kBlock) void utf8_len_kernel(const int32_t *__restrict__ src_off,
                                                          const int32_t *__restrict__ rows, int64_t n,
                                                          const uint64_t *__restrict__ d_n, uint32_t *__restrict__ counts) {
    if (d_n) n = min(n, (int64_t)*d_n);
    const int64_t i0 = (int64_t)blockIdx.x * kLenTile + threadIdx.x;
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kLenItems; ++k)
        if (i0 + k * kBlock < n) {
            const int2 o = load_off_pair(src_off, rows[i0 + k * kBlock]);
            mine += (uint32_t)(o.y - o.x);
        }
    const uint32_t incl = wave_incl_scan_u32(mine);
    if (lane_id() == 63) counts[(size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)] = incl;
}

// The same for up to four columns gathered with ONE row list (the three Utf8 columns of q3's join output; the Utf8 columns of
// a relation in the exchange): the row numbers are read once, the columns' counts lie one after the other
// (column c: counts + c * tiles_stride * 4), so that ONE tile scan over k * tiles_stride pseudo-tiles serves all of them.
struct Utf8Cols {
    const int32_t *src_off[kMaxUtf8Multi];
    const uint8_t *src[kMaxUtf8Multi];
    int32_t *out_off[kMaxUtf8Multi];
    uint8_t *out[kMaxUtf8Multi];
    int32_t k;
};
__global__ __launch_bounds__(kBlock) void utf8_len_multi_kernel(Utf8Cols cols, const int32_t *__restrict__ rows, int64_t n,
                                                                const uint64_t *__restrict__ d_n, int64_t tiles_stride,
                                                                uint32_t *__restrict__ counts) {
    if (d_n) n = min(n, (int64_t)*d_n);
    const int64_t i0 = (int64_t)blockIdx.x * kLenTile + threadIdx.x;
    int32_t r[kLenItems];
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) r[k] = i0 + k * kBlock < n ? rows[i0 + k * kBlock] : -1;
#pragma unroll
    for (int c = 0; c < kMaxUtf8Multi; ++c) {
        if (c >= cols.k) break;
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < kLenItems; ++k)
            if (r[k] >= 0) {
                const int2 o = load_off_pair(cols.src_off[c], r[k]);
                mine += (uint32_t)(o.y - o.x);
            }
        const uint32_t incl = wave_incl_scan_u32(mine);
        if (lane_id() == 63) counts[((size_t)c * tiles_stride + blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6)] = incl;
    }
}
// col_base[c] = tile_base[c * tiles_stride], c = 0 .. k: where column c's bytes start in the scan over all columns
// (h_col_base: the host's pinned copy, written by the same threads instead of a copy call behind the kernel)
__global__ void utf8_col_bases_kernel(const uint64_t *__restrict__ tile_base, int64_t tiles_stride, int32_t k, uint64_t *__restrict__ col_base,
                                      uint64_t *__restrict__ h_col_base) {
    if ((int)threadIdx.x <= k) {
        const uint64_t b = tile_base[(int64_t)threadIdx.x * tiles_stride];
        col_base[threadIdx.x] = b;
        h_col_base[threadIdx.x] = b;
    }
}

// Bytes of every Utf8 column per RUN of send-order rows (the exchange's per-destination byte counts, comm.hip) from the take's own scan: the
// bytes in front of row b = the scanned base of b's tile + the lengths of the tile's rows below b.  Workgroup (d, c) sums at most two partial
// tiles (rounds 2-5: a pass of its own over every row and column, 12 B of scattered reads per value -- 3 x 44 us per call of q3's exchange).
struct RunByteOut {
    unsigned long long *run_bytes[kMaxUtf8Multi];
};
__device__ __forceinline__ uint64_t utf8_bytes_before(const int32_t *__restrict__ src_off, const int32_t *__restrict__ rows, int64_t b,
                                                      const uint64_t *__restrict__ col_tile_base) {
    const int64_t tile = b / kLenTile;
    uint64_t sum = 0;
    for (int64_t i = tile * kLenTile + threadIdx.x; i < b; i += kBlock) {
        const int2 o = load_off_pair(src_off, rows[i]);
        sum += (uint64_t)(uint32_t)(o.y - o.x);
    }
    sum = wave_sum_u64(sum);   // (every lane of the wave holds it)
    return sum + (threadIdx.x == 0 ? col_tile_base[tile] - col_tile_base[0] : 0);
}
__global__ __launch_bounds__(kBlock) void utf8_run_bytes_kernel(Utf8Cols cols, const int32_t *__restrict__ rows, const int64_t *__restrict__ run_start,
                                                                const uint64_t *__restrict__ tile_base, int64_t tiles_stride, RunByteOut out) {
    __shared__ unsigned long long s_part[2][kWavesPerBlock];
    const int d = blockIdx.x, c = blockIdx.y;
    const uint64_t *ctb = tile_base + (int64_t)c * tiles_stride;
    const uint64_t lo = utf8_bytes_before(cols.src_off[c], rows, run_start[d], ctb), hi = utf8_bytes_before(cols.src_off[c], rows, run_start[d + 1], ctb);
    if (lane_id() == 0) {
        s_part[0][threadIdx.x >> 6] = lo;
        s_part[1][threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0, b = 0;
        for (int w = 0; w < kWavesPerBlock; ++w) {
            a += s_part[0][w];
            b += s_part[1][w];
        }
        out.run_bytes[c][d] = b - a;
    }
}

// Writes out_off and the bytes.  The tile's bytes form ONE contiguous range of the output, so they are assembled
// in LDS (byte writes are cheap there) and streamed out with aligned 16-byte stores; byte-granular global stores
// made this kernel 10x slower than everything else in q8.
__device__ __forceinline__ void utf8_emit_tile_at(const int32_t *__restrict__ src_off, const uint8_t *__restrict__ src,
                                                  const int32_t *__restrict__ rows, int64_t n, const uint32_t *__restrict__ counts,
                                                  const uint64_t base, int32_t *__restrict__ out_off,
                                                  uint8_t *__restrict__ out, uint8_t *s_stage, uint32_t *s_it) {
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int64_t i0 = (int64_t)blockIdx.x * kLenTile + threadIdx.x;
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)blockIdx.x * kWavesPerBlock);
    const uint32_t tile_bytes = wc.x + wc.y + wc.z + wc.w;
    int32_t b[kLenItems];
    uint32_t len[kLenItems], excl[kLenItems];
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
        b[k] = 0;
        len[k] = 0;
        if (i0 + k * kBlock < n) {
            const int2 o = load_off_pair(src_off, rows[i0 + k * kBlock]);
            b[k] = o.x;
            len[k] = (uint32_t)(o.y - o.x);
        }
        const uint32_t incl = wave_incl_scan_u32(len[k]);
        excl[k] = incl - len[k];
        if (lane == 63) s_it[k * kWavesPerBlock + wave] = incl;
    }
    __syncthreads();
    // byte offset of (iteration k, wave) inside the tile: values are ordered iteration-major, then thread
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) {
            if (w == wave) excl[k] += run;
            run += s_it[k * kWavesPerBlock + w];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out_off[0] = 0;
#pragma unroll
    for (int k = 0; k < kLenItems; ++k)
        if (i0 + k * kBlock < n) stream_store(&out_off[i0 + k * kBlock + 1], (int32_t)(base + excl[k] + len[k]));
    if (tile_bytes == 0) return;
    const uint32_t phase = (uint32_t)(base & 15);  // LDS byte i holds output byte (base - phase) + i
    const bool staged = phase + tile_bytes <= (uint32_t)kStageBytes;  // block-uniform
    if (!staged) {
        // A tile of LONG values (an auction's description: ~75 bytes a value, 77 KB a tile): the tile's output range goes through the
        // stage in rounds of kStageBytes -- every value writes the bytes of it that fall into the round's window, the window is
        // streamed out with aligned 16-byte stores.  (Byte-granular global stores, what this case did until round 5, ran the
        // reference's join.sql at 0.12 TB/s: 14 of its 15 ms.)
        const uint32_t end = phase + tile_bytes;
        uint8_t *gout = out + (base - phase);  // 16-byte aligned
        for (uint32_t w0 = 0; w0 < end; w0 += (uint32_t)kStageBytes) {   // (block-uniform)
            const uint32_t w1 = w0 + (uint32_t)kStageBytes < end ? w0 + (uint32_t)kStageBytes : end;
#pragma unroll
            for (int k = 0; k < kLenItems; ++k) {
                const uint32_t p0 = phase + excl[k], p1 = p0 + len[k];
                const uint32_t lo = p0 > w0 ? p0 : w0, hi = p1 < w1 ? p1 : w1;
                if (lo >= hi) continue;
                const uintptr_t a = reinterpret_cast<uintptr_t>(src) + (uint32_t)b[k] + (lo - p0);
                const uint32_t *w = reinterpret_cast<const uint32_t *>(a & ~uintptr_t(3));
                uint32_t cur = *w++ >> (8 * (uint32_t)(a & 3)), have = 4 - (uint32_t)(a & 3);
                uint8_t *dst = s_stage + (lo - w0);
                for (uint32_t done = 0, todo = hi - lo; done < todo; ++done) {
                    if (have == 0) {
                        cur = *w++;
                        have = 4;
                    }
                    dst[done] = (uint8_t)cur;
                    cur >>= 8;
                    --have;
                }
            }
            __syncthreads();
            for (uint32_t o = w0 + threadIdx.x * 16; o < w1; o += kBlock * 16) {
                if (o >= phase && o + 16 <= end) {
                    stream_store4(gout + o, *reinterpret_cast<const uint4 *>(s_stage + (o - w0)));
                } else {  // the tile's first / last chunk is shared with the neighbouring tile: only this tile's bytes
                    for (uint32_t c = (o < phase ? phase : o); c < o + 16 && c < end; ++c) gout[c] = s_stage[c - w0];
                }
            }
            __syncthreads();   // (the next round overwrites the stage)
        }
        return;
    }
    // The first 16 bytes of every value through ONE or TWO 16-byte ALIGNED loads (a string starts at any byte), all of them requested
    // together, before any is used: one memory round trip for the whole tile.  An aligned 16-byte chunk that holds at least one byte
    // of the value never crosses a page, so nothing is read that could fault; the second chunk is asked for only by the lanes whose
    // head runs into it.  (Four dword loads per value before: twice to four times the divergent load instructions.)
    uint4 c0[kLenItems], c1[kLenItems];
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
        const uintptr_t addr = reinterpret_cast<uintptr_t>(src) + (uint32_t)b[k];
        const uint4 *q = reinterpret_cast<const uint4 *>(addr & ~uintptr_t(15));
        const uint32_t o = (uint32_t)(addr & 15), head_len = len[k] < 16u ? len[k] : 16u;
        c0[k] = make_uint4(0, 0, 0, 0);
        c1[k] = make_uint4(0, 0, 0, 0);
        if (len[k]) c0[k] = q[0];
        if (o + head_len > 16u) c1[k] = q[1];
    }
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
        if (len[k] == 0) continue;
        const uintptr_t addr = reinterpret_cast<uintptr_t>(src) + (uint32_t)b[k];
        const uint32_t o = (uint32_t)(addr & 15), qd = o >> 2, sh = (o & 3) * 8;
        // bytes 0..15 of the string, realigned: d[i] = bytes 4i .. 4i+3 = dwords qd + i, qd + i + 1 of the 32 loaded bytes, shifted
        const uint32_t W[9] = {c0[k].x, c0[k].y, c0[k].z, c0[k].w, c1[k].x, c1[k].y, c1[k].z, c1[k].w, 0u};
        // (the window of five dwords that starts qd dwords in: two rounds of bit-selects -- the four-way ?: came back from the compiler as four
        // divergent copies of the code behind it, utf8_emit_long_kernel below)
        const uint32_t by1 = 0u - (qd & 1u), by2 = 0u - (qd >> 1);
        uint32_t X[7], V[5];
#pragma unroll
        for (int j = 0; j < 7; ++j) X[j] = (W[j + 1] & by1) | (W[j] & ~by1);
#pragma unroll
        for (int j = 0; j < 5; ++j) V[j] = (X[j + 2] & by2) | (X[j] & ~by2);
        uint32_t d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __funnelshift_r(V[i], V[i + 1], sh);
        uint8_t *dst = s_stage + phase + excl[k];
        const uint32_t head = 16;  // bytes in d[]
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if ((uint32_t)c < len[k]) dst[c] = (uint8_t)(d[c >> 2] >> (8 * (c & 3)));
        if (len[k] > head) {  // long string: the rest word by word, from the first word that holds byte 16 on
            const uintptr_t a16 = addr + 16;
            const uint32_t *w = reinterpret_cast<const uint32_t *>(a16 & ~uintptr_t(3));
            uint32_t done = head, cur = *w++ >> (8 * (uint32_t)(a16 & 3)), have = 4 - (uint32_t)(a16 & 3);
            while (done < len[k]) {
                if (have == 0) {
                    cur = *w++;
                    have = 4;
                }
                dst[done++] = (uint8_t)cur;
                cur >>= 8;
                --have;
            }
        }
    }
    __syncthreads();
    uint8_t *gout = out + (base - phase);  // 16-byte aligned
    const uint32_t end = phase + tile_bytes;
    for (uint32_t o = threadIdx.x * 16; o < end; o += kBlock * 16) {
        if (o >= phase && o + 16 <= end) {
            stream_store4(gout + o, *reinterpret_cast<const uint4 *>(s_stage + o));
        } else {  // first / last chunk is shared with the neighbouring tile: only this tile's bytes
            for (uint32_t c = (o < phase ? phase : o); c < o + 16 && c < end; ++c) gout[c] = s_stage[c];
        }
    }
}

// ---- the same emit for LONG values (an auction's description: ~75 bytes a value, 77 KB a tile): no staging of bytes at all.  LDS holds the
// values' END positions in the tile's output window, their source offsets, and a map from every 16-byte chunk of the window to the value that
// holds its first byte -- written by the values themselves, ~5 two-byte stores each.  Every lane then makes whole 16-byte chunks of the OUTPUT:
// one read of the map, the piece of that value and the piece of the value behind it (each from the one or two ALIGNED 16-byte source chunks
// that hold it, realigned in registers), one split mask between the two, one aligned 16-byte store.  A chunk inside one value -- four of five
// at 75 bytes a value -- takes the same path with an empty second piece; third and further pieces (short values in a long column) are merged
// one by one.  Neighbouring lanes read neighbouring source chunks of the same value and write neighbouring output chunks.
// History (arch/ops/join.sql, 2.3e7 joined bids, 1.7 GB of descriptions): byte-wise copies through utf8_emit_tile_at's stage in rounds, 1.24 ms;
// chunk-centric with a binary search for the chunk's value and range masks per piece, 0.93 ms and ALU-bound -- a wave64 instruction occupies its
// SIMD16 for four cycles, and ~250 instructions per chunk were 0.8 ms of every SIMD's time whatever the memory side did (ablations: no source
// loads 0.83 ms; no stores 1.20 with four chunks in flight per lane; 2 / 4 chunks in flight: 1.00 / 1.25); a lane per VALUE (every interior
// chunk one unaligned 16-byte load, four funnel shifts, one store) 3.8 ms with non-temporal stores, 1.48 with plain ones -- 64 lanes storing 16
// bytes each 75 bytes apart cost the texture path a line per lane.  A kernel of its own: the short values' kernel -- q3's and q8's names --
// runs eight workgroups per CU on 59 VGPRs.
constexpr int kLongMapChunks = 8192;   // 16 KB of LDS: the chunk -> value map of a 128 KB output window
__global__ __launch_bounds__(kBlock) void utf8_emit_long_kernel(const int32_t *__restrict__ src_off, const uint8_t *__restrict__ src, const int32_t *__restrict__ rows,
                                                                int64_t n, const uint32_t *__restrict__ counts, const uint64_t *__restrict__ tile_base,
                                                                const uint64_t *__restrict__ col_base, int32_t *__restrict__ out_off, uint8_t *__restrict__ out) {
    // (col_base: null, or where this column's bytes start in a scan that runs over several columns -- gather_utf8_multi_*)
    __shared__ uint32_t s_end_[kLenTile];
    __shared__ int32_t s_b_[kLenTile];
    __shared__ uint16_t s_first[kLongMapChunks];   // chunk of the output window -> the value that holds its first in-tile byte
    __shared__ uint32_t s_it[kLenItems * kWavesPerBlock];  // bytes of (iteration, wave)
    uint32_t *s_end = s_end_;   // [kLenTile]  value v's end in the window (window byte i = output byte (base - phase) + i)
    int32_t *s_b = s_b_;        // [kLenTile]  value v's first byte in `src`
    const uint64_t base = tile_base[blockIdx.x] - (col_base ? *col_base : 0);
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int64_t i0 = (int64_t)blockIdx.x * kLenTile + threadIdx.x;
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)blockIdx.x * kWavesPerBlock);
    const uint32_t tile_bytes = wc.x + wc.y + wc.z + wc.w;
    int32_t b[kLenItems];
    uint32_t len[kLenItems], excl[kLenItems];
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
        b[k] = 0;
        len[k] = 0;
        if (i0 + k * kBlock < n) {
            const int2 o = load_off_pair(src_off, rows[i0 + k * kBlock]);
            b[k] = o.x;
            len[k] = (uint32_t)(o.y - o.x);
        }
        const uint32_t incl = wave_incl_scan_u32(len[k]);
        excl[k] = incl - len[k];
        if (lane == 63) s_it[k * kWavesPerBlock + wave] = incl;
    }
    __syncthreads();
    uint32_t run = 0;   // byte offset of (iteration k, wave) inside the tile: values are ordered iteration-major, then thread
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) {
            if (w == wave) excl[k] += run;
            run += s_it[k * kWavesPerBlock + w];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out_off[0] = 0;
#pragma unroll
    for (int k = 0; k < kLenItems; ++k)
        if (i0 + k * kBlock < n) stream_store(&out_off[i0 + k * kBlock + 1], (int32_t)(base + excl[k] + len[k]));
    if (tile_bytes == 0) return;
    const uint32_t phase = (uint32_t)(base & 15);  // window byte i holds output byte (base - phase) + i
#pragma unroll
    for (int k = 0; k < kLenItems; ++k) {
        s_end[k * kBlock + threadIdx.x] = phase + excl[k] + len[k];
        s_b[k * kBlock + threadIdx.x] = b[k];
    }
    const uint32_t end = phase + tile_bytes;
    uint8_t *gout = out + (base - phase);  // 16-byte aligned
    const uintptr_t sbase = reinterpret_cast<uintptr_t>(src);
    // bytes 0 .. x - 1 of a dword, x clamped to 0 .. 4 (no branches: a clamp, a 64-bit shift whose low word runs empty at x = 4, a complement)
    auto low_mask = [](int32_t x) -> uint32_t {
        const uint32_t c = (uint32_t)min(max(x, 0), 4);
        return ~(uint32_t)(0xffffffffull << (8u * c));
    };
    // sixteen chunk bytes from address A on (the byte at A lands on chunk byte 0; only chunk bytes a .. bnd - 1 are the value's -- the rest comes
    // back as whatever the two aligned chunks around them hold, or zero).  An aligned 16-byte chunk that holds at least one byte of the value
    // never crosses a page: only such chunks are read.
    auto realigned = [&](uintptr_t A, uint32_t a, uint32_t bnd, uint32_t (&V4)[4]) {
        const uint32_t sh = (uint32_t)(A & 15), qd = sh >> 2, bs = (sh & 3) * 8;
        const uint4 *q = reinterpret_cast<const uint4 *>(A & ~uintptr_t(15));
        uint4 c0 = make_uint4(0, 0, 0, 0), c1 = make_uint4(0, 0, 0, 0);
        if (bnd > a && sh + a < 16u) c0 = q[0];
        if (bnd > a && sh + bnd > 16u) c1 = q[1];
        // the window of five dwords that starts `qd` dwords into the eight: two rounds of bit-selects (by one dword, by two) -- a lane's qd is
        // its own, and the four-way choice written with ?: came back from the compiler as four divergent copies of everything behind it
        const uint32_t W[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const uint32_t by1 = 0u - (qd & 1u), by2 = 0u - (qd >> 1);
        uint32_t X[7], V[5];
#pragma unroll
        for (int j = 0; j < 7; ++j) X[j] = (W[j + 1] & by1) | (W[j] & ~by1);
#pragma unroll
        for (int j = 0; j < 5; ++j) V[j] = (X[j + 2] & by2) | (X[j] & ~by2);
#pragma unroll
        for (int i = 0; i < 4; ++i) V4[i] = __funnelshift_r(V[i], V[i + 1], bs);
    };
    const uint32_t n_chunks = (end + 15u) >> 4;
    for (uint32_t cbase = 0; cbase < n_chunks; cbase += (uint32_t)kLongMapChunks) {   // (one round for tiles up to 128 KB: 128 bytes a value)
        const uint32_t cend = cbase + (uint32_t)kLongMapChunks < n_chunks ? cbase + (uint32_t)kLongMapChunks : n_chunks;
        // ---- every value names itself in the chunks whose first in-tile byte it holds: from the first 16-byte boundary at or behind its start
        // (the tile's very first chunk for the value that starts the tile) to the chunk of its last byte
        __syncthreads();   // (the lists above are complete; the previous round's map has been read)
#pragma unroll
        for (int k = 0; k < kLenItems; ++k) {
            if (len[k] == 0) continue;
            const uint32_t p0 = phase + excl[k], p1 = p0 + len[k];
            uint32_t c = p0 == phase ? 0u : (p0 + 15u) >> 4;
            const uint32_t c_last = (p1 - 1u) >> 4;
            if (c < cbase) c = cbase;
            for (; c <= c_last && c < cend; ++c) s_first[c - cbase] = (uint16_t)(k * kBlock + threadIdx.x);
        }
        __syncthreads();
        // ---- every lane makes whole chunks of the output
        for (uint32_t ci = cbase + threadIdx.x; ci < cend; ci += kBlock) {
            const uint32_t o = ci << 4, c_lo = o < phase ? phase : o, chi = o + 16 < end ? o + 16 : end;
            uint32_t v = s_first[ci - cbase];
            // the value the chunk starts in, and the one behind it: two pieces with ONE split between them is what a chunk of long values holds
            const uint32_t p0 = v ? s_end[v - 1] : phase, p1 = s_end[v], hi1 = p1 < chi ? p1 : chi;
            const uint32_t w = v + 1 < (uint32_t)kLenTile ? v + 1 : v;
            const uint32_t q1 = s_end[w], hi2 = hi1 < chi ? (q1 < chi ? q1 : chi) : hi1;
            uint32_t P1[4], P2[4], acc[4];
            realigned(sbase + (uint32_t)s_b[v] + o - p0, c_lo - o, hi1 - o, P1);
            realigned(sbase + (uint32_t)s_b[w] + o - p1, hi1 - o, hi2 - o, P2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t m = low_mask((int32_t)(hi1 - o) - 4 * i);   // chunk bytes below the split are the first value's
                acc[i] = (P1[i] & m) | (P2[i] & ~m);
            }
            uint32_t pos = hi2;
            if (pos < chi) {   // (short values in a long column: a third piece and more, one by one under range masks)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] &= low_mask((int32_t)(pos - o) - 4 * i);
                for (v = w + 1; pos < chi; ++v) {
                    const uint32_t e = s_end[v], hi = e < chi ? e : chi;
                    if (hi <= pos) continue;   // (an empty value)
                    uint32_t P[4];
                    realigned(sbase + (uint32_t)s_b[v] + o - s_end[v - 1], pos - o, hi - o, P);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] |= P[i] & low_mask((int32_t)(hi - o) - 4 * i) & ~low_mask((int32_t)(pos - o) - 4 * i);
                    pos = hi;
                }
            }
            if (o >= phase && o + 16 <= end) {
                stream_store4(gout + o, make_uint4(acc[0], acc[1], acc[2], acc[3]));
            } else {  // the tile's first / last chunk is shared with the neighbouring tile: only this tile's bytes
                for (uint32_t c = c_lo; c < chi; ++c) gout[c] = (uint8_t)(acc[(c - o) >> 2] >> (8 * ((c - o) & 3)));
            }
        }
    }
}

__device__ __forceinline__ void utf8_emit_tile(const int32_t *__restrict__ src_off, const uint8_t *__restrict__ src,
                                               const int32_t *__restrict__ rows, int64_t n, const uint32_t *__restrict__ counts,
                                               const uint64_t *__restrict__ tile_base, uint64_t col_base, int32_t *__restrict__ out_off,
                                               uint8_t *__restrict__ out, uint8_t *s_stage, uint32_t *s_it) {
    utf8_emit_tile_at(src_off, src, rows, n, counts, tile_base[blockIdx.x] - col_base, out_off, out, s_stage, s_it);
}

__global__ __launch_bounds__(kBlock) void utf8_emit_kernel(const int32_t *__restrict__ src_off,
                                                           const uint8_t *__restrict__ src, const int32_t *__restrict__ rows,
                                                           int64_t n, const uint32_t *__restrict__ counts,
                                                           const uint64_t *__restrict__ tile_base,
                                                           int32_t *__restrict__ out_off, uint8_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kStageBytes];
    __shared__ uint32_t s_it[kLenItems * kWavesPerBlock];  // bytes of (iteration, wave)
    utf8_emit_tile(src_off, src, rows, n, counts, tile_base, 0, out_off, out, s_stage, s_it);
}
// column blockIdx.y of a multi-column gather
__global__ __launch_bounds__(kBlock) void utf8_emit_multi_kernel(Utf8Cols cols, const int32_t *__restrict__ rows, int64_t n,
                                                                 int64_t tiles_stride, const uint32_t *__restrict__ counts,
                                                                 const uint64_t *__restrict__ tile_base) {
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kStageBytes];
    __shared__ uint32_t s_it[kLenItems * kWavesPerBlock];
    const int c = blockIdx.y;
    const size_t shift = (size_t)c * tiles_stride;
    utf8_emit_tile(cols.src_off[c], cols.src[c], rows, n, counts + shift * kWavesPerBlock, tile_base + shift, tile_base[shift], cols.out_off[c],
                   cols.out[c], s_stage, s_it);
}

// column c alone (its neighbours went to the long-value kernel)
__global__ __launch_bounds__(kBlock) void utf8_emit_one_of_multi_kernel(Utf8Cols cols, int c, const int32_t *__restrict__ rows, int64_t n,
                                                                        int64_t tiles_stride, const uint32_t *__restrict__ counts,
                                                                        const uint64_t *__restrict__ tile_base) {
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kStageBytes];
    __shared__ uint32_t s_it[kLenItems * kWavesPerBlock];
    const size_t shift = (size_t)c * tiles_stride;
    utf8_emit_tile(cols.src_off[c], cols.src[c], rows, n, counts + shift * kWavesPerBlock, tile_base + shift, tile_base[shift], cols.out_off[c],
                   cols.out[c], s_stage, s_it);
}

// The multi-column emit WITHOUT a scan launch in front of it: workgroup (tile, column) sums the byte counts of the column's lower
// tiles itself (block_base_of_tile).  The byte buffers were sized from the PREVIOUS call's totals (the host has not seen this call's
// yet): a tile that would write past its column's capacity writes nothing and raises `h_over`; the last tile of a column reports the
// column's total -- both straight into pinned host memory, read after the call's one synchronisation.
struct Utf8Caps {
    uint64_t cap[kMaxUtf8Multi];
};
__global__ __launch_bounds__(kBlock) void utf8_emit_multi_self_kernel(Utf8Cols cols, const int32_t *__restrict__ rows, int64_t n,
                                                                      const uint64_t *__restrict__ d_n, int64_t tiles_stride,
                                                                      const uint32_t *__restrict__ counts, Utf8Caps caps,
                                                                      uint64_t *__restrict__ h_tot, uint32_t *__restrict__ h_over) {
    __shared__ __attribute__((aligned(16))) uint8_t s_stage[kStageBytes];
    __shared__ uint32_t s_it[kLenItems * kWavesPerBlock];
    __shared__ uint64_t s_red[kWavesPerBlock];
    if (d_n) n = min(n, (int64_t)*d_n);
    const int c = blockIdx.y;
    const uint32_t *col_counts = counts + (size_t)c * tiles_stride * kWavesPerBlock;
    const uint64_t base = block_base_of_tile(col_counts, (int32_t)blockIdx.x, s_red);
    const uint4 wc = *reinterpret_cast<const uint4 *>(col_counts + (size_t)blockIdx.x * kWavesPerBlock);
    const uint64_t mine = (uint64_t)wc.x + wc.y + wc.z + wc.w;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) h_tot[c] = base + mine;
    if (blockIdx.x == 0 && threadIdx.x == 0) cols.out_off[c][0] = 0;
    if (base + mine > caps.cap[c]) {   // (block-uniform) the hint was too small: the host redoes the take with exact sizes
        if (threadIdx.x == 0) __hip_atomic_store(h_over, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if ((int64_t)blockIdx.x * kLenTile >= n) return;   // tiles of the bound beyond the true row count
    utf8_emit_tile_at(cols.src_off[c], cols.src[c], rows, n, col_counts, base, cols.out_off[c], cols.out[c], s_stage, s_it);
}

// ---- in-place inclusive scan: tile sums -> tile scan -> apply -----------------------------------------------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;  // thread t: values t*8 .. t*8+7

// A thread's eight consecutive values as two 16-byte accesses at dword-aligned addresses (all the hardware asks of a global access; `data` may start
// anywhere: a column's offsets + 1).  Eight 4-byte accesses per thread put every lane of a wave instruction in a 32-byte segment of its own -- the
// texture path takes such an instruction segment by segment: the one-workgroup scan of 32768 values took 29 us that way, 5 us this way.
__device__ __forceinline__ void scan_load8(const int32_t *data, int64_t i0, int64_t n, uint32_t (&v)[kScanItems]) {
    static_assert(kScanItems == 8, "two 16-byte accesses per thread");
    if (i0 + kScanItems <= n) {
        uint4 a, b;
        __builtin_memcpy(&a, data + i0, 16);
        __builtin_memcpy(&b, data + i0 + 4, 16);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) v[k] = (i0 + k < n) ? (uint32_t)data[i0 + k] : 0u;
    }
}
// Adds the running position to the thread's eight values and stores them (inclusive scan)
__device__ __forceinline__ void scan_store8(int32_t *data, int64_t i0, int64_t n, const uint32_t (&v)[kScanItems], uint32_t pos) {
    uint32_t o[kScanItems];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        pos += v[k];
        o[k] = pos;
    }
    if (i0 + kScanItems <= n) {
        const uint4 a = make_uint4(o[0], o[1], o[2], o[3]), b = make_uint4(o[4], o[5], o[6], o[7]);
        __builtin_memcpy(data + i0, &a, 16);
        __builtin_memcpy(data + i0 + 4, &b, 16);
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k)
            if (i0 + k < n) data[i0 + k] = (int32_t)o[k];
    }
}

__global__ __launch_bounds__(kBlock) void scan_sum_kernel(const int32_t *__restrict__ data, int64_t n,
                                                          uint32_t *__restrict__ counts) {
    const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems], mine = 0;
    scan_load8(data, i0, n, v);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) mine += v[k];
    const uint32_t incl = wave_incl_scan_u32(mine);
    if (lane_id() == 63) counts[(size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)] = incl;
}

__global__ __launch_bounds__(kBlock) void scan_apply_kernel(int32_t *data, int64_t n, const uint32_t *__restrict__ counts,
                                                            const uint64_t *__restrict__ tile_base) {
    const int64_t i0 = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    const int wave = threadIdx.x >> 6;
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)blockIdx.x * kWavesPerBlock);
    uint32_t v[kScanItems], mine = 0;
    scan_load8(data, i0, n, v);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) mine += v[k];
    const uint32_t incl = wave_incl_scan_u32(mine);
    const uint64_t pos = tile_base[blockIdx.x] + (wave > 0 ? wc.x : 0u) + (wave > 1 ? wc.y : 0u) + (wave > 2 ? wc.z : 0u) + (incl - mine);
    scan_store8(data, i0, n, v, (uint32_t)pos);   // (an int32 scan: the positions wrap as the stored values do)
}


// The same scan in ONE launch for inputs of a few thousand values (a radix pass's histogram over a relation of a few hundred thousand rows, the
// offsets of a small Utf8 column): one workgroup of 1024 threads takes the input 8192 values a round (four rounds at most, all loaded up front) with a
// running carry -- one launch instead of three and two boundaries.
constexpr int kSmallScanBlock = 1024;
constexpr int64_t kSmallScanMax = 4 * kSmallScanBlock * kScanItems;   // 32768 values: four rounds
__global__ __launch_bounds__(kSmallScanBlock) void scan_small_kernel(int32_t *data, int64_t n) {
    __shared__ uint32_t s_wave[4][kSmallScanBlock / 64];
    const int wave = threadIdx.x >> 6, lane = lane_id();
    constexpr int kRounds = (int)(kSmallScanMax / ((int64_t)kSmallScanBlock * kScanItems));
    // every round's values are asked for up front (one memory round trip for the whole input, not one per round)
    uint32_t v[kRounds][kScanItems];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) scan_load8(data, ((int64_t)r * kSmallScanBlock + threadIdx.x) * kScanItems, n, v[r]);
    uint32_t mine[kRounds], incl[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        mine[r] = 0;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) mine[r] += v[r][k];
        incl[r] = wave_incl_scan_u32(mine[r]);
        if (lane == 63) s_wave[r][wave] = incl[r];
    }
    __syncthreads();
    uint32_t carry = 0;
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const uint32_t wt = lane < kSmallScanBlock / 64 ? s_wave[r][lane] : 0u;   // (every wave scans the sixteen wave totals itself)
        const uint32_t wincl = wave_incl_scan_u32(wt);
        const uint32_t below = wave > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)wincl, wave - 1) : 0u;   // (wave is uniform in a wave)
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)wincl, kSmallScanBlock / 64 - 1);
        scan_store8(data, ((int64_t)r * kSmallScanBlock + threadIdx.x) * kScanItems, n, v[r], carry + below + incl[r] - mine[r]);
        carry += total;
    }
}

// ---- fill_words / publish_words (gather.hpp)
__global__ __launch_bounds__(kBlock) void fill_words_kernel(FillList f) {
    const uint64_t t0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x, stride = (uint64_t)gridDim.x * kBlock;
    for (int r = 0; r < f.n; ++r) {
        uint32_t *p = static_cast<uint32_t *>(f.p[r]);
        const uint32_t v = f.v[r];
        const uint64_t n = f.words[r];
        // the 16-byte aligned middle with 16-byte stores, the ragged ends word by word
        const uint64_t head = std::min<uint64_t>(n, (4 - ((reinterpret_cast<uintptr_t>(p) >> 2) & 3)) & 3);
        const uint64_t n4 = (n - head) >> 2;
        if (t0 < head) p[t0] = v;
        uint4 *q = reinterpret_cast<uint4 *>(p + head);
        for (uint64_t i = t0; i < n4; i += stride) q[i] = make_uint4(v, v, v, v);
        const uint64_t tail0 = head + 4 * n4;
        if (t0 < n - tail0) p[tail0 + t0] = v;
    }
}
__global__ void publish_words_kernel(PublishList l) {
    for (int r = 0; r < l.n; ++r)
        if ((int)threadIdx.x < l.words[r]) static_cast<uint32_t *>(l.h[r])[threadIdx.x] = static_cast<const uint32_t *>(l.d[r])[threadIdx.x];
}

}  // namespace

namespace flockgpu {

int launch_tile_scan(flockgpu_ctx *ctx, const uint32_t *counts, int32_t n_tiles, uint64_t *tile_base,
                     const int32_t *tile_first, int32_t n_seg, int64_t *seg_out_off) {
    constexpr int32_t kChunk = kScanRounds * kScanBlock;
    if (n_tiles > kChunk) {
        const int32_t n_chunks = (int32_t)div_up(n_tiles, kChunk);
        uint64_t *chunk_total = nullptr, *chunk_base = nullptr;
        FG_TRY(arena_get_t(ctx, "scan.chunk_total", (size_t)n_chunks + 1, &chunk_total));
        FG_TRY(arena_get_t(ctx, "scan.chunk_base", (size_t)n_chunks + 1, &chunk_base));
        {
            LaunchScope ls(ctx, "tile_scan_kernel");  // (reported under one name: the three passes of a chunked scan)
            hipLaunchKernelGGL(tile_scan_chunk_kernel, dim3((unsigned)n_chunks), dim3(kScanBlock), 0, ctx->stream, counts, n_tiles,
                               tile_base, chunk_total);
            hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(kScanBlock), 0, ctx->stream, chunk_total, n_chunks, chunk_base);
            hipLaunchKernelGGL(tile_scan_fix_kernel, dim3((unsigned)div_up((int64_t)n_tiles + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                               tile_base, n_tiles, chunk_base, n_chunks);
            if (seg_out_off)
                hipLaunchKernelGGL(seg_offsets_kernel, dim3((unsigned)div_up((int64_t)n_seg + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                                   tile_base, n_tiles, tile_first, n_seg, seg_out_off);
        }
        return check_launch(ctx, "tile_scan (chunked)");
    }
    {
        LaunchScope ls(ctx, "tile_scan_kernel");
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(kScanBlock), 0, ctx->stream, counts, n_tiles, tile_base,
                           tile_first, n_seg, seg_out_off);
    }
    return check_launch(ctx, "tile_scan_kernel");
}

int emit_flagged_rows(flockgpu_ctx *ctx, const SegTiles &st, const uint32_t *flag_words, const uint32_t *counts,
                      const uint64_t *tile_base, int32_t *out_rows) {
    if (st.n_tiles <= 0) return FLOCKGPU_OK;
    {
        LaunchScope ls(ctx, "emit_rows_kernel");
        hipLaunchKernelGGL(emit_rows_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, st, flag_words,
                           counts, tile_base, out_rows);
    }
    return check_launch(ctx, "emit_rows_kernel");
}

int emit_flagged_rows_self(flockgpu_ctx *ctx, const SegTiles &st, const uint32_t *flag_words, const uint32_t *counts, int32_t *out_rows, int64_t *h_off) {
    if (st.n_seg != 1 || st.n_tiles <= 0 || st.n_tiles > kSelfScanMaxTiles) return fail(ctx, FLOCKGPU_ERR_INVALID, "emit_flagged_rows_self: one segment of 1 .. %d tiles", kSelfScanMaxTiles);
    {
        LaunchScope ls(ctx, "emit_rows_kernel");
        hipLaunchKernelGGL(emit_rows_self_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, st, flag_words, counts, out_rows, h_off);
    }
    return check_launch(ctx, "emit_rows_self_kernel");
}

int emit_flagged_bids(flockgpu_ctx *ctx, const SegTiles &st, const uint32_t *flag_words, const uint32_t *counts,
                      const uint64_t *tile_base, const flockgpu_bid_cols &bid, int32_t *o_auction, int32_t *o_price,
                      int32_t *o_bidder, int64_t *o_time) {
    if (st.n_tiles <= 0) return FLOCKGPU_OK;
    {
        LaunchScope ls(ctx, "emit_bids_kernel");
        hipLaunchKernelGGL(emit_bids_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, bid.auction, bid.price,
                           bid.bidder, bid.b_date_time, st, flag_words, counts, tile_base, o_auction, o_price, o_bidder, o_time);
    }
    return check_launch(ctx, "emit_bids_kernel");
}

int segment_key_stats(flockgpu_ctx *ctx, const int32_t *col, int64_t n_rows, const SegTiles &st, int32_t *d_min,
                      int32_t *d_max, int32_t *d_sorted) {
    if (st.n_seg <= 0) return FLOCKGPU_OK;
    hipLaunchKernelGGL(fill_stats_kernel, dim3((unsigned)div_up(st.n_seg, kBlock)), dim3(kBlock), 0, ctx->stream, d_min,
                       d_max, d_sorted, st.n_seg);
    FG_TRY(check_launch(ctx, "fill_stats_kernel"));
    if (st.n_tiles <= 0) return FLOCKGPU_OK;
    {
        LaunchScope ls(ctx, "segment_stats_kernel");
        hipLaunchKernelGGL(segment_stats_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, col, n_rows, st,
                           d_min, d_max, d_sorted);
    }
    return check_launch(ctx, "segment_stats_kernel");
}

int inclusive_scan_i32(flockgpu_ctx *ctx, const char *name, int32_t *data, int64_t n) {
    if (n <= 0) return FLOCKGPU_OK;
    if (n <= kSmallScanMax) {
        LaunchScope ls(ctx, "scan_small_kernel");
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(kSmallScanBlock), 0, ctx->stream, data, n);
        return check_launch(ctx, "scan_small_kernel");
    }
    const int64_t tiles = div_up(n, kScanTile);
    if (tiles > 0x7fffffff / 2) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: too many tiles", name);
    const std::string k_c = std::string(name) + ".counts", k_b = std::string(name) + ".base";
    uint32_t *counts = nullptr;
    uint64_t *tile_base = nullptr;
    FG_TRY(arena_get_t(ctx, k_c.c_str(), (size_t)tiles * kWavesPerBlock, &counts));
    FG_TRY(arena_get_t(ctx, k_b.c_str(), (size_t)tiles + 1, &tile_base));
    {
        LaunchScope ls(ctx, "scan_sum_kernel");
        hipLaunchKernelGGL(scan_sum_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream, data, n, counts);
    }
    FG_TRY(check_launch(ctx, "scan_sum_kernel"));
    FG_TRY(launch_tile_scan(ctx, counts, (int32_t)tiles, tile_base, nullptr, 0, nullptr));
    {
        LaunchScope ls(ctx, "scan_apply_kernel");
        hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream, data, n, counts, tile_base);
    }
    return check_launch(ctx, "scan_apply_kernel");
}

int gather_i32(flockgpu_ctx *ctx, const int32_t *src, const int32_t *rows, int64_t n, int32_t *out) {
    if (n <= 0) return FLOCKGPU_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(n, kBlock), (int64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "gather_i32_kernel");
        hipLaunchKernelGGL(gather_i32_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, src, rows, n, out);
    }
    return check_launch(ctx, "gather_i32_kernel");
}

int gather_i64(flockgpu_ctx *ctx, const int64_t *src, const int32_t *rows, int64_t n, int64_t *out) {
    if (n <= 0) return FLOCKGPU_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(n, kBlock), (int64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "gather_i64_kernel");
        hipLaunchKernelGGL(gather_i64_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, src, rows, n, out);
    }
    return check_launch(ctx, "gather_i64_kernel");
}

int gather_fixed_multi(flockgpu_ctx *ctx, const GatherCols &cols, const int32_t *rows, int64_t n);
// ---- fixed-width take at rows in NO order (ORDER BY's result): the columns are first interleaved into 16-byte records -- a streaming pass --, so that
// the take reads ONE 16-byte record per row at a random position instead of one 32-byte sector per row and column (sort.sql's auction, price and
// b_date_time: 4 + 4 + 8 bytes; the sorted key column is not taken at all, plan.hip exec_sort).
struct PackedCols {
    const void *src[4];
    void *out[4];
    int32_t width[4];    // 4 | 8
    int32_t offset[4];   // byte offset inside the record (8-byte fields first)
    int32_t n = 0;
};
__global__ __launch_bounds__(kBlock) void pack_records_kernel(PackedCols cols, int64_t n, uint8_t *__restrict__ rec) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        // the record is made in registers and leaves as ONE 16-byte store per lane (fields stored one by one were 4-byte pieces 16 bytes apart: 0.70 ms
        // for 9.2e7 rows); which word a field lands in is the same for every lane
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        for (int c = 0; c < cols.n; ++c) {   // (uniform trip count)
            uint32_t lo, hi = 0;
            if (cols.width[c] == 4) lo = static_cast<const uint32_t *>(cols.src[c])[i];
            else {
                const uint64_t v = static_cast<const uint64_t *>(cols.src[c])[i];
                lo = (uint32_t)v;
                hi = (uint32_t)(v >> 32);
            }
            const int at = cols.offset[c] >> 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (at == k) w[k] = lo;
                if (cols.width[c] == 8 && at + 1 == k) w[k] = hi;
            }
        }
        stream_store4(rec + i * 16, make_uint4(w[0], w[1], w[2], w[3]));
    }
}
__device__ __forceinline__ uint32_t record_word(const uint4 &r, int w) { return w == 0 ? r.x : w == 1 ? r.y : w == 2 ? r.z : r.w; }   // (w is wave-uniform)
__global__ __launch_bounds__(kBlock) void gather_records_kernel(PackedCols cols, const uint4 *__restrict__ rec, const int32_t *__restrict__ rows, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < n4; q += (int64_t)gridDim.x * kBlock) {
        const int4 r = *reinterpret_cast<const int4 *>(rows + q * 4);   // (the list is an arena buffer: 16-byte aligned)
        const uint4 a = rec[r.x], b = rec[r.y], d = rec[r.z], e = rec[r.w];   // four records in flight per lane
        for (int c = 0; c < cols.n; ++c) {
            const int w = cols.offset[c] >> 2;
            if (cols.width[c] == 4) {
                *reinterpret_cast<uint4 *>(static_cast<uint32_t *>(cols.out[c]) + q * 4) = make_uint4(record_word(a, w), record_word(b, w), record_word(d, w), record_word(e, w));
            } else {
                uint32_t *o = reinterpret_cast<uint32_t *>(static_cast<int64_t *>(cols.out[c]) + q * 4);
                *reinterpret_cast<uint4 *>(o) = make_uint4(record_word(a, w), record_word(a, w + 1), record_word(b, w), record_word(b, w + 1));
                *reinterpret_cast<uint4 *>(o + 4) = make_uint4(record_word(d, w), record_word(d, w + 1), record_word(e, w), record_word(e, w + 1));
            }
        }
    }
    const int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x;   // the up to three rows behind the last whole group of four
    if (blockIdx.x == 0 && i < n) {
        const uint4 a = rec[rows[i]];
        for (int c = 0; c < cols.n; ++c) {
            const int w = cols.offset[c] >> 2;
            if (cols.width[c] == 4) static_cast<uint32_t *>(cols.out[c])[i] = record_word(a, w);
            else static_cast<uint64_t *>(cols.out[c])[i] = (uint64_t)record_word(a, w) | ((uint64_t)record_word(a, w + 1) << 32);
        }
    }
}

int gather_fixed_packed(flockgpu_ctx *ctx, const char *name, const GatherCols &cols, int64_t in_rows, const int32_t *rows, int64_t n) {
    int total = 0;
    for (int c = 0; c < cols.n; ++c) total += cols.width[c];
    bool ok = cols.n >= 2 && cols.n <= 4 && total <= 16 && n >= (int64_t(1) << 20) && n * 2 >= in_rows && (reinterpret_cast<uintptr_t>(rows) & 15) == 0;
    for (int c = 0; c < cols.n && ok; ++c) ok = (reinterpret_cast<uintptr_t>(cols.out[c]) & 15) == 0 && (cols.width[c] == 4 || cols.width[c] == 8);   // (four rows per lane in the packing pass -- 16-byte column loads, four record stores 64 bytes apart per lane -- ran at 1.47 ms against 0.61: stores of one instruction must be neighbours)
    if (!ok) return gather_fixed_multi(ctx, cols, rows, n);   // (few rows of many, one column, wide rows: the plain take)
    PackedCols p;
    p.n = cols.n;
    int at = 0, k = 0;
    for (int pass = 0; pass < 2; ++pass)   // 8-byte fields first: every field aligned to its width
        for (int c = 0; c < cols.n; ++c)
            if ((cols.width[c] == 8) == (pass == 0)) {
                p.src[k] = cols.src[c];
                p.out[k] = cols.out[c];
                p.width[k] = cols.width[c];
                p.offset[k] = at;
                at += cols.width[c];
                ++k;
            }
    uint8_t *rec = nullptr;
    FG_TRY(arena_get_t(ctx, (std::string(name) + ".rec").c_str(), ((size_t)in_rows + 1) * 16, &rec));
    {
        LaunchScope ls(ctx, "pack_records_kernel");
        const unsigned blocks = (unsigned)std::min<int64_t>(div_up(in_rows, (int64_t)kBlock), (int64_t)ctx->num_cus * 16);
        hipLaunchKernelGGL(pack_records_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, p, in_rows, rec);
    }
    FG_TRY(check_launch(ctx, "pack_records_kernel"));
    {
        LaunchScope ls(ctx, "gather_records_kernel");
        const unsigned blocks = (unsigned)std::min<int64_t>(div_up(div_up(n, 4), kBlock), (int64_t)ctx->num_cus * 8);
        hipLaunchKernelGGL(gather_records_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, p, reinterpret_cast<const uint4 *>(rec), rows, n);
    }
    return check_launch(ctx, "gather_records_kernel");
}

int gather_fixed_multi(flockgpu_ctx *ctx, const GatherCols &cols, const int32_t *rows, int64_t n) {
    if (n <= 0 || cols.n <= 0) return FLOCKGPU_OK;
    if (cols.n > kGatherMulti) return fail(ctx, FLOCKGPU_ERR_INVALID, "gather_fixed_multi: more than %d columns", kGatherMulti);
    bool aligned = (reinterpret_cast<uintptr_t>(rows) & 15) == 0;
    for (int c = 0; c < cols.n; ++c) aligned = aligned && (reinterpret_cast<uintptr_t>(cols.out[c]) & 15) == 0;
    if (!aligned) {   // (a row list or an output that is not a buffer of its own: column by column, as before)
        for (int c = 0; c < cols.n; ++c) {
            if (cols.width[c] == 4) FG_TRY(gather_i32(ctx, static_cast<const int32_t *>(cols.src[c]), rows, n, static_cast<int32_t *>(cols.out[c])));
            else FG_TRY(gather_i64(ctx, static_cast<const int64_t *>(cols.src[c]), rows, n, static_cast<int64_t *>(cols.out[c])));
        }
        return FLOCKGPU_OK;
    }
    const unsigned blocks = (unsigned)std::min<int64_t>(div_up(div_up(n, 4), kBlock), (int64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "gather_multi_kernel");
        hipLaunchKernelGGL(gather_multi_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, cols, rows, n);
    }
    return check_launch(ctx, "gather_multi_kernel");
}

int gather_utf8_begin(flockgpu_ctx *ctx, const char *name, const flockgpu_utf8 &src, const int32_t *rows, int64_t n,
                      Utf8Gather *g, const uint64_t *d_n) {
    g->name = name;
    g->src = src;
    g->rows = rows;
    g->n = n;
    g->tiles = n > 0 ? div_up(n, kLenTile) : 0;
    const std::string k_t = g->name + ".total";
    FG_TRY(pinned_get_t(ctx, k_t.c_str(), 1, &g->h_total));
    *g->h_total = 0;
    if (n <= 0) return FLOCKGPU_OK;
    if (g->tiles > 0x7fffffff / 2) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: too many tiles", name);
    const std::string k_c = g->name + ".counts", k_b = g->name + ".base";
    FG_TRY(arena_get_t(ctx, k_c.c_str(), (size_t)g->tiles * kWavesPerBlock, &g->counts));
    FG_TRY(arena_get_t(ctx, k_b.c_str(), (size_t)g->tiles + 1, &g->tile_base));
    {
        LaunchScope ls(ctx, "utf8_len_kernel");
        hipLaunchKernelGGL(utf8_len_kernel, dim3((unsigned)g->tiles), dim3(kBlock), 0, ctx->stream, src.offsets, rows, n, d_n,
                           g->counts);
    }
    FG_TRY(check_launch(ctx, "utf8_len_kernel"));
    FG_TRY(launch_tile_scan(ctx, g->counts, (int32_t)g->tiles, g->tile_base, nullptr, 0, nullptr));
    return publish_words(ctx, PublishList().add(g->h_total, g->tile_base + g->tiles, 2));   // (one small kernel instead of a copy call)
}

void gather_utf8_narrow(Utf8Gather *g, int64_t n) {
    g->n = n;
    g->tiles = n > 0 ? div_up(n, kLenTile) : 0;
}

static int gather_utf8_emit(flockgpu_ctx *ctx, const Utf8Gather &g, uint64_t total, flockgpu_utf8 *out, int64_t *n_bytes) {
    const std::string k_off = g.name + ".off", k_bytes = g.name + ".bytes";
    int32_t *o_off = nullptr;
    uint8_t *o_b = nullptr;
    FG_TRY(arena_get_t(ctx, k_off.c_str(), (size_t)std::max<int64_t>(g.n, 0) + 1, &o_off));
    out->offsets = o_off;
    out->data = nullptr;
    *n_bytes = 0;
    if (g.n <= 0) {
        FG_HIP(ctx, hipMemsetAsync(o_off, 0, sizeof(int32_t), ctx->stream));
        FG_TRY(arena_get_t(ctx, k_bytes.c_str(), 16, &o_b));
        out->data = o_b;
        return FLOCKGPU_OK;
    }
    if (total > 0x7fffffffull)
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: gathered Utf8 column exceeds 2^31 bytes (Arrow Utf8 offsets are int32)",
                    g.name.c_str());
    FG_TRY(arena_get_t(ctx, k_bytes.c_str(), (size_t)total + 16, &o_b));
    if (total > (uint64_t)g.n * (kStageBytes / kLenTile)) {   // values longer on average than the stage holds per value: the chunk-wise kernel
        LaunchScope ls(ctx, "utf8_emit_long_kernel");
        hipLaunchKernelGGL(utf8_emit_long_kernel, dim3((unsigned)g.tiles), dim3(kBlock), 0, ctx->stream, g.src.offsets, g.src.data,
                           g.rows, g.n, g.counts, g.tile_base, (const uint64_t *)nullptr, o_off, o_b);
    } else {
        LaunchScope ls(ctx, "utf8_emit_kernel");
        hipLaunchKernelGGL(utf8_emit_kernel, dim3((unsigned)g.tiles), dim3(kBlock), 0, ctx->stream, g.src.offsets, g.src.data,
                           g.rows, g.n, g.counts, g.tile_base, o_off, o_b);
    }
    FG_TRY(check_launch(ctx, "utf8_emit_kernel"));
    out->data = o_b;
    *n_bytes = (int64_t)total;
    return FLOCKGPU_OK;
}

int gather_utf8_finish(flockgpu_ctx *ctx, const Utf8Gather &g, flockgpu_utf8 *out, int64_t *n_bytes) {
    return gather_utf8_emit(ctx, g, g.n > 0 ? *g.h_total : 0, out, n_bytes);
}

int gather_utf8_finish_known(flockgpu_ctx *ctx, Utf8Gather &g, int64_t total_bytes, flockgpu_utf8 *out) {
    int64_t nb = 0;
    return gather_utf8_emit(ctx, g, (uint64_t)std::max<int64_t>(total_bytes, 0), out, &nb);
}

int gather_utf8_multi_begin(flockgpu_ctx *ctx, const char *name, const flockgpu_utf8 *srcs, int k, const int32_t *rows, int64_t n,
                            Utf8MultiGather *g, const uint64_t *d_n) {
    if (k < 1 || k > kMaxUtf8Multi) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: 1 to %d Utf8 columns per gather", name, kMaxUtf8Multi);
    g->name = name;
    g->k = k;
    for (int c = 0; c < k; ++c) g->src[c] = srcs[c];
    g->rows = rows;
    g->n = n;
    g->tiles = g->tiles_stride = n > 0 ? div_up(n, kLenTile) : 0;
    FG_TRY(pinned_get_t(ctx, (g->name + ".totals").c_str(), (size_t)kMaxUtf8Multi + 1, &g->h_col_base));
    for (int c = 0; c <= kMaxUtf8Multi; ++c) g->h_col_base[c] = 0;
    if (n <= 0) return FLOCKGPU_OK;
    pinned_pending(g->h_col_base, k + 1);   // (utf8_col_bases_kernel writes each of them once: gather_utf8_multi_wait)
    const int64_t all = g->tiles_stride * k;
    if (all > 0x7fffffff / 2) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: too many tiles", name);
    uint64_t *d_col_base = nullptr;
    FG_TRY(arena_get_t(ctx, (g->name + ".counts").c_str(), (size_t)all * kWavesPerBlock, &g->counts));
    FG_TRY(arena_get_t(ctx, (g->name + ".base").c_str(), (size_t)all + 1, &g->tile_base));
    FG_TRY(arena_get_t(ctx, (g->name + ".totals").c_str(), (size_t)kMaxUtf8Multi + 1, &d_col_base));
    Utf8Cols cols{};
    cols.k = k;
    for (int c = 0; c < k; ++c) cols.src_off[c] = srcs[c].offsets;
    {
        LaunchScope ls(ctx, "utf8_len_kernel");
        hipLaunchKernelGGL(utf8_len_multi_kernel, dim3((unsigned)g->tiles), dim3(kBlock), 0, ctx->stream, cols, rows, n, d_n, g->tiles_stride, g->counts);
    }
    FG_TRY(check_launch(ctx, "utf8_len_multi_kernel"));
    FG_TRY(launch_tile_scan(ctx, g->counts, (int32_t)all, g->tile_base, nullptr, 0, nullptr));
    hipLaunchKernelGGL(utf8_col_bases_kernel, dim3(1), dim3(64), 0, ctx->stream, g->tile_base, g->tiles_stride, k, d_col_base, g->h_col_base);
    return check_launch(ctx, "utf8_col_bases_kernel");
}

int gather_utf8_multi_wait(flockgpu_ctx *ctx, const Utf8MultiGather &g) {
    if (g.n <= 0) return FLOCKGPU_OK;   // (nothing was queued)
    return wait_pinned(ctx, g.h_col_base, g.k + 1);
}

int gather_utf8_multi_run_bytes(flockgpu_ctx *ctx, const Utf8MultiGather &g, const int64_t *d_run_start, int n_runs, unsigned long long *const *d_run_bytes) {
    if (n_runs <= 0) return FLOCKGPU_OK;
    if (g.n <= 0) {
        for (int c = 0; c < g.k; ++c) FG_HIP(ctx, hipMemsetAsync(d_run_bytes[c], 0, sizeof(unsigned long long) * (size_t)n_runs, ctx->stream));
        return FLOCKGPU_OK;
    }
    Utf8Cols cols{};
    cols.k = g.k;
    RunByteOut out{};
    for (int c = 0; c < g.k; ++c) {
        cols.src_off[c] = g.src[c].offsets;
        out.run_bytes[c] = d_run_bytes[c];
    }
    LaunchScope ls(ctx, "utf8_run_bytes_kernel");
    hipLaunchKernelGGL(utf8_run_bytes_kernel, dim3((unsigned)n_runs, (unsigned)g.k), dim3(kBlock), 0, ctx->stream, cols, g.rows, d_run_start, g.tile_base, g.tiles_stride, out);
    return check_launch(ctx, "utf8_run_bytes_kernel");
}

void gather_utf8_multi_narrow(Utf8MultiGather *g, int64_t n) {
    g->n = n;
    g->tiles = n > 0 ? div_up(n, kLenTile) : 0;  // (the counts keep the stride of the bound they were laid out for)
}

// known_bytes (may be null): the columns' byte totals when the caller knows them without asking the device
int gather_utf8_multi_finish(flockgpu_ctx *ctx, const Utf8MultiGather &g, flockgpu_utf8 *outs, int64_t *n_bytes, const int64_t *known_bytes) {
    Utf8Cols cols{};
    cols.k = g.k;
    for (int c = 0; c < g.k; ++c) {
        const std::string k_off = g.name + ".off" + std::to_string(c), k_bytes = g.name + ".bytes" + std::to_string(c);
        int32_t *o_off = nullptr;
        uint8_t *o_b = nullptr;
        const uint64_t total = g.n <= 0 ? 0 : (known_bytes ? (uint64_t)known_bytes[c] : g.h_col_base[c + 1] - g.h_col_base[c]);
        if (total > 0x7fffffffull)
            return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: gathered Utf8 column exceeds 2^31 bytes (Arrow Utf8 offsets are int32)", g.name.c_str());
        FG_TRY(arena_get_t(ctx, k_off.c_str(), (size_t)std::max<int64_t>(g.n, 0) + 1, &o_off));
        FG_TRY(arena_get_t(ctx, k_bytes.c_str(), (size_t)total + 16, &o_b));
        if (g.n <= 0) FG_HIP(ctx, hipMemsetAsync(o_off, 0, sizeof(int32_t), ctx->stream));
        cols.src_off[c] = g.src[c].offsets;
        cols.src[c] = g.src[c].data;
        cols.out_off[c] = o_off;
        cols.out[c] = o_b;
        outs[c] = flockgpu_utf8{o_off, o_b};
        n_bytes[c] = (int64_t)total;
    }
    if (g.n <= 0) return FLOCKGPU_OK;
    // columns whose values are longer on average than the stage holds per value (an auction's description) go to the chunk-wise kernel, one
    // launch each; the others stay together
    bool any_long = false, any_short = false;
    for (int c = 0; c < g.k; ++c) ((uint64_t)n_bytes[c] > (uint64_t)g.n * (kStageBytes / kLenTile) ? any_long : any_short) = true;
    if (any_long) {
        for (int c = 0; c < g.k; ++c) {
            const size_t shift = (size_t)c * (size_t)g.tiles_stride;
            if ((uint64_t)n_bytes[c] > (uint64_t)g.n * (kStageBytes / kLenTile)) {
                LaunchScope ls(ctx, "utf8_emit_long_kernel");
                hipLaunchKernelGGL(utf8_emit_long_kernel, dim3((unsigned)g.tiles), dim3(kBlock), 0, ctx->stream, cols.src_off[c], cols.src[c], g.rows, g.n,
                                   g.counts + shift * kWavesPerBlock, g.tile_base + shift, g.tile_base + shift, cols.out_off[c], cols.out[c]);
            } else {
                LaunchScope ls(ctx, "utf8_emit_kernel");
                hipLaunchKernelGGL(utf8_emit_one_of_multi_kernel, dim3((unsigned)g.tiles), dim3(kBlock), 0, ctx->stream, cols, c, g.rows, g.n, g.tiles_stride, g.counts, g.tile_base);
            }
        }
        (void)any_short;
        return check_launch(ctx, "utf8_emit_long_kernel");
    }
    {
        LaunchScope ls(ctx, "utf8_emit_kernel");
        hipLaunchKernelGGL(utf8_emit_multi_kernel, dim3((unsigned)g.tiles, (unsigned)g.k), dim3(kBlock), 0, ctx->stream, cols, g.rows, g.n, g.tiles_stride,
                           g.counts, g.tile_base);
    }
    return check_launch(ctx, "utf8_emit_multi_kernel");
}

int gather_utf8_multi_fast(flockgpu_ctx *ctx, const char *name, const flockgpu_utf8 *srcs, int k, const int32_t *rows, int64_t n_bound,
                           const uint64_t *d_n, const int64_t *cap_bytes, Utf8FastGather *g) {
    if (k < 1 || k > kMaxUtf8Multi) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: 1 to %d Utf8 columns per gather", name, kMaxUtf8Multi);
    g->name = name;
    g->k = k;
    g->tiles = n_bound > 0 ? div_up(n_bound, kLenTile) : 0;
    if (g->tiles > kSelfScanMaxTiles) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: too many tiles for the self-scanning take", name);
    uint64_t *h = nullptr;
    FG_TRY(pinned_get_t(ctx, (g->name + ".fast").c_str(), (size_t)kMaxUtf8Multi + 2, &h));
    g->h_tot = h;
    g->h_over = reinterpret_cast<uint32_t *>(h + kMaxUtf8Multi);
    for (int c = 0; c < kMaxUtf8Multi; ++c) h[c] = 0;      // (the previous call's values were read under its synchronisation)
    *g->h_over = 0;
    Utf8Cols cols{};
    Utf8Caps caps{};
    cols.k = k;
    for (int c = 0; c < k; ++c) {
        const std::string k_off = g->name + ".off" + std::to_string(c), k_bytes = g->name + ".bytes" + std::to_string(c);
        int32_t *o_off = nullptr;
        uint8_t *o_b = nullptr;
        FG_TRY(arena_get_t(ctx, k_off.c_str(), (size_t)std::max<int64_t>(n_bound, 0) + 1, &o_off));
        FG_TRY(arena_get_t(ctx, k_bytes.c_str(), (size_t)std::max<int64_t>(cap_bytes[c], 0) + 16, &o_b));
        cols.src_off[c] = srcs[c].offsets;
        cols.src[c] = srcs[c].data;
        cols.out_off[c] = o_off;
        cols.out[c] = o_b;
        caps.cap[c] = (uint64_t)std::max<int64_t>(cap_bytes[c], 0);
        g->out[c] = flockgpu_utf8{o_off, o_b};
    }
    if (g->tiles == 0) return FLOCKGPU_OK;
    uint32_t *counts = nullptr;
    FG_TRY(arena_get_t(ctx, (g->name + ".counts").c_str(), (size_t)g->tiles * k * kWavesPerBlock, &counts));
    {
        LaunchScope ls(ctx, "utf8_len_kernel");
        hipLaunchKernelGGL(utf8_len_multi_kernel, dim3((unsigned)g->tiles), dim3(kBlock), 0, ctx->stream, cols, rows, n_bound, d_n, g->tiles, counts);
    }
    FG_TRY(check_launch(ctx, "utf8_len_multi_kernel"));
    {
        LaunchScope ls(ctx, "utf8_emit_kernel");
        hipLaunchKernelGGL(utf8_emit_multi_self_kernel, dim3((unsigned)g->tiles, (unsigned)k), dim3(kBlock), 0, ctx->stream, cols, rows, n_bound, d_n, g->tiles, counts,
                           caps, h, g->h_over);
    }
    return check_launch(ctx, "utf8_emit_multi_self_kernel");
}

int gather_utf8(flockgpu_ctx *ctx, const char *name, const flockgpu_utf8 &src, const int32_t *rows, int64_t n,
                flockgpu_utf8 *out, int64_t *n_bytes) {
    Utf8Gather g;
    FG_TRY(gather_utf8_begin(ctx, name, src, rows, n, &g));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return gather_utf8_finish(ctx, g, out, n_bytes);
}

int fill_words(flockgpu_ctx *ctx, const FillList &f) {
    uint64_t most = 0;
    for (int r = 0; r < f.n; ++r) most = std::max(most, f.words[r]);
    if (!most) return FLOCKGPU_OK;
    const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(div_up((int64_t)most, (int64_t)kBlock * 4), 1), (uint64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "fill_words_kernel");
        hipLaunchKernelGGL(fill_words_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, f);
    }
    return check_launch(ctx, "fill_words_kernel");
}

int publish_words(flockgpu_ctx *ctx, const PublishList &l) {
    if (!l.n) return FLOCKGPU_OK;
    for (int r = 0; r < l.n; ++r)
        if (l.words[r] > 64) return fail(ctx, FLOCKGPU_ERR_INVALID, "publish_words: at most 64 words per run");
    {
        LaunchScope ls(ctx, "publish_words_kernel");
        hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, ctx->stream, l);
    }
    return check_launch(ctx, "publish_words_kernel");
}

}  // namespace flockgpu
