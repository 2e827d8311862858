// NEXMark q13 "bounded side-input join" for gfx950, per ElementWise window:
//   SELECT auction, bidder, price, b_date_time, value FROM bid JOIN side_input ON auction = key
// (benchmarks/src/nexmark/query/q13.sql, q13_plan.fmt; side_input schema flock/src/datasource/nexmark/event.rs:375-388;
//  the table itself is a user-supplied CSV, benchmarks/src/nexmark/main.rs:44,353-361 -- any keys, duplicates allowed).
//
// HBM-bound integer work, no MFMA.  The build side is small and static, the probe side is the whole bid stream:
//   build : multimap keyed `key` (one 64-bit CAS slot {key, head row} + chain array, hashtab.hpp), in global memory
//   probe : 16-wave workgroups walk the bid tiles b, b + G, ... with the next tile's keys prefetched; each first copies
//           the table's KEYS into LDS once (32-bit keys + one occupancy bit per slot: up to 16384 slots = 8192 side rows
//           in 66 KiB, two workgroups per CU) and probes there -- 4 B of HBM traffic per bid and no table traffic except on a hit; larger
//           tables are probed in global memory (L2 / MALL resident).  A key's duplicates are chained through `next[]`.
//   filter: when the side keys span <= 2^31 ids, an exact membership BITMAP over that range is tested first (one cached,
//           unconditional load per bid): a bid that cannot join never enters the divergent probe loop.
//   count -> scan -> emit: matching (bid row, side row) pairs in bid order, then the five output columns.
#include <algorithm>

#include "gather.hpp"
#include "hashtab.hpp"

using namespace flockgpu;

namespace {

constexpr int kProbeBlock = 1024;                 // 16 waves share one LDS copy of the table
constexpr int kLdsSlots = 16384;                  // keys 64 KiB + occupancy 2 KiB: two workgroups (32 waves) per CU
constexpr int kProbeTile = kFlagTile;             // 8192 bids per tile, 8 per lane at 1024 lanes
constexpr int kProbeWaves = kProbeBlock / 64;

__global__ __launch_bounds__(kBlock) void q13_build_kernel(const int32_t *__restrict__ key, int32_t n, uint64_t *table,
                                                           uint32_t cap, int32_t *next, uint32_t *err) {
    const int32_t i = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (i < n && !multimap_insert(table, cap, next, key[i], i)) atomicOr(err, 1u);
}

// Exact membership bitmap over [base, base + n_bits) of the side keys: one cached bit test rejects a bid that cannot join
// before any hash probing (divergent probe loops cost ~100 instructions per wave and row: 4 ms for 1e9 bids).
struct KeyBitmap {
    const uint32_t *words;  // nullptr: key range too wide, every bid is probed
    int32_t base;
    uint32_t n_bits;
};

__global__ __launch_bounds__(kBlock) void q13_bitmap_kernel(const int32_t *__restrict__ key, int32_t n, int32_t base, uint32_t *words) {
    const int32_t i = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (i < n) {
        const uint32_t idx = (uint32_t)key[i] - (uint32_t)base;
        atomicOr(&words[idx >> 5], 1u << (idx & 31));
    }
}

// Head row of `key`'s chain or -1, probing the GLOBAL table (tables too large for LDS).
__device__ __forceinline__ int32_t find_global(const uint64_t *__restrict__ tab, uint32_t cap, int32_t key) {
    uint32_t s = slot_of((uint32_t)key, cap);
#pragma unroll 1
    for (uint32_t probe = 0, lim = probe_limit(cap); probe < lim; ++probe) {
        const uint64_t cur = tab[s];
        if (cur == kEmpty64) return -1;
        if ((int32_t)(cur >> 32) == key) return (int32_t)(uint32_t)cur;
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    return -1;
}

// Same through the LDS copy: 32-bit keys + one occupancy bit per slot (any Int32 may be a key, so no key value can
// mark an empty slot); the head row is fetched from the global slot only on a hit.
__device__ __forceinline__ int32_t find_lds(const uint32_t *s_key, const uint32_t *s_occ, const uint64_t *__restrict__ tab,
                                            uint32_t cap, int32_t key) {
    uint32_t s = slot_of((uint32_t)key, cap);
#pragma unroll 1
    for (uint32_t probe = 0, lim = probe_limit(cap); probe < lim; ++probe) {
        if (!((s_occ[s >> 5] >> (s & 31)) & 1u)) return -1;
        if (s_key[s] == (uint32_t)key) return (int32_t)(uint32_t)tab[s];
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    return -1;
}

// Lane l of wave w holds rows  w*512 + it*256 + 4l .. 4l+3  (it = 0, 1) of the tile.
__device__ __forceinline__ bool q13_maybe(const KeyBitmap &bm, int32_t key) {
    if (!bm.words) return true;
    const uint32_t idx = (uint32_t)key - (uint32_t)bm.base;
    const bool in = idx < bm.n_bits;
    return in & ((bm.words[in ? idx >> 5 : 0u] >> (idx & 31)) & 1u);  // unconditional load from a clamped index
}

// COUNT pass: streams `auction` once (next tile's keys prefetched), tests the bitmap, probes the rows that
// may join and leaves, per lane, one byte of "this row has pairs" flags and, per wave, the number of pairs.
template <bool kLds>
__device__ __forceinline__ void q13_probe_tile(const TileRange &tr, int32_t tile, const int32_t (&k)[2][4], const uint32_t *s_key,
                                               const uint32_t *s_occ, const uint64_t *__restrict__ table, uint32_t cap,
                                               const int32_t *__restrict__ next, const KeyBitmap &bm, uint32_t *__restrict__ counts,
                                               uint8_t *__restrict__ flags8) {
    constexpr int kWaveRows = kProbeTile / kProbeWaves;  // 512
    constexpr int kIters = kWaveRows / 256;              // 2
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int64_t wbase = tr.tile_begin + (int64_t)wave * kWaveRows + lane * 4;
    // Bitmap test of the wave's 512 rows.  Bids arrive roughly in auction order, so the rows of a wave name a narrow
    // id range: when it fits 64 bitmap words, lane l fetches word (first + l) ONCE and every row reads its word from
    // the owning lane (ds_bpermute) -- eight texture-path loads per lane become one.  Wider ranges test row by row.
    uint32_t maybe = 0;
    if (bm.words) {
        uint32_t imin = ~0u, imax = 0;
#pragma unroll
        for (int it = 0; it < kIters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = wbase + it * 256 + j;
                const uint32_t idx = (uint32_t)k[it][j] - (uint32_t)bm.base;
                if (r >= tr.lo && r < tr.hi && idx < bm.n_bits) {
                    imin = min(imin, idx);
                    imax = max(imax, idx);
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            imin = min(imin, (uint32_t)__shfl_xor((int)imin, o, 64));
            imax = max(imax, (uint32_t)__shfl_xor((int)imax, o, 64));
        }
        if (imin <= imax) {  // (wave-uniform) some row of the wave lies inside the bitmap's range
            const uint32_t w0 = imin >> 5, w1 = imax >> 5;
            if (w1 - w0 < 64u) {
                const uint32_t word = bm.words[min(w0 + (uint32_t)lane, w1)];
#pragma unroll
                for (int it = 0; it < kIters; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t r = wbase + it * 256 + j;
                        const uint32_t idx = (uint32_t)k[it][j] - (uint32_t)bm.base;
                        const bool in = r >= tr.lo && r < tr.hi && idx < bm.n_bits;
                        const uint32_t wv = (uint32_t)__shfl((int)word, in ? (int)((idx >> 5) - w0) : 0, 64);
                        maybe |= (in ? (wv >> (idx & 31)) & 1u : 0u) << (it * 4 + j);
                    }
            } else {
#pragma unroll
                for (int it = 0; it < kIters; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t r = wbase + it * 256 + j;
                        maybe |= (uint32_t)(r >= tr.lo && r < tr.hi && q13_maybe(bm, k[it][j])) << (it * 4 + j);
                    }
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < kIters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = wbase + it * 256 + j;
                maybe |= (uint32_t)(r >= tr.lo && r < tr.hi) << (it * 4 + j);
            }
    }
    uint32_t mine = 0, bits = 0;
#pragma unroll
    for (int it = 0; it < kIters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((maybe >> (it * 4 + j)) & 1u) {
                const int32_t head = kLds ? find_lds(s_key, s_occ, table, cap, k[it][j]) : find_global(table, cap, k[it][j]);
                uint32_t n = 0;
                for (int32_t p = head; p >= 0; p = next[p]) ++n;
                mine += n;
                bits |= (n ? 1u : 0u) << (it * 4 + j);
            }
        }
    flags8[(size_t)tile * kProbeBlock + threadIdx.x] = (uint8_t)bits;
    const uint32_t incl = wave_incl_scan_u32(mine);
    if (lane == 63) counts[(size_t)tile * kProbeWaves + wave] = incl;
}

template <bool kLds>
__global__ __launch_bounds__(kProbeBlock) void q13_probe_count_kernel(const int32_t *__restrict__ auction, int64_t n_rows,
                                                                      SegTiles st, const uint64_t *__restrict__ table,
                                                                      uint32_t cap, const int32_t *__restrict__ next, KeyBitmap bm,
                                                                      uint32_t *__restrict__ counts, uint8_t *__restrict__ flags8) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    uint32_t *s_key = s_mem, *s_occ = s_mem + cap;
    if (kLds) {
        for (uint32_t i = threadIdx.x; i < (cap + 31) / 32; i += kProbeBlock) s_occ[i] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cap; i += kProbeBlock) {
            const uint64_t e = table[i];
            s_key[i] = (uint32_t)(e >> 32);
            if (e != kEmpty64) atomicOr(&s_occ[i >> 5], 1u << (i & 31));
        }
        __syncthreads();
    }
    const int wave = threadIdx.x >> 6, lane = lane_id();
    constexpr int kWaveRows = kProbeTile / kProbeWaves;  // 512
    constexpr int kIters = kWaveRows / 256;              // 2
    // One tile in hand, the next one requested.  (Two in hand + two in flight needed 81 VGPRs: one 16-wave workgroup per
    // CU instead of two, no faster.)
    int32_t tile = (int32_t)blockIdx.x;
    if (tile >= st.n_tiles) return;
    TileRange tr = locate_tile(st, tile, kProbeTile);
    int32_t k[kIters][4], kn[kIters][4];
    const int64_t lane_off = (int64_t)wave * kWaveRows + lane * 4;
#pragma unroll
    for (int it = 0; it < kIters; ++it) load4_i32(auction, tr.tile_begin + lane_off + it * 256, n_rows, k[it]);
#pragma unroll 1
    for (;;) {
        const int32_t nxt = tile + (int32_t)gridDim.x;
        TileRange trn = tr;
        if (nxt < st.n_tiles) {
            trn = locate_tile(st, nxt, kProbeTile);
#pragma unroll
            for (int it = 0; it < kIters; ++it) load4_i32(auction, trn.tile_begin + lane_off + it * 256, n_rows, kn[it]);
        }
        q13_probe_tile<kLds>(tr, tile, k, s_key, s_occ, table, cap, next, bm, counts, flags8);
        if (nxt >= st.n_tiles) break;
        tile = nxt;
        tr = trn;
#pragma unroll
        for (int it = 0; it < kIters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) k[it][j] = kn[it][j];
    }
}

// EMIT pass: reads the flag bytes (1 KiB per tile instead of 32 KiB of keys), re-probes only the flagged rows and writes
// their (bid row, side row) pairs in bid order (iteration-major, then lane, then j inside a wave -- the row order).
__global__ __launch_bounds__(kProbeBlock) void q13_probe_emit_kernel(const int32_t *__restrict__ auction, SegTiles st,
                                                                     const uint64_t *__restrict__ table, uint32_t cap,
                                                                     const int32_t *__restrict__ next,
                                                                     const uint32_t *__restrict__ counts,
                                                                     const uint8_t *__restrict__ flags8,
                                                                     const uint64_t *__restrict__ tile_base,
                                                                     int32_t *__restrict__ out_bid_row,
                                                                     int32_t *__restrict__ out_side_row) {
    const int wave = threadIdx.x >> 6, lane = lane_id();
    constexpr int kWaveRows = kProbeTile / kProbeWaves, kIters = kWaveRows / 256;
#pragma unroll 1
    for (int32_t tile = (int32_t)blockIdx.x; tile < st.n_tiles; tile += (int32_t)gridDim.x) {
        if (tile_base[tile + 1] == tile_base[tile]) continue;  // block-uniform: no pair in this tile
        const uint32_t bits = flags8[(size_t)tile * kProbeBlock + threadIdx.x];
        if (!__ballot(bits != 0)) continue;  // wave-uniform
        const TileRange tr = locate_tile(st, tile, kProbeTile);
        const int64_t wbase = tr.tile_begin + (int64_t)wave * kWaveRows + lane * 4;
        uint64_t pos = tile_base[tile];
        for (int w = 0; w < wave; ++w) pos += counts[(size_t)tile * kProbeWaves + w];
#pragma unroll
        for (int it = 0; it < kIters; ++it) {
            int32_t head[4];
            uint32_t mine = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                head[j] = -1;
                if (bits & (1u << (it * 4 + j))) {
                    head[j] = find_global(table, cap, auction[wbase + it * 256 + j]);
                    for (int32_t p = head[j]; p >= 0; p = next[p]) ++mine;
                }
            }
            const uint32_t incl = wave_incl_scan_u32(mine);
            uint64_t p = pos + (incl - mine);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                for (int32_t q = head[j]; q >= 0; q = next[q]) {
                    out_bid_row[p] = (int32_t)(wbase + it * 256 + j);
                    out_side_row[p] = q;
                    ++p;
                }
            pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
}

// ---- the bitmap path: standard flag tiles --------------------------------------------------------------------------
// With the membership bitmap almost every bid is rejected by one bit test, so the hash table is probed a few times per
// tile only and needs no LDS copy: the count pass becomes a plain streaming kernel in the flag-tile geometry (256 lanes,
// 32 rows per lane, tiles walked with the next descriptor requested early, as q2 / q7), the per-tile fixed costs (range
// reduction for the shared bitmap words, scan, stores) are paid per 32 rows of a lane instead of per 8.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) void q13_flag_kernel(const int32_t *__restrict__ auction, int64_t n_rows, SegTiles st,
                                                          const uint64_t *__restrict__ table, uint32_t cap,
                                                          const int32_t *__restrict__ next, KeyBitmap bm,
                                                          uint32_t *__restrict__ flag_words, uint32_t *__restrict__ counts) {
    int32_t tile = (int32_t)blockIdx.x;
    if (tile >= st.n_tiles) return;
    TileRange tr = locate_tile(st, tile, kFlagTile);
    const int32_t rel0 = flag_rel0();
    const int lane = lane_id();
#pragma unroll 1
    for (;;) {
        int32_t a[kFlagIters][4];
        load_flag_tile(auction, n_rows, tr, a);
        const int32_t nxt = tile + (int32_t)gridDim.x;
        TileRange trn = tr;
        if (nxt < st.n_tiles) trn = locate_tile(st, nxt, kFlagTile);
        const int32_t rel_lo = (int32_t)(tr.lo - tr.tile_begin), rel_hi = (int32_t)(tr.hi - tr.tile_begin);
        // (block-uniform) every row of the tile is a row of the segment: all but a segment's first and last tile.  Their rows need no position
        // test, and the range below comes from the keys' plain minimum and maximum -- a min3 / max3 per two rows.  (The pass ran at 60 % of the
        // HBM rate with ~20 vector instructions per row: a wave64 instruction holds its SIMD16 for four cycles, which at 4 bytes a row IS the
        // budget of a pass that wants to stay memory-bound.)
        const bool full = rel_lo <= 0 && rel_hi >= kFlagTile && tr.tile_begin >= 0 && tr.tile_begin + kFlagTile <= n_rows;
        // range of the wave's 2048 rows inside the bitmap
        uint32_t imin = ~0u, imax = 0;
        if (full) {
            int32_t mn = a[0][0], mx = a[0][0];
#pragma unroll
            for (int it = 0; it < kFlagIters; ++it) {
                mn = min(mn, min(min(a[it][0], a[it][1]), min(a[it][2], a[it][3])));
                mx = max(mx, max(max(a[it][0], a[it][1]), max(a[it][2], a[it][3])));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn = min(mn, __shfl_xor(mn, o, 64));
                mx = max(mx, __shfl_xor(mx, o, 64));
            }
            // the keys' range cut to the bitmap's: [base, base + n_bits)
            const int64_t lo = max((int64_t)mn, (int64_t)bm.base), hi = min((int64_t)mx, (int64_t)bm.base + (int64_t)bm.n_bits - 1);
            if (lo <= hi) {
                imin = (uint32_t)(lo - (int64_t)bm.base);
                imax = (uint32_t)(hi - (int64_t)bm.base);
            }
        } else {
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int32_t rel = rel0 + it * 256 + j;
                const uint32_t idx = (uint32_t)a[it][j] - (uint32_t)bm.base;
                if (rel >= rel_lo && rel < rel_hi && idx < bm.n_bits) {
                    imin = min(imin, idx);
                    imax = max(imax, idx);
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            imin = min(imin, (uint32_t)__shfl_xor((int)imin, o, 64));
            imax = max(imax, (uint32_t)__shfl_xor((int)imax, o, 64));
        }
        }
        uint32_t maybe = 0;
        if (imin <= imax && full) {  // (wave-uniform) the same two cases as below without the rows' position tests
            const uint32_t w0 = imin >> 5, w1 = imax >> 5;
            if (w1 - w0 < 64u) {
                const uint32_t word = bm.words[min(w0 + (uint32_t)lane, w1)];
                const uint32_t w0_4 = w0 << 2;
#pragma unroll
                for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t idx = (uint32_t)a[it][j] - (uint32_t)bm.base;
                        const bool in = idx - imin <= imax - imin;   // (inside the wave's words; outside them the lane index below would be another row's)
                        // lane (idx >> 5) - w0 holds the word: ds_bpermute takes the lane's byte address; the shift by idx uses its low five bits
                        const uint32_t wv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((idx >> 3) & ~3u) - w0_4), (int)word);
                        maybe |= (in ? (wv >> (idx & 31u)) & 1u : 0u) << (it * 4 + j);
                    }
            } else {
#pragma unroll
                for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) maybe |= (uint32_t)q13_maybe(bm, a[it][j]) << (it * 4 + j);
            }
        } else
        if (imin <= imax) {  // (wave-uniform)
            const uint32_t w0 = imin >> 5, w1 = imax >> 5;
            if (w1 - w0 < 64u) {
                const uint32_t word = bm.words[min(w0 + (uint32_t)lane, w1)];
#pragma unroll
                for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int32_t rel = rel0 + it * 256 + j;
                        const uint32_t idx = (uint32_t)a[it][j] - (uint32_t)bm.base;
                        const bool in = rel >= rel_lo && rel < rel_hi && idx < bm.n_bits;
                        const uint32_t wv = (uint32_t)__shfl((int)word, in ? (int)((idx >> 5) - w0) : 0, 64);
                        maybe |= (in ? (wv >> (idx & 31)) & 1u : 0u) << (it * 4 + j);
                    }
            } else {
#pragma unroll
                for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int32_t rel = rel0 + it * 256 + j;
                        maybe |= (uint32_t)(rel >= rel_lo && rel < rel_hi && q13_maybe(bm, a[it][j])) << (it * 4 + j);
                    }
            }
        }
        uint32_t flags = 0, mine = 0;
        if (__ballot(maybe != 0)) {  // (wave-uniform) some row of the wave may join
#pragma unroll
            for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((maybe >> (it * 4 + j)) & 1u) {
                        uint32_t n = 0;
                        for (int32_t p = find_global(table, cap, a[it][j]); p >= 0; p = next[p]) ++n;
                        mine += n;
                        flags |= (n ? 1u : 0u) << (it * 4 + j);
                    }
        }
        flag_words[(size_t)tile * kBlock + threadIdx.x] = flags;
        const uint32_t incl = wave_incl_scan_u32(mine);
        if (lane == 63) counts[(size_t)tile * kWavesPerBlock + (threadIdx.x >> 6)] = incl;
        if (nxt >= st.n_tiles) break;
        tile = nxt;
        tr = trn;
    }
}

// pairs of the flagged rows, in row order (wave, iteration, lane, j)
__global__ __launch_bounds__(kBlock) void q13_flag_emit_kernel(const int32_t *__restrict__ auction, SegTiles st,
                                                               const uint64_t *__restrict__ table, uint32_t cap,
                                                               const int32_t *__restrict__ next, const uint32_t *__restrict__ counts,
                                                               const uint32_t *__restrict__ flag_words,
                                                               const uint64_t *__restrict__ tile_base, int32_t *__restrict__ out_bid_row,
                                                               int32_t *__restrict__ out_side_row) {
    const int32_t tile = (int32_t)blockIdx.x;
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)tile * kWavesPerBlock);
    if (wc.x + wc.y + wc.z + wc.w == 0) return;
    const uint32_t flags = flag_words[(size_t)tile * kBlock + threadIdx.x];
    if (!__ballot(flags != 0)) return;  // wave-uniform
    const int wave = threadIdx.x >> 6;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const int64_t wbase = tr.tile_begin + flag_rel0();
    uint64_t pos = tile_base[tile] + (wave > 0 ? wc.x : 0u) + (wave > 1 ? wc.y : 0u) + (wave > 2 ? wc.z : 0u);
#pragma unroll 1
    for (int it = 0; it < kFlagIters; ++it) {
        const uint32_t f4 = (flags >> (it * 4)) & 15u;
        if (!__ballot(f4 != 0)) continue;
        int32_t head[4];
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            head[j] = -1;
            if (f4 & (1u << j)) {
                head[j] = find_global(table, cap, auction[wbase + it * 256 + j]);
                for (int32_t p = head[j]; p >= 0; p = next[p]) ++mine;
            }
        }
        const uint32_t incl = wave_incl_scan_u32(mine);
        uint64_t p = pos + (incl - mine);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (int32_t q = head[j]; q >= 0; q = next[q]) {
                out_bid_row[p] = (int32_t)(wbase + it * 256 + j);
                out_side_row[p] = q;
                ++p;
            }
        pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
}

// tile totals for the generic tile scan (which sums kWavesPerBlock = 4 counts per tile): 16 wave counts -> 4
__global__ __launch_bounds__(kBlock) void fold_counts_kernel(const uint32_t *__restrict__ c16, int32_t n_tiles,
                                                             uint32_t *__restrict__ c4) {
    const int32_t i = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (i < n_tiles * kWavesPerBlock) {
        const uint32_t *s = c16 + (size_t)(i / kWavesPerBlock) * kProbeWaves + (i % kWavesPerBlock) * (kProbeWaves / kWavesPerBlock);
        c4[i] = s[0] + s[1] + s[2] + s[3];
    }
}

}  // namespace

extern "C" {

int flockgpu_q13_side_join(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win, const int32_t *side_key,
                           const int32_t *side_value, int64_t side_rows, flockgpu_q13_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!bid || !out || bid->rows < 0 || side_rows < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "q13: null argument");
    FG_TRY(check_windows(ctx, win, bid->rows, "q13"));
    if (bid->rows > 0 && (!bid->auction || !bid->price || !bid->bidder || !bid->b_date_time))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q13: null bid column (the projection keeps all four)");
    if (side_rows > 0 && (!side_key || !side_value)) return fail(ctx, FLOCKGPU_ERR_INVALID, "q13: null side-input column");
    if (reinterpret_cast<uintptr_t>(bid->auction) & 15) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q13: auction column must be 16-byte aligned");
    if (bid->rows >= (int64_t(1) << 31) || side_rows >= (int64_t(1) << 30))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q13: relations are limited to 2^31 / 2^30 rows per call");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_win = win->n_windows;
    std::vector<int64_t> sb(n_win), se(n_win);
    for (int w = 0; w < n_win; ++w) {
        sb[w] = win->pane_row_offsets[win->win_pane_lo[w]];
        se[w] = win->pane_row_offsets[win->win_pane_hi[w]];
    }
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, "q13", sb.data(), se.data(), n_win, kProbeTile, &st));

    // ---- build (load factor <= 0.5)
    uint32_t cap = (uint32_t)std::max<int64_t>(64, side_rows * 2 + 1);
    uint64_t *table = nullptr;
    int32_t *next = nullptr;
    uint32_t *d_err = nullptr;
    FG_TRY(arena_get_t(ctx, "q13.table", (size_t)cap, &table));
    FG_TRY(arena_get_t(ctx, "q13.next", (size_t)side_rows + 1, &next));
    FG_TRY(arena_get_t(ctx, "q13.err", 4, &d_err));
    FG_HIP(ctx, hipMemsetAsync(table, 0xFF, sizeof(uint64_t) * cap, ctx->stream));
    FG_HIP(ctx, hipMemsetAsync(d_err, 0, sizeof(uint32_t), ctx->stream));
    if (side_rows > 0) {
        LaunchScope ls(ctx, "q13_build_kernel");
        hipLaunchKernelGGL(q13_build_kernel, dim3((unsigned)div_up(side_rows, kBlock)), dim3(kBlock), 0, ctx->stream, side_key,
                           (int32_t)side_rows, table, cap, next, d_err);
    }
    FG_TRY(check_launch(ctx, "q13_build_kernel"));

    // ---- membership bitmap over the side keys' range, when that range is affordable (<= 2^31 bits = 256 MiB)
    KeyBitmap bm{nullptr, 0, 0};
    if (side_rows > 0) {
        SegTiles st_side;
        const int64_t zero = 0;
        FG_TRY(build_seg_tiles(ctx, "q13.side", &zero, &side_rows, 1, kFlagTile, &st_side));
        int32_t *d_stats = nullptr, *h_stats = nullptr;
        FG_TRY(arena_get_t(ctx, "q13.stats", 4, &d_stats));
        FG_TRY(pinned_get_t(ctx, "q13.stats", 4, &h_stats));
        FG_TRY(segment_key_stats(ctx, side_key, side_rows, st_side, d_stats, d_stats + 1, d_stats + 2));
        FG_HIP(ctx, hipMemcpyAsync(h_stats, d_stats, sizeof(int32_t) * 3, hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int64_t range = (int64_t)h_stats[1] - h_stats[0] + 1;
        if (range <= (int64_t(1) << 31)) {
            uint32_t *words = nullptr;
            const size_t n_words = (size_t)div_up(range, 32) + 1;
            FG_TRY(arena_get_t(ctx, "q13.bitmap", n_words, &words));
            FG_HIP(ctx, hipMemsetAsync(words, 0, sizeof(uint32_t) * n_words, ctx->stream));
            {
                LaunchScope ls(ctx, "q13_bitmap_kernel");
                hipLaunchKernelGGL(q13_bitmap_kernel, dim3((unsigned)div_up(side_rows, kBlock)), dim3(kBlock), 0, ctx->stream,
                                   side_key, (int32_t)side_rows, h_stats[0], words);
            }
            FG_TRY(check_launch(ctx, "q13_bitmap_kernel"));
            bm = KeyBitmap{words, h_stats[0], (uint32_t)std::min<int64_t>(range, 0xFFFFFFFFll)};
        }
    }

    // ---- probe: count -> scan -> emit
    const bool lds = cap <= (uint32_t)kLdsSlots;
    const size_t lds_bytes = lds ? sizeof(uint32_t) * ((size_t)cap + (cap + 31) / 32) : 0;
    uint32_t *c16 = nullptr, *c4 = nullptr;
    uint64_t *tile_base = nullptr;
    int64_t *d_off = nullptr, *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, "q13.counts16", (size_t)st.n_tiles * kProbeWaves + 4, &c16));
    FG_TRY(arena_get_t(ctx, "q13.counts4", (size_t)st.n_tiles * kWavesPerBlock + 4, &c4));
    FG_TRY(arena_get_t(ctx, "q13.tile_base", (size_t)st.n_tiles + 1, &tile_base));
    FG_TRY(arena_get_t(ctx, "q13.seg_out_off", (size_t)n_win + 1, &d_off));
    FG_TRY(pinned_get_t(ctx, "q13.seg_out_off", (size_t)n_win + 2, &h_off));
    // two 16-wave workgroups per CU (each pays for its LDS copy once)
    const int per_cu = 2;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(st.n_tiles, (int64_t)ctx->num_cus * per_cu));
    uint8_t *flags8 = nullptr;
    FG_TRY(arena_get_t(ctx, "q13.flags8", (size_t)st.n_tiles * kProbeBlock + 16, &flags8));
    const bool flag_path = bm.words != nullptr;  // bitmap: plain flag tiles, no LDS table
    uint32_t *flag_words = nullptr;
    if (flag_path) FG_TRY(arena_get_t(ctx, "q13.flag_words", (size_t)st.n_tiles * kBlock + 4, &flag_words));
    if (flag_path && st.n_tiles > 0) {
        LaunchScope ls(ctx, "q13_flag_kernel");
        const unsigned g = (unsigned)std::min<int64_t>(st.n_tiles, (int64_t)ctx->num_cus * kStreamBlocksPerCu);
        hipLaunchKernelGGL(q13_flag_kernel, dim3(g), dim3(kBlock), 0, ctx->stream, bid->auction, bid->rows, st, table, cap, next, bm,
                           flag_words, c4);
    }
    FG_TRY(check_launch(ctx, "q13_flag_kernel"));
    if (!flag_path && lds && lds_bytes > 64 * 1024)
        FG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&q13_probe_count_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (!flag_path && st.n_tiles > 0) {
        LaunchScope ls(ctx, "q13_probe_count_kernel");
        if (lds)
            hipLaunchKernelGGL(q13_probe_count_kernel<true>, dim3(grid), dim3(kProbeBlock), lds_bytes, ctx->stream, bid->auction,
                               bid->rows, st, table, cap, next, bm, c16, flags8);
        else
            hipLaunchKernelGGL(q13_probe_count_kernel<false>, dim3(grid), dim3(kProbeBlock), 0, ctx->stream, bid->auction, bid->rows,
                               st, table, cap, next, bm, c16, flags8);
    }
    FG_TRY(check_launch(ctx, "q13_probe_count_kernel"));
    if (!flag_path && st.n_tiles > 0) {
        hipLaunchKernelGGL(fold_counts_kernel, dim3((unsigned)div_up((int64_t)st.n_tiles * kWavesPerBlock, kBlock)), dim3(kBlock), 0,
                           ctx->stream, c16, st.n_tiles, c4);
        FG_TRY(check_launch(ctx, "fold_counts_kernel"));
    }
    FG_TRY(launch_tile_scan(ctx, c4, st.n_tiles, tile_base, st.tile_first, st.n_seg, d_off));
    FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * ((size_t)n_win + 1), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipMemcpyAsync(h_off + n_win + 1, d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (*reinterpret_cast<uint32_t *>(h_off + n_win + 1)) return fail(ctx, FLOCKGPU_ERR_CAPACITY, "q13: side-input table overflow");
    std::vector<int64_t> &offs = ctx->host_i64["q13.win_out_offsets"];
    offs.assign(h_off, h_off + n_win + 1);
    const int64_t n_out = offs[n_win];
    if (n_out >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q13: join output exceeds 2^31 rows");
    int32_t *o_br = nullptr, *o_sr = nullptr, *o_a = nullptr, *o_b = nullptr, *o_p = nullptr, *o_v = nullptr;
    int64_t *o_t = nullptr;
    FG_TRY(arena_get_t(ctx, "q13.out_bid_row", (size_t)n_out + 1, &o_br));
    FG_TRY(arena_get_t(ctx, "q13.out_side_row", (size_t)n_out + 1, &o_sr));
    FG_TRY(arena_get_t(ctx, "q13.out_auction", (size_t)n_out + 1, &o_a));
    FG_TRY(arena_get_t(ctx, "q13.out_bidder", (size_t)n_out + 1, &o_b));
    FG_TRY(arena_get_t(ctx, "q13.out_price", (size_t)n_out + 1, &o_p));
    FG_TRY(arena_get_t(ctx, "q13.out_time", (size_t)n_out + 1, &o_t));
    FG_TRY(arena_get_t(ctx, "q13.out_value", (size_t)n_out + 1, &o_v));
    if (st.n_tiles > 0 && n_out > 0) {
        if (flag_path) {
            LaunchScope ls(ctx, "q13_flag_emit_kernel");
            hipLaunchKernelGGL(q13_flag_emit_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, bid->auction, st, table, cap,
                               next, c4, flag_words, tile_base, o_br, o_sr);
        } else {
            LaunchScope ls(ctx, "q13_probe_emit_kernel");
            hipLaunchKernelGGL(q13_probe_emit_kernel, dim3(grid), dim3(kProbeBlock), 0, ctx->stream, bid->auction, st, table, cap,
                               next, c16, flags8, tile_base, o_br, o_sr);
        }
        FG_TRY(check_launch(ctx, "q13 emit"));
        FG_TRY(gather_i32(ctx, bid->auction, o_br, n_out, o_a));
        FG_TRY(gather_i32(ctx, bid->bidder, o_br, n_out, o_b));
        FG_TRY(gather_i32(ctx, bid->price, o_br, n_out, o_p));
        FG_TRY(gather_i64(ctx, bid->b_date_time, o_br, n_out, o_t));
        FG_TRY(gather_i32(ctx, side_value, o_sr, n_out, o_v));
    }
    out->auction = o_a;
    out->bidder = o_b;
    out->price = o_p;
    out->b_date_time = o_t;
    out->value = o_v;
    out->bid_row = o_br;
    out->side_row = o_sr;
    out->win_out_offsets = offs.data();
    out->rows = n_out;
    return FLOCKGPU_OK;
}

}  // extern "C"
