// Key-partitioned exchange step of the distributed plans, device side:
//   RepartitionExec: Hash([seller] | [p_id] | [auction], n)   (flock/src/distributed_plan/planner.rs:152-171,
//   playground/src/distributed_plan/shuffle_writer.rs:106-148, q5.dag / q8.dag)
// The reference computes  create_hashes(ahash seeds 0,0,0,0) % n  per row and `take`s the rows of every
// destination; which destination a key lands on is unobservable in the query result,
// so a fixed integer mix is used here instead of ahash.
//
// flockgpu_partition_by_key: rows of every window -> row numbers grouped by (destination, window), input order
// kept inside a group; count -> scan -> emit over (destination, tile) pairs (scan.hpp), so the send buffers of
// the all-to-all are contiguous per destination and per window inside a destination.
// flockgpu_take_*: the `take` that builds the send buffers / regroups the received rows.
#include <algorithm>

#include "gather.hpp"

using namespace flockgpu;

namespace {

constexpr int kMaxParts = 64;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // murmur3 finaliser
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t part_of(int32_t key, uint32_t n_parts) {
#if defined(FLOCKGPU_EXPERIMENTAL) && defined(FLOCKGPU_AB_PART_NOHASH)   // (A/B builds only: what the three integer multiplies of the hash cost the count pass)
    return (uint32_t)key & (n_parts - 1);
#else
    return (uint32_t)(((uint64_t)mix32((uint32_t)key) * n_parts) >> 32);
#endif
}

// Destination of each of the lane's 32 rows, one byte each (0xFF = row outside the window), four per word.
__device__ __forceinline__ void tile_parts(const int32_t *__restrict__ keys, int64_t n_rows, const TileRange &tr,
                                           uint32_t n_parts, uint32_t (&d)[kFlagIters]) {
    int32_t a[kFlagIters][4];
    load_flag_tile(keys, n_rows, tr, a);   // (read once: the emit pass takes the destination bytes the count pass leaves, not the keys)
    const int32_t rel_lo = (int32_t)(tr.lo - tr.tile_begin), rel_hi = (int32_t)(tr.hi - tr.tile_begin);
    const int32_t rel0 = flag_rel0();
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) {
        d[it] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t rel = rel0 + it * 256 + j;
            const uint32_t p = (rel >= rel_lo && rel < rel_hi) ? part_of(a[it][j], n_parts) : 0xFFu;
            d[it] |= p << (8 * j);
        }
    }
}

// ---- every destination's ranks from ONE pass over the lane's rows (round 6; rounds 1-5 looped over the destinations inside every tile: flags by
// 32 compares, a list build of eight DPP scans and two barriers PER destination -- made for <= 8 ranks and paid n times).
// A lane counts its 32 rows of a GROUP of eight destinations in one 64-bit word of 8-bit fields (<= 32 each), spread into four words of 16-bit
// fields (a wave's 2048 rows fit) -- so four DPP scans give every lane its offset inside every destination of the group at once.
constexpr uint32_t kPartGroup = 8;
constexpr int kPartLoopMax = 4;   // up to here the emit pass loops over the destinations (partition_emit_loop_kernel)

__device__ __forceinline__ uint64_t group_counts(const uint32_t (&d)[kFlagIters], uint32_t g8) {
    uint64_t acc = 0;
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t pg = ((d[it] >> (8 * j)) & 0xFFu) - g8;   // (0xFF = outside the window: in no group)
            acc += pg < kPartGroup ? 1ull << (pg * 8) : 0ull;
        }
    return acc;
}

__device__ __forceinline__ void spread16(uint64_t acc, uint32_t (&w)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t t = (uint32_t)(acc >> (16 * k)) & 0xFFFFu;
        w[k] = (t & 0xFFu) | ((t & 0xFF00u) << 8);
    }
}

// counts[((part * n_tiles) + tile) * 4 + wave]
// dest[tile * 8192 + row of the tile]: the row's destination (0xFF outside the window), for the emit pass -- one byte per row written here
// instead of four bytes of key read and a hash computed a second time there.
__global__ __launch_bounds__(kBlock) void partition_count_kernel(const int32_t *__restrict__ keys, int64_t n_rows, SegTiles st,
                                                                 uint32_t n_parts, uint32_t *__restrict__ counts, uint32_t *__restrict__ dest) {
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    uint32_t d[kFlagIters];
    tile_parts(keys, n_rows, tr, n_parts, d);
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) __builtin_nontemporal_store(d[it], dest + (size_t)tile * (kFlagTile / 4) + (flag_rel0() >> 2) + it * 64);
    const int wave = threadIdx.x >> 6;
#pragma unroll 1
    for (uint32_t g8 = 0; g8 < n_parts; g8 += kPartGroup) {
        uint32_t w[4];
        spread16(group_counts(d, g8), w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t incl = wave_incl_scan_u32(w[k]);
            if (lane_id() == 63) {
                const uint32_t p0 = g8 + 2 * k;
                if (p0 < n_parts) counts[((size_t)p0 * st.n_tiles + tile) * kWavesPerBlock + wave] = incl & 0xFFFFu;
                if (p0 + 1 < n_parts) counts[((size_t)(p0 + 1) * st.n_tiles + tile) * kWavesPerBlock + wave] = incl >> 16;
            }
        }
    }
}

// The tile's rows grouped by destination in LDS (row order kept inside a destination), then written out run by run: positions inside
// the list = start of (destination, wave) + the lane's offset (the four scans) + a running count -- the last two live in ONE LDS counter per
// (destination of the group, thread), so a row costs one returning LDS add and one 2-byte LDS store.  For that a lane takes 32 CONSECUTIVE
// rows (two 16-byte reads of the count pass's destination bytes); the bytes are kept in LDS, where the output loop finds an entry's destination.
__global__ __launch_bounds__(kBlock) void partition_emit_kernel(const uint32_t *__restrict__ dest, SegTiles st, uint32_t n_parts,
                                                                const uint32_t *__restrict__ counts, const uint64_t *__restrict__ tile_base,
                                                                int32_t *__restrict__ out_rows, PartPayload pl) {
    __shared__ uint16_t s_list[kFlagTile];
    __shared__ __attribute__((aligned(16))) uint32_t s_dest[kFlagTile / 4];
    __shared__ uint32_t s_cnt[kPartGroup * kBlock];
    __shared__ uint32_t s_start[kMaxParts * kWavesPerBlock];   // list position of (destination, wave)'s first row
    __shared__ uint32_t s_gdelta[kMaxParts];                   // output position - list position, per destination
    __shared__ uint32_t s_total;
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const int wave = threadIdx.x >> 6, lane = lane_id();
    uint32_t d[kFlagIters];   // the lane's rows: wave * 2048 + lane * 32 + (0 .. 31)
    {
        const uint32_t *src = dest + (size_t)tile * (kFlagTile / 4) + threadIdx.x * 8;
        const uint4 a = stream_load4u(src), b = stream_load4u(src + 4);
        d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        *reinterpret_cast<uint4 *>(s_dest + threadIdx.x * 8) = a;
        *reinterpret_cast<uint4 *>(s_dest + threadIdx.x * 8 + 4) = b;
    }
    if (wave == 0) {   // (n_parts <= 64: one lane per destination)
        const bool live = (uint32_t)lane < n_parts;
        const size_t slot = (size_t)(live ? lane : 0) * st.n_tiles + tile;
        uint4 wc = *reinterpret_cast<const uint4 *>(counts + slot * kWavesPerBlock);
        if (!live) wc = make_uint4(0, 0, 0, 0);
        const uint32_t total = wc.x + wc.y + wc.z + wc.w;
        const uint32_t incl = wave_incl_scan_u32(total);
        const uint32_t off = incl - total;
        if (live) {
            s_start[lane * kWavesPerBlock + 0] = off;
            s_start[lane * kWavesPerBlock + 1] = off + wc.x;
            s_start[lane * kWavesPerBlock + 2] = off + wc.x + wc.y;
            s_start[lane * kWavesPerBlock + 3] = off + wc.x + wc.y + wc.z;
            s_gdelta[lane] = (uint32_t)tile_base[slot] - off;   // (positions stay below 2^31: partition_by_key_async checks the row count)
            if ((uint32_t)lane == n_parts - 1) s_total = incl;
        }
    }
    __syncthreads();
    const uint32_t rel_first = (uint32_t)threadIdx.x * 32;
#pragma unroll 1
    for (uint32_t g8 = 0; g8 < n_parts; g8 += kPartGroup) {
        uint32_t w[4];
        spread16(group_counts(d, g8), w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t excl = wave_incl_scan_u32(w[k]) - w[k];
            const uint32_t p0 = g8 + 2 * k;
            if (p0 < n_parts) s_cnt[(2 * k) * kBlock + threadIdx.x] = s_start[p0 * kWavesPerBlock + wave] + (excl & 0xFFFFu);
            if (p0 + 1 < n_parts) s_cnt[(2 * k + 1) * kBlock + threadIdx.x] = s_start[(p0 + 1) * kWavesPerBlock + wave] + (excl >> 16);
        }
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t pg = ((d[it] >> (8 * j)) & 0xFFu) - g8;
                if (pg < kPartGroup) {
                    const uint32_t pos = atomicAdd(&s_cnt[pg * kBlock + threadIdx.x], 1u);   // (the thread's own counter: ds_add_rtn_u32, bank = lane)
                    s_list[pos] = (uint16_t)(rel_first + it * 4 + j);
                }
            }
    }
    __syncthreads();
    const uint32_t total = s_total;
    const uint8_t *s_dest_b = reinterpret_cast<const uint8_t *>(s_dest);
    for (uint32_t i = threadIdx.x; i < total; i += kBlock) {
        const uint32_t rel = s_list[i];
        const uint32_t pos = i + s_gdelta[s_dest_b[rel]];
        const int64_t r = tr.tile_begin + rel;
        if (!pl.skip_rows) out_rows[pos] = (int32_t)r;
        for (int c = 0; c < pl.n; ++c) pl.dst[c][pos] = pl.src[c][r];
    }
}

// Up to four destinations: the write-out destination by destination (the form of rounds 1-5, now on the count pass's destination bytes: no key
// read, no hash).  With two payload columns, emit pass per 8e7 rows, loop / one pass: 0.30 / 0.40 ms at two destinations, 0.34 / 0.37 at four,
// (0.47) / 0.41 at eight (`profiles/r06/partition_ab.txt`) -- a workgroup of the loop reaches its first write-out after one list build, the one
// pass after its scans and the whole scatter phase; the loop's cost grows with the destinations,
// the one pass's does not.  (Asking for the payload columns' tile as a coalesced stream first, so that the scattered reads of the write-out find
// their lines in the L2: 0.40 -> 0.35 for the one pass at two destinations, nothing at eight, nothing for the loop -- not kept.)
__device__ __forceinline__ uint32_t flags_of(const uint32_t (&d)[kFlagIters], uint32_t part) {
    uint32_t flags = 0;
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) flags |= (((d[it] >> (8 * j)) & 0xFFu) == part ? 1u : 0u) << (it * 4 + j);
    return flags;
}
__global__ __launch_bounds__(kBlock) void partition_emit_loop_kernel(const uint32_t *__restrict__ dest, SegTiles st, uint32_t n_parts,
                                                                     const uint32_t *__restrict__ counts, const uint64_t *__restrict__ tile_base,
                                                                     int32_t *__restrict__ out_rows, PartPayload pl) {
    __shared__ uint16_t s_list[kFlagTile];
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    uint32_t d[kFlagIters];   // (the count pass's geometry: byte j of d[it] = row flag_rel0() + it * 256 + j)
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) d[it] = __builtin_nontemporal_load(dest + (size_t)tile * (kFlagTile / 4) + (flag_rel0() >> 2) + it * 64);
#pragma unroll 1
    for (uint32_t part = 0; part < n_parts; ++part) {
        const size_t slot = (size_t)part * st.n_tiles + tile;
        const uint4 wc = *reinterpret_cast<const uint4 *>(counts + slot * kWavesPerBlock);
        if (wc.x + wc.y + wc.z + wc.w == 0) continue;  // block-uniform
        const uint32_t total = build_flag_list(flags_of(d, part), wc, s_list);
        __syncthreads();
        const uint64_t base = tile_base[slot];
        for (uint32_t i = threadIdx.x; i < total; i += kBlock) {
            const int64_t r = tr.tile_begin + s_list[i];
            if (!pl.skip_rows) out_rows[base + i] = (int32_t)r;
            for (int c = 0; c < pl.n; ++c) pl.dst[c][base + i] = pl.src[c][r];
        }
        __syncthreads();  // s_list is rewritten for the next destination
    }
}

// Hash([key], 1): every row of a window goes to the one destination, in order -- no key is read.
__global__ __launch_bounds__(kBlock) void partition_count_one_kernel(SegTiles st, uint32_t *__restrict__ counts) {
    const int32_t t = (int32_t)(blockIdx.x * kBlock + threadIdx.x);   // one thread per (tile, wave)
    if (t >= st.n_tiles * kWavesPerBlock) return;
    const TileRange tr = st.tiles[t / kWavesPerBlock];
    const int64_t lo = tr.tile_begin + (int64_t)(t % kWavesPerBlock) * kFlagWaveRows, hi = lo + kFlagWaveRows;
    const int64_t a = lo > tr.lo ? lo : tr.lo, b = hi < tr.hi ? hi : tr.hi;
    counts[t] = b > a ? (uint32_t)(b - a) : 0u;
}

__global__ __launch_bounds__(kBlock) void partition_emit_one_kernel(SegTiles st, const uint64_t *__restrict__ tile_base, int32_t *__restrict__ out_rows,
                                                                    PartPayload pl) {
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const uint64_t base = tile_base[tile];
    const int32_t n = (int32_t)(tr.hi - tr.lo);
    // four consecutive rows per lane and step: 16-byte loads and stores at dword-aligned addresses (all the hardware asks of a global access;
    // source and destination are offset against each other by whatever the windows in front left)
    for (int32_t i0 = 0; i0 < n; i0 += kBlock * 8) {
        int4 v[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int32_t i = i0 + u * kBlock * 4 + (int32_t)threadIdx.x * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c >= pl.n) break;
                if (i + 4 <= n) {
                    __builtin_memcpy(&v[u][c], pl.src[c] + tr.lo + i, 16);
                } else {
                    int32_t t[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = i + j < n ? pl.src[c][tr.lo + i + j] : 0;
                    v[u][c] = make_int4(t[0], t[1], t[2], t[3]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int32_t i = i0 + u * kBlock * 4 + (int32_t)threadIdx.x * 4;
            if (i >= n) continue;
            if (i + 4 <= n) {
                if (!pl.skip_rows) {
                    const int32_t r0 = (int32_t)(tr.lo + i);
                    const int4 r = make_int4(r0, r0 + 1, r0 + 2, r0 + 3);
                    __builtin_memcpy(out_rows + base + i, &r, 16);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < pl.n) __builtin_memcpy(pl.dst[c] + base + i, &v[u][c], 16);
            } else {
                for (int j = 0; i + j < n; ++j) {
                    if (!pl.skip_rows) out_rows[base + i + j] = (int32_t)(tr.lo + i + j);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < pl.n) pl.dst[c][base + i + j] = j == 0 ? v[u][c].x : j == 1 ? v[u][c].y : j == 2 ? v[u][c].z : v[u][c].w;
                }
            }
        }
    }
}

// ---- Partial DISTINCT per tile.  What repeats inside a tile of sellers is the hot key (3/4 of a window's auctions name one of a few
// sellers; the others are ~2000 different ids of millions): each wave follows the key most of its lanes hold -- kept while it covers
// >= 16 lanes, re-elected from two lanes otherwise, as q8_sellers_bitmap_kernel does -- and hands it on ONCE; every other row is handed
// on as it is.  No LDS, no atomics: a version with an 8192-slot LDS set (exact per tile) took 0.18 ms for 6e7 keys, i.e. more than the
// shuffle of the rows it saved.  Duplicates that survive are the FinalPartitioned DISTINCT's job.
__global__ __launch_bounds__(kBlock) void tile_distinct_flag_kernel(const int32_t *__restrict__ keys, int64_t n_rows, SegTiles st,
                                                                    uint32_t *__restrict__ flag_words, uint32_t *__restrict__ counts) {
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    int32_t a[kFlagIters][4];
    load_flag_tile(keys, n_rows, tr, a);
    const int32_t rel_lo = (int32_t)(tr.lo - tr.tile_begin), rel_hi = (int32_t)(tr.hi - tr.tile_begin);
    const int32_t rel0 = flag_rel0();
    const int lane = lane_id();
    uint32_t flags = 0;
    int32_t hot = 0;
    bool have_hot = false, hot_sent = false;  // wave-uniform
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t rel = rel0 + it * 256 + j;
            const int32_t key = a[it][j];
            const bool live = rel >= rel_lo && rel < rel_hi;
            uint64_t m = have_hot ? __ballot(live && key == hot) : 0;
            if (__popcll((unsigned long long)m) < 16) {
                const uint64_t lv = __ballot(live);
                const bool prev_have = have_hot;
                const int32_t prev_hot = hot;
                have_hot = false;
                m = 0;
                if (lv) {
                    const int32_t c1 = __builtin_amdgcn_readlane(key, __ffsll((unsigned long long)lv) - 1);
                    uint64_t m1 = __ballot(live && key == c1);
                    int32_t c = c1;
                    const uint64_t rest = lv & ~m1;
                    if (__popcll((unsigned long long)m1) < 16 && rest) {
                        const int32_t c2 = __builtin_amdgcn_readlane(key, __ffsll((unsigned long long)rest) - 1);
                        const uint64_t m2 = __ballot(live && key == c2);
                        if (__popcll((unsigned long long)m2) > __popcll((unsigned long long)m1)) {
                            m1 = m2;
                            c = c2;
                        }
                    }
                    if (__popcll((unsigned long long)m1) >= 2) {   // worth following
                        hot_sent = hot_sent && prev_have && c == prev_hot;   // the same key re-elected: it was handed on already
                        hot = c;
                        have_hot = true;
                        m = m1;
                    }
                }
            }
            bool first = live;
            if (m) {
                const bool in_m = (m >> lane) & 1;
                if (in_m) first = !hot_sent && lane == __ffsll((unsigned long long)m) - 1;
                hot_sent = true;
            }
            flags |= (first ? 1u : 0u) << (it * 4 + j);
        }
    store_flags_and_counts(flags, tile, flag_words, counts);
}

}  // namespace

namespace flockgpu {

// The partition pass without its host wait: the row numbers grouped by (destination, window) and the n_parts * n_win + 1
// group offsets -- on the device (*d_group_off) and queued for copy into pinned memory (*h_group_off, valid after the next
// synchronisation of the ctx stream).  The number of rows written is known up front: every row of every window.
int partition_by_key_async(flockgpu_ctx *ctx, const int32_t *keys, int64_t rows, const flockgpu_windows *win, int32_t n_parts,
                           const int32_t **d_rows, const int64_t **d_group_off, const int64_t **h_group_off, int64_t *n_out,
                           const PartPayload *payload, const char *cache_name) {
    // `cache_name` keys the schedule-dependent state (tile descriptors, first-tile table): a caller that partitions two relations
    // per call (the q3 / q8 exchanges) gives each its own, so that neither evicts the other's cached upload (a re-upload waits for the stream)
    const std::string nm = cache_name ? cache_name : "partition";
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (payload && (payload->n < 0 || payload->n > 4)) return fail(ctx, FLOCKGPU_ERR_INVALID, "partition: at most four payload columns");
    if (rows < 0 || (rows > 0 && !keys)) return fail(ctx, FLOCKGPU_ERR_INVALID, "partition: null argument");
    if (n_parts < 1 || n_parts > kMaxParts)
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "partition: n_parts must be in [1, %d]", kMaxParts);
    FG_TRY(check_windows(ctx, win, rows, "partition"));
    if (reinterpret_cast<uintptr_t>(keys) & 15) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "partition: key column must be 16-byte aligned");
    if (rows >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "partition: relations are limited to 2^31 rows per call");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_win = win->n_windows;
    std::vector<int64_t> sb(n_win), se(n_win);
    int64_t covered = 0;
    for (int w = 0; w < n_win; ++w) {
        sb[w] = win->pane_row_offsets[win->win_pane_lo[w]];
        se[w] = win->pane_row_offsets[win->win_pane_hi[w]];
        covered += se[w] - sb[w];
    }
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, nm.c_str(), sb.data(), se.data(), n_win, kFlagTile, &st));
    const int64_t slots = (int64_t)st.n_tiles * n_parts;
    if (slots > 0x7fffffff / 2) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "partition: too many (destination, tile) pairs");
    const size_t n_groups = (size_t)n_parts * n_win;

    // first pseudo-tile of every (destination, window) group: destination-major.  Rebuilt only when the schedule changes
    // (the pinned staging of the previous upload may still be in flight otherwise).
    int32_t *d_first = nullptr, *h_first = nullptr;
    FG_TRY(arena_get_t(ctx, (nm + ".first").c_str(), n_groups + 1, &d_first));
    FG_TRY(pinned_get_t(ctx, (nm + ".first").c_str(), n_groups + 1, &h_first));
    std::vector<int64_t> &first_key = ctx->host_i64[nm + ".first_key"];
    std::vector<int64_t> key_now;
    key_now.reserve((size_t)2 * n_win + 3);
    key_now.push_back(n_parts);
    key_now.push_back((int64_t)reinterpret_cast<uintptr_t>(d_first));
    key_now.push_back((int64_t)reinterpret_cast<uintptr_t>(st.tiles));
    for (int w = 0; w < n_win; ++w) {
        key_now.push_back(sb[w]);
        key_now.push_back(se[w]);
    }
    if (first_key != key_now) {
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int64_t tiles = 0;
        std::vector<int32_t> tf(n_win + 1);
        for (int w = 0; w < n_win; ++w) {
            tf[w] = (int32_t)tiles;
            if (se[w] > sb[w]) tiles += div_up(se[w] - (sb[w] & ~int64_t(3)), kFlagTile);
        }
        for (int p = 0; p < n_parts; ++p)
            for (int w = 0; w < n_win; ++w) h_first[(size_t)p * n_win + w] = (int32_t)((int64_t)p * st.n_tiles + tf[w]);
        h_first[n_groups] = (int32_t)slots;
        FG_HIP(ctx, hipMemcpyAsync(d_first, h_first, sizeof(int32_t) * (n_groups + 1), hipMemcpyHostToDevice, ctx->stream));
        first_key = key_now;
    }

    uint32_t *counts = nullptr;
    uint64_t *tile_base = nullptr;
    int64_t *d_off = nullptr, *h_off = nullptr;
    int32_t *o_rows = nullptr;
    FG_TRY(arena_get_t(ctx, "partition.counts", (size_t)slots * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, "partition.tile_base", (size_t)slots + 1, &tile_base));
    FG_TRY(arena_get_t(ctx, "partition.group_off", n_groups + 1, &d_off));
    FG_TRY(pinned_get_t(ctx, "partition.group_off", n_groups + 1, &h_off));
    FG_TRY(arena_get_t(ctx, "partition.rows", (size_t)covered + 1, &o_rows));
    uint32_t *dest = nullptr;   // the rows' destination bytes, tile by tile (count pass -> emit pass)
    if (n_parts > 1) FG_TRY(arena_get_t(ctx, "partition.dest", (size_t)st.n_tiles * (kFlagTile / 4) + 4, &dest));
    const PartPayload pay = payload ? *payload : PartPayload{};
    if (st.n_tiles > 0) {
        LaunchScope ls(ctx, "partition_count_kernel");
        if (n_parts == 1)
            hipLaunchKernelGGL(partition_count_one_kernel, dim3((unsigned)div_up((int64_t)st.n_tiles * kWavesPerBlock, kBlock)), dim3(kBlock), 0, ctx->stream, st, counts);
        else
            hipLaunchKernelGGL(partition_count_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, keys, rows, st, (uint32_t)n_parts, counts, dest);
    }
    FG_TRY(check_launch(ctx, "partition_count_kernel"));
    FG_TRY(launch_tile_scan(ctx, counts, (int32_t)slots, tile_base, d_first, (int32_t)n_groups, d_off));
    if (st.n_tiles > 0) {
        LaunchScope ls(ctx, "partition_emit_kernel");
        if (n_parts == 1)
            hipLaunchKernelGGL(partition_emit_one_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, st, tile_base, o_rows, pay);
        else if (n_parts <= kPartLoopMax)
            hipLaunchKernelGGL(partition_emit_loop_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, dest, st, (uint32_t)n_parts, counts, tile_base, o_rows, pay);
        else
            hipLaunchKernelGGL(partition_emit_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, dest, st, (uint32_t)n_parts, counts, tile_base, o_rows, pay);
    }
    FG_TRY(check_launch(ctx, "partition_emit_kernel"));
    FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * (n_groups + 1), hipMemcpyDeviceToHost, ctx->stream));
    *d_rows = o_rows;
    *d_group_off = d_off;
    *h_group_off = h_off;
    *n_out = covered;
    return FLOCKGPU_OK;
}

int tile_distinct_i32(flockgpu_ctx *ctx, const char *name, const int32_t *keys, int64_t rows, const flockgpu_windows *win,
                      const int32_t **out_keys, std::vector<int64_t> *out_win_off, int64_t *n_out) {
    const std::string nm = name;
    *out_keys = nullptr;
    *n_out = 0;
    FG_TRY(check_windows(ctx, win, rows, "partial distinct"));
    if (rows > 0 && !keys) return fail(ctx, FLOCKGPU_ERR_INVALID, "partial distinct: null key column");
    if (reinterpret_cast<uintptr_t>(keys) & 15) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "partial distinct: key column must be 16-byte aligned");
    if (rows >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "partial distinct: relations are limited to 2^31 rows per call");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_win = win->n_windows;
    std::vector<int64_t> sb((size_t)std::max(n_win, 1)), se(sb.size());
    int64_t covered = 0;
    for (int w = 0; w < n_win; ++w) {
        sb[(size_t)w] = win->pane_row_offsets[win->win_pane_lo[w]];
        se[(size_t)w] = win->pane_row_offsets[win->win_pane_hi[w]];
        covered += se[(size_t)w] - sb[(size_t)w];
    }
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, nm.c_str(), sb.data(), se.data(), n_win, kFlagTile, &st));
    uint32_t *flag_words = nullptr, *counts = nullptr;
    uint64_t *tile_base = nullptr;
    int64_t *d_off = nullptr, *h_off = nullptr;
    int32_t *o_rows = nullptr, *o_keys = nullptr;
    FG_TRY(arena_get_t(ctx, (nm + ".flags").c_str(), (size_t)st.n_tiles * kBlock + 4, &flag_words));
    FG_TRY(arena_get_t(ctx, (nm + ".counts").c_str(), (size_t)st.n_tiles * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, (nm + ".tile_base").c_str(), (size_t)st.n_tiles + 2, &tile_base));
    FG_TRY(arena_get_t(ctx, (nm + ".off").c_str(), (size_t)n_win + 2, &d_off));
    FG_TRY(pinned_get_t(ctx, (nm + ".off").c_str(), (size_t)n_win + 2, &h_off));
    FG_TRY(arena_get_t(ctx, (nm + ".rows").c_str(), (size_t)covered + 4, &o_rows));
    FG_TRY(arena_get_t(ctx, (nm + ".keys").c_str(), (size_t)covered + 4, &o_keys));
    if (st.n_tiles > 0) {
        LaunchScope ls(ctx, "tile_distinct_flag_kernel");
        hipLaunchKernelGGL(tile_distinct_flag_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, keys, rows, st, flag_words, counts);
    }
    FG_TRY(check_launch(ctx, "tile_distinct_flag_kernel"));
    FG_TRY(launch_tile_scan(ctx, counts, st.n_tiles, tile_base, st.tile_first, st.n_seg, d_off));
    FG_TRY(emit_flagged_rows(ctx, st, flag_words, counts, tile_base, o_rows));
    FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * ((size_t)n_win + 1), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out_win_off->assign(h_off, h_off + n_win + 1);
    if (n_win == 0) out_win_off->assign(1, 0);
    *n_out = (*out_win_off)[(size_t)n_win];
    FG_TRY(gather_i32(ctx, keys, o_rows, *n_out, o_keys));
    *out_keys = o_keys;
    return FLOCKGPU_OK;
}

}  // namespace flockgpu

extern "C" {

int flockgpu_partition_by_key(flockgpu_ctx *ctx, const int32_t *keys, int64_t rows, const flockgpu_windows *win,
                              int32_t n_parts, flockgpu_partition_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!out || !win) return fail(ctx, FLOCKGPU_ERR_INVALID, "partition: null argument");
    const int32_t *d_rows = nullptr;
    const int64_t *d_off = nullptr, *h_off = nullptr;
    int64_t n_out = 0;
    const PartPayload *pay_p = nullptr;
#if defined(FLOCKGPU_EXPERIMENTAL)   // (A/B builds only, tools/gpu_partition_ab.py --payload k: k payload columns ride in the emit pass as in comm.hip's exchange -- the key column k times)
    PartPayload pay;
    if (const char *e = exp_env("FLOCKGPU_AB_PART_PAYLOAD")) {
        for (int c = 0; c < atoi(e) && c < 4; ++c) {
            void *p = nullptr;
            FG_TRY(arena_get(ctx, ("partition.ab_payload" + std::to_string(c)).c_str(), (size_t)rows * 4 + 16, &p));
            pay.src[pay.n] = keys;
            pay.dst[pay.n++] = static_cast<int32_t *>(p);
        }
        pay.skip_rows = true;
        pay_p = &pay;
    }
#endif
    FG_TRY(partition_by_key_async(ctx, keys, rows, win, n_parts, &d_rows, &d_off, &h_off, &n_out, pay_p));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t n_groups = (size_t)n_parts * win->n_windows;
    std::vector<int64_t> &offs = ctx->host_i64["partition.group_offsets"];
    offs.assign(h_off, h_off + n_groups + 1);
    out->row = d_rows;
    out->part_win_offsets = offs.data();
    out->rows = offs[n_groups];
    return FLOCKGPU_OK;
}

int flockgpu_take_i32(flockgpu_ctx *ctx, const int32_t *src, const int32_t *rows, int64_t n, int32_t *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (n < 0 || (n > 0 && (!src || !rows || !out))) return fail(ctx, FLOCKGPU_ERR_INVALID, "take_i32: null argument");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    return gather_i32(ctx, src, rows, n, out);
}

int flockgpu_take_i64(flockgpu_ctx *ctx, const int64_t *src, const int32_t *rows, int64_t n, int64_t *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (n < 0 || (n > 0 && (!src || !rows || !out))) return fail(ctx, FLOCKGPU_ERR_INVALID, "take_i64: null argument");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    return gather_i64(ctx, src, rows, n, out);
}

int flockgpu_take_utf8(flockgpu_ctx *ctx, const flockgpu_utf8 *src, const int32_t *rows, int64_t n, int32_t slot,
                       flockgpu_utf8 *out, int64_t *out_bytes) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!src || !out || !out_bytes || n < 0 || (n > 0 && (!rows || !src->offsets || !src->data)))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "take_utf8: null argument");
    if (slot < 0 || slot > 15) return fail(ctx, FLOCKGPU_ERR_INVALID, "take_utf8: slot must be in [0, 15]");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    char name[32];
    snprintf(name, sizeof name, "take_utf8.%d", slot);
    return gather_utf8(ctx, name, *src, rows, n, out, out_bytes);
}

int flockgpu_inclusive_scan_i32(flockgpu_ctx *ctx, int32_t *data, int64_t n) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (n < 0 || (n > 0 && !data)) return fail(ctx, FLOCKGPU_ERR_INVALID, "inclusive_scan_i32: null argument");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    return inclusive_scan_i32(ctx, "abi.scan", data, n);
}

}  // extern "C"
