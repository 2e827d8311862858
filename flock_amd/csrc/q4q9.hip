// NEXMark q4 / q9 for gfx950 ("remaining join / aggregate queries on the same kernels"),
// per ElementWise window (benchmarks/src/nexmark/main.rs:117):
//   Q    : SELECT a_id, [category,] MAX(price) AS final FROM auction INNER JOIN bid ON a_id = auction
//          WHERE b_date_time BETWEEN a_date_time AND expires GROUP BY a_id [, category]
//   q9   : bid JOIN Q ON auction = id AND price = final  ->  [auction, bidder, price, b_date_time]   (q9.sql, q9_plan.fmt)
//   q4   : SELECT category, AVG(final) FROM Q GROUP BY category                                       (q4.sql; stages in
//          flock/src/distributed_plan/planner.rs:218-256)
//
// HBM-bound integer work, no MFMA.  The join table is a direct-address array over the window's a_id range (auction ids
// are strictly increasing inside a window: verified per call, FLOCKGPU_ERR_UNSUPPORTED otherwise -- the host keeps
// DataFusion for such input).
//   stats / build : as q3's dense path -- direct[a_id - min] = auction row, final[row] = "none"
//   final         : ONE pass over the bids (auction, price, b_date_time = 16 B / bid): row lookup, BETWEEN against the
//                   auction's two timestamps, then MAX(price) per auction row.  Half of all bids hit one auction
//                   (event.rs:355-359): a tile's matched rows span a few hundred auctions, so the workgroup
//                   pre-aggregates in an LDS max-table (the hot row's maximum stays in a register per lane and is folded
//                   in once per tile) and issues one fire-and-forget global atomicMax per distinct auction per tile.
//   q9            : second pass over (auction, price): flag `price = final[row]` (the outer join does NOT repeat the
//                   BETWEEN), tile scan, emit the four bid columns in input order.
//   q4            : pass over the auctions: (category, final) -> per-category {sum, count} (block-level LDS
//                   accumulation, one atomic pair per category per block); AVG = sum / count as one IEEE division.
//                   DataFusion's AVG keeps a Float64 running sum: integer-valued partial sums below 2^53 are exact, so
//                   the order of accumulation cannot change a bit of the result.
#include <algorithm>

#include "gather.hpp"

using namespace flockgpu;

namespace {

constexpr int32_t kNone = (int32_t)0x80000000;  // "no valid bid": prices are compared as Int32, MAX of nothing is NULL
constexpr int kSpan = 4096;                     // auction rows an LDS-staged tile may span
constexpr int kMaxCategories = 64;              // per-window category range handled by the LDS accumulators

struct WinTable {
    int32_t base;       // min a_id of the window
    uint32_t range;     // max - min + 1 (0: the window has no auctions)
    uint64_t off;       // offset of the window's entries in the table arena
    int32_t first_row;  // row of the window's first auction
    int32_t gapless;    // range == rows: ids are consecutive, so  row = first_row + (a_id - base)  without any lookup
};

__global__ __launch_bounds__(kBlock) void aq_build_kernel(const int32_t *__restrict__ a_id, int64_t n_rows, SegTiles st,
                                                          const WinTable *__restrict__ wins, int32_t *direct,
                                                          int32_t *__restrict__ final_price) {
    const TileRange tr = locate_tile(st, (int32_t)blockIdx.x, kFlagTile);
    const WinTable wt = wins[tr.seg];
    int32_t a[kFlagIters][4];
    load_flag_tile(a_id, n_rows, tr, a);
    const int64_t wbase = tr.tile_begin + flag_rel0();
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = wbase + it * 256 + j;
            if (r >= tr.lo && r < tr.hi) {
                direct[wt.off + (uint32_t)(a[it][j] - wt.base)] = (int32_t)r;
                final_price[r] = kNone;
            }
        }
}

// Auction row of a bid's `auction` key inside its window, or -1.  Unconditional load from a clamped index.
__device__ __forceinline__ int32_t auction_row_of(int32_t key, const WinTable &wt, const int32_t *__restrict__ tab, bool in) {
    const uint32_t idx = (uint32_t)key - (uint32_t)wt.base;
    const bool ok = in && idx < wt.range;
    if (wt.gapless) return ok ? wt.first_row + (int32_t)idx : -1;  // (block-uniform) the generator's consecutive ids
    const int32_t row = tab[ok ? idx : 0u];
    return ok ? row : -1;
}

// kIters x 1024 bids per tile: lane l of wave w holds rows  w*256*kIters + it*256 + 4l .. 4l+3  (the flag-tile layout
// with fewer iterations: (row, price) of every bid stay in registers between the two phases, and 4 iterations instead
// of 8 keep the kernel at 6 instead of 4 workgroups per CU).
constexpr int kFinalIters = 4;
constexpr int kFinalTile = kBlock * 4 * kFinalIters;
__global__ __launch_bounds__(kBlock) void aq_final_kernel(const int32_t *__restrict__ b_auction,
                                                          const int32_t *__restrict__ b_price,
                                                          const int64_t *__restrict__ b_time, int64_t n_bids, SegTiles st,
                                                          const WinTable *__restrict__ wins, const int32_t *__restrict__ direct,
                                                          const int64_t *__restrict__ a_time, const int64_t *__restrict__ a_expires,
                                                          int32_t *final_price) {
    __shared__ int32_t s_max[kSpan];
    __shared__ int32_t s_red[2 * kWavesPerBlock];
    for (int s = threadIdx.x; s < kSpan; s += kBlock) s_max[s] = kNone;
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFinalTile);
    const WinTable wt = wins[tr.seg];
    if (wt.range == 0) return;  // the window has no auctions: nothing can join
    const int32_t *tab = direct + wt.off;
    const int64_t wbase = tr.tile_begin + (int64_t)(threadIdx.x >> 6) * (kFinalTile / kWavesPerBlock) + lane_id() * 4;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    int32_t row[kFinalIters][4], price[kFinalIters][4];
    int32_t mn = 0x7fffffff, mx = -1;
#pragma unroll
    for (int it = 0; it < kFinalIters; ++it) {
        const int64_t r0 = wbase + it * 256;
        int32_t key[4];
        load4_i32(b_auction, r0, n_bids, key);
        load4_i32(b_price, r0, n_bids, price[it]);
        int64_t when[4];
        if (r0 >= 0 && r0 + 4 <= n_bids) {  // four timestamps = two 16-byte loads (r0 is a multiple of 4)
            const longlong2 w01 = *reinterpret_cast<const longlong2 *>(b_time + r0),
                            w23 = *reinterpret_cast<const longlong2 *>(b_time + r0 + 2);
            when[0] = w01.x; when[1] = w01.y; when[2] = w23.x; when[3] = w23.y;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) when[j] = (r0 + j >= 0 && r0 + j < n_bids) ? b_time[r0 + j] : 0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + j;
            row[it][j] = auction_row_of(key[j], wt, tab, r >= tr.lo && r < tr.hi);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t ar = row[it][j] < 0 ? 0 : row[it][j];  // clamped: loads stay unconditional
            const int64_t t0 = a_time[ar], t1 = a_expires[ar];
            if (!(row[it][j] >= 0 && when[j] >= t0 && when[j] <= t1)) row[it][j] = -1;
            if (row[it][j] >= 0) {
                mn = min(mn, row[it][j]);
                mx = max(mx, row[it][j]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = min(mn, __shfl_xor(mn, o, 64));
        mx = max(mx, __shfl_xor(mx, o, 64));
    }
    if (lane == 0) {
        s_red[wave] = mn;
        s_red[kWavesPerBlock + wave] = mx;
    }
    __syncthreads();  // also orders the initialisation of s_max
    mn = min(min(s_red[0], s_red[1]), min(s_red[2], s_red[3]));
    mx = max(max(s_red[4], s_red[5]), max(s_red[6], s_red[7]));
    if (mx < 0) return;  // no bid of the tile survives the join + BETWEEN
    const bool staged = (uint32_t)(mx - mn) < (uint32_t)kSpan;  // block-uniform
    // the hot auction of this wave: its maximum is kept per lane in a register and folded in once
    int32_t hot = -2, hot_max = kNone;
#pragma unroll
    for (int it = 0; it < kFinalIters; ++it) {
        uint64_t m = __ballot(row[it][0] == hot);
        if (__popcll((unsigned long long)m) < 16) {  // re-elect (see q8_sellers_bitmap_kernel)
            if (hot >= 0) {  // park the outgoing candidate's maximum
                int32_t v = hot_max;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
                if (lane == 0 && v != kNone) {
                    if (staged) atomicMax(&s_max[hot - mn], v);
                    else atomicMax(&final_price[hot], v);
                }
                hot_max = kNone;
            }
            const uint64_t live = __ballot(row[it][0] >= 0);
            hot = -2;
            if (live) {
                const int l1 = __ffsll((unsigned long long)live) - 1;
                const int32_t c1 = __builtin_amdgcn_readlane(row[it][0], l1);
                const uint64_t m1 = __ballot(row[it][0] == c1);
                hot = c1;
                const uint64_t rest = live & ~m1;
                if (__popcll((unsigned long long)m1) < 16 && rest) {
                    const int l2 = __ffsll((unsigned long long)rest) - 1;
                    const int32_t c2 = __builtin_amdgcn_readlane(row[it][0], l2);
                    if (__popcll((unsigned long long)__ballot(row[it][0] == c2)) > __popcll((unsigned long long)m1)) hot = c2;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t ar = row[it][j];
            if (ar < 0) continue;
            if (ar == hot) {
                hot_max = max(hot_max, price[it][j]);
            } else if (staged) {
                atomicMax(&s_max[ar - mn], price[it][j]);
            } else {
                atomicMax(&final_price[ar], price[it][j]);
            }
        }
    }
    if (hot >= 0) {
        int32_t v = hot_max;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
        if (lane == 0 && v != kNone) {
            if (staged) atomicMax(&s_max[hot - mn], v);
            else atomicMax(&final_price[hot], v);
        }
    }
    if (!staged) return;
    __syncthreads();
    for (int32_t s = threadIdx.x; s <= mx - mn; s += kBlock) {
        const int32_t v = s_max[s];
        if (v != kNone) __hip_atomic_fetch_max(&final_price[mn + s], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// q9 outer join: bids whose price equals their auction's final (no BETWEEN here: q9.sql joins bid with Q on
// auction = id AND price = final only).
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) void q9_flag_kernel(const int32_t *__restrict__ b_auction,
                                                         const int32_t *__restrict__ b_price, int64_t n_bids, SegTiles st,
                                                         const WinTable *__restrict__ wins, const int32_t *__restrict__ direct,
                                                         const int32_t *__restrict__ final_price,
                                                         uint32_t *__restrict__ flag_words, uint32_t *__restrict__ counts) {
    int32_t tile = (int32_t)blockIdx.x;
    if (tile >= st.n_tiles) return;
    TileRange tr = locate_tile(st, tile, kFlagTile);
    const int32_t rel0 = flag_rel0();
#pragma unroll 1
    for (;;) {  // tiles b, b + G, ... with the next descriptor requested early (scan.hpp)
        int32_t key[kFlagIters][4], price[kFlagIters][4];
        load_flag_tile(b_auction, n_bids, tr, key);
        load_flag_tile(b_price, n_bids, tr, price);
        const WinTable wt = wins[tr.seg];
        const int32_t next = tile + (int32_t)gridDim.x;
        TileRange trn = tr;
        if (next < st.n_tiles) trn = locate_tile(st, next, kFlagTile);
        const int32_t rel_lo = (int32_t)(tr.lo - tr.tile_begin), rel_hi = (int32_t)(tr.hi - tr.tile_begin);
        const int32_t *tab = direct + wt.off;
        uint32_t flags = 0;
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int32_t rel = rel0 + it * 256 + j;
                const int32_t ar = auction_row_of(key[it][j], wt, tab, rel >= rel_lo && rel < rel_hi && wt.range != 0);
                const int32_t fin = final_price[ar < 0 ? 0 : ar];
                const bool f = ar >= 0 && fin != kNone && fin == price[it][j];
                flags |= (f ? 1u : 0u) << (it * 4 + j);
            }
        store_flags_and_counts(flags, tile, flag_words, counts);
        if (next >= st.n_tiles) break;
        tile = next;
        tr = trn;
    }
}

// q4: per window and category {sum of finals, number of auctions with a final}.  acc[(win * kMaxCategories + c) * 2 + {0,1}]
// The first kRegCats category values of a window are accumulated in registers (one compare chain per row, no memory
// traffic: a window has 5 categories, config.rs:130,141 -- 8192 LDS atomics on 5 words would serialise), the rest in LDS.
constexpr int kRegCats = 8;
__global__ __launch_bounds__(kBlock) void q4_category_kernel(const int32_t *__restrict__ category,
                                                             const int32_t *__restrict__ final_price, int64_t n_rows,
                                                             SegTiles st, const int32_t *__restrict__ cat_min,
                                                             unsigned long long *acc) {
    __shared__ unsigned long long s_acc[2 * kMaxCategories];
    for (int s = threadIdx.x; s < 2 * kMaxCategories; s += kBlock) s_acc[s] = 0;
    __syncthreads();
    const TileRange tr = locate_tile(st, (int32_t)blockIdx.x, kFlagTile);
    const int32_t cmin = cat_min[tr.seg];
    int32_t c[kFlagIters][4], f[kFlagIters][4];
    load_flag_tile(category, n_rows, tr, c);
    load_flag_tile(final_price, n_rows, tr, f);
    const int32_t rel_lo = (int32_t)(tr.lo - tr.tile_begin), rel_hi = (int32_t)(tr.hi - tr.tile_begin);
    const int32_t rel0 = flag_rel0();
    long long sum[kRegCats];
    uint32_t cnt[kRegCats];
#pragma unroll
    for (int k = 0; k < kRegCats; ++k) {
        sum[k] = 0;
        cnt[k] = 0;
    }
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t rel = rel0 + it * 256 + j;
            if (!(rel >= rel_lo && rel < rel_hi && f[it][j] != kNone)) continue;
            const int32_t k = c[it][j] - cmin;  // 0 <= k < kMaxCategories (checked on the host)
#pragma unroll
            for (int q = 0; q < kRegCats; ++q)
                if (k == q) {
                    sum[q] += f[it][j];
                    cnt[q] += 1;
                }
            if (k >= kRegCats) {  // finals are any Int32 in general: sum as two's-complement Int64
                atomicAdd(&s_acc[2 * k], (unsigned long long)(long long)f[it][j]);
                atomicAdd(&s_acc[2 * k + 1], 1ull);
            }
        }
#pragma unroll
    for (int q = 0; q < kRegCats; ++q) {
        const uint64_t s64 = wave_sum_u64((uint64_t)sum[q]);
        const uint64_t c64 = wave_sum_u64((uint64_t)cnt[q]);
        if (lane_id() == 0 && c64) {
            atomicAdd(&s_acc[2 * q], (unsigned long long)s64);
            atomicAdd(&s_acc[2 * q + 1], (unsigned long long)c64);
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < 2 * kMaxCategories; s += kBlock)
        if (s_acc[s]) atomicAdd(&acc[(size_t)tr.seg * 2 * kMaxCategories + s], s_acc[s]);
}

__global__ __launch_bounds__(kBlock) void fill_i32_kernel(int32_t *p, int32_t v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = v;
}

// Everything q4 and q9 share: tables, finals.  Leaves device state in `s`.
struct Shared {
    SegTiles st_a, st_b, st_bf;  // auctions, bids (flag tiles), bids (tiles of the final pass)
    WinTable *d_wins = nullptr;
    int32_t *direct = nullptr, *final_price = nullptr;
    int n_win = 0;
};

int compute_finals(flockgpu_ctx *ctx, const char *who, const flockgpu_auction_time_cols *auction, const flockgpu_windows *auction_win,
                   const flockgpu_bid_cols *bid, const flockgpu_windows *bid_win, Shared *s) {
    if (!auction || !bid || auction->rows < 0 || bid->rows < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: null argument", who);
    FG_TRY(check_windows(ctx, auction_win, auction->rows, who));
    FG_TRY(check_windows(ctx, bid_win, bid->rows, who));
    if (auction_win->n_windows != bid_win->n_windows)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: auction and bid schedules differ in window count", who);
    if (auction->rows >= (int64_t(1) << 31) || bid->rows >= (int64_t(1) << 31))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: relations are limited to 2^31 rows per call", who);
    if (auction->rows > 0 && (!auction->a_id || !auction->a_date_time || !auction->expires))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: null auction column", who);
    if (bid->rows > 0 && (!bid->auction || !bid->price || !bid->b_date_time))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: null bid column", who);
    if ((reinterpret_cast<uintptr_t>(auction->a_id) & 15) || (reinterpret_cast<uintptr_t>(bid->auction) & 15) ||
        (reinterpret_cast<uintptr_t>(bid->price) & 15) || (reinterpret_cast<uintptr_t>(bid->b_date_time) & 15))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: a_id, auction, price and b_date_time columns must be 16-byte aligned", who);
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_win = s->n_win = auction_win->n_windows;
    std::vector<int64_t> ab(n_win), ae(n_win), bb(n_win), be(n_win);
    for (int w = 0; w < n_win; ++w) {
        ab[w] = auction_win->pane_row_offsets[auction_win->win_pane_lo[w]];
        ae[w] = auction_win->pane_row_offsets[auction_win->win_pane_hi[w]];
        bb[w] = bid_win->pane_row_offsets[bid_win->win_pane_lo[w]];
        be[w] = bid_win->pane_row_offsets[bid_win->win_pane_hi[w]];
    }
    FG_TRY(build_seg_tiles(ctx, "aq.auction", ab.data(), ae.data(), n_win, kFlagTile, &s->st_a));
    FG_TRY(build_seg_tiles(ctx, "aq.bid", bb.data(), be.data(), n_win, kFlagTile, &s->st_b));
    FG_TRY(build_seg_tiles(ctx, "aq.bid_final", bb.data(), be.data(), n_win, kFinalTile, &s->st_bf));
    int32_t *d_stats = nullptr, *h_stats = nullptr;
    FG_TRY(arena_get_t(ctx, "aq.stats", (size_t)3 * std::max(n_win, 1), &d_stats));
    FG_TRY(pinned_get_t(ctx, "aq.stats", (size_t)3 * std::max(n_win, 1), &h_stats));
    FG_TRY(segment_key_stats(ctx, auction->a_id, auction->rows, s->st_a, d_stats, d_stats + n_win, d_stats + 2 * n_win));
    if (n_win > 0) FG_HIP(ctx, hipMemcpyAsync(h_stats, d_stats, sizeof(int32_t) * 3 * n_win, hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t n_entries = 0;
    std::vector<WinTable> wins(std::max(n_win, 1));
    for (int w = 0; w < n_win; ++w) {
        wins[w] = WinTable{0, 0, n_entries, 0, 0};
        if (ae[w] == ab[w]) continue;
        const int64_t mn = h_stats[w], mx = h_stats[n_win + w], range = mx - mn + 1;
        if (!h_stats[2 * n_win + w] || range > 8 * (ae[w] - ab[w]) + 1024)
            return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED,
                        "%s: window %d: auction ids must be strictly increasing over a dense range (duplicate / unsorted / sparse "
                        "a_id is left to the host engine)", who, w);
        wins[w].base = (int32_t)mn;
        wins[w].range = (uint32_t)range;
        wins[w].first_row = (int32_t)ab[w];
        wins[w].gapless = range == ae[w] - ab[w];
        n_entries += (uint64_t)range;
    }
    WinTable *h_wins = nullptr;
    FG_TRY(arena_get_t(ctx, "aq.wins", (size_t)std::max(n_win, 1), &s->d_wins));
    FG_TRY(pinned_get_t(ctx, "aq.wins", (size_t)std::max(n_win, 1), &h_wins));
    std::copy(wins.begin(), wins.begin() + n_win, h_wins);
    FG_TRY(arena_get_t(ctx, "aq.direct", (size_t)n_entries + 4, &s->direct));
    FG_TRY(arena_get_t(ctx, "aq.final", (size_t)auction->rows + 4, &s->final_price));
    if (n_win > 0) FG_HIP(ctx, hipMemcpyAsync(s->d_wins, h_wins, sizeof(WinTable) * n_win, hipMemcpyHostToDevice, ctx->stream));
    FG_HIP(ctx, hipMemsetAsync(s->direct, 0xFF, sizeof(int32_t) * ((size_t)n_entries + 4), ctx->stream));
    {   // rows that belong to no window keep "none" too (q4 reads final[] by tiles of windows only, but be defined)
        const int64_t n = auction->rows + 4;
        hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)div_up(n, kBlock)), dim3(kBlock), 0, ctx->stream, s->final_price, kNone, n);
        FG_TRY(check_launch(ctx, "fill_i32_kernel"));
    }
    if (s->st_a.n_tiles > 0) {
        LaunchScope ls(ctx, "aq_build_kernel");
        hipLaunchKernelGGL(aq_build_kernel, dim3((unsigned)s->st_a.n_tiles), dim3(kBlock), 0, ctx->stream, auction->a_id,
                           auction->rows, s->st_a, s->d_wins, s->direct, s->final_price);
    }
    FG_TRY(check_launch(ctx, "aq_build_kernel"));
    if (s->st_bf.n_tiles > 0) {
        LaunchScope ls(ctx, "aq_final_kernel");
        hipLaunchKernelGGL(aq_final_kernel, dim3((unsigned)s->st_bf.n_tiles), dim3(kBlock), 0, ctx->stream, bid->auction, bid->price,
                           bid->b_date_time, bid->rows, s->st_bf, s->d_wins, s->direct, auction->a_date_time, auction->expires,
                           s->final_price);
    }
    return check_launch(ctx, "aq_final_kernel");
}

}  // namespace

extern "C" {

int flockgpu_q9_winning_bids(flockgpu_ctx *ctx, const flockgpu_auction_time_cols *auction, const flockgpu_windows *auction_win,
                             const flockgpu_bid_cols *bid, const flockgpu_windows *bid_win, flockgpu_q9_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!out) return fail(ctx, FLOCKGPU_ERR_INVALID, "q9: null result");
    if (bid && bid->rows > 0 && !bid->bidder) return fail(ctx, FLOCKGPU_ERR_INVALID, "q9: null bidder column");
    Shared s;
    FG_TRY(compute_finals(ctx, "q9", auction, auction_win, bid, bid_win, &s));
    const int n_win = s.n_win;
    uint32_t *flag_words = nullptr, *counts = nullptr;
    uint64_t *tile_base = nullptr;
    int64_t *d_off = nullptr, *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, "q9.flag_words", (size_t)s.st_b.n_tiles * kBlock, &flag_words));
    FG_TRY(arena_get_t(ctx, "q9.counts", (size_t)s.st_b.n_tiles * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, "q9.tile_base", (size_t)s.st_b.n_tiles + 1, &tile_base));
    FG_TRY(arena_get_t(ctx, "q9.seg_out_off", (size_t)n_win + 1, &d_off));
    FG_TRY(pinned_get_t(ctx, "q9.seg_out_off", (size_t)n_win + 1, &h_off));
    if (s.st_b.n_tiles > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>(s.st_b.n_tiles, (int64_t)ctx->num_cus * kStreamBlocksPerCu);
        LaunchScope ls(ctx, "q9_flag_kernel");
        hipLaunchKernelGGL(q9_flag_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, bid->auction, bid->price, bid->rows, s.st_b,
                           s.d_wins, s.direct, s.final_price, flag_words, counts);
    }
    FG_TRY(check_launch(ctx, "q9_flag_kernel"));
    FG_TRY(launch_tile_scan(ctx, counts, s.st_b.n_tiles, tile_base, s.st_b.tile_first, s.st_b.n_seg, d_off));
    FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * ((size_t)n_win + 1), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> &offs = ctx->host_i64["q9.win_out_offsets"];
    offs.assign(h_off, h_off + n_win + 1);
    const int64_t n_out = offs[n_win];
    int32_t *o_a = nullptr, *o_p = nullptr, *o_b = nullptr;
    int64_t *o_t = nullptr;
    FG_TRY(arena_get_t(ctx, "q9.out_auction", (size_t)n_out + 1, &o_a));
    FG_TRY(arena_get_t(ctx, "q9.out_price", (size_t)n_out + 1, &o_p));
    FG_TRY(arena_get_t(ctx, "q9.out_bidder", (size_t)n_out + 1, &o_b));
    FG_TRY(arena_get_t(ctx, "q9.out_time", (size_t)n_out + 1, &o_t));
    if (n_out > 0) FG_TRY(emit_flagged_bids(ctx, s.st_b, flag_words, counts, tile_base, *bid, o_a, o_p, o_b, o_t));
    out->auction = o_a;
    out->price = o_p;
    out->bidder = o_b;
    out->b_date_time = o_t;
    out->win_out_offsets = offs.data();
    out->rows = n_out;
    return FLOCKGPU_OK;
}

int flockgpu_q4_avg_final_by_category(flockgpu_ctx *ctx, const flockgpu_auction_time_cols *auction,
                                      const flockgpu_windows *auction_win, const flockgpu_bid_cols *bid,
                                      const flockgpu_windows *bid_win, flockgpu_q4_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!out) return fail(ctx, FLOCKGPU_ERR_INVALID, "q4: null result");
    if (auction && auction->rows > 0 && !auction->category) return fail(ctx, FLOCKGPU_ERR_INVALID, "q4: null category column");
    if (auction && (reinterpret_cast<uintptr_t>(auction->category) & 15))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q4: category column must be 16-byte aligned");
    Shared s;
    FG_TRY(compute_finals(ctx, "q4", auction, auction_win, bid, bid_win, &s));
    const int n_win = s.n_win;
    // category range per window (exact)
    int32_t *d_cs = nullptr, *h_cs = nullptr;
    FG_TRY(arena_get_t(ctx, "q4.cat_stats", (size_t)3 * std::max(n_win, 1), &d_cs));
    FG_TRY(pinned_get_t(ctx, "q4.cat_stats", (size_t)3 * std::max(n_win, 1), &h_cs));
    FG_TRY(segment_key_stats(ctx, auction->category, auction->rows, s.st_a, d_cs, d_cs + n_win, d_cs + 2 * n_win));
    if (n_win > 0) FG_HIP(ctx, hipMemcpyAsync(h_cs, d_cs, sizeof(int32_t) * 3 * n_win, hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int w = 0; w < n_win; ++w)
        if (h_cs[w] <= h_cs[n_win + w] && (int64_t)h_cs[n_win + w] - h_cs[w] >= kMaxCategories)
            return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q4: window %d spans more than %d category values", w, kMaxCategories);
    const size_t n_acc = (size_t)std::max(n_win, 1) * 2 * kMaxCategories;
    unsigned long long *d_acc = nullptr, *h_acc = nullptr;
    FG_TRY(arena_get_t(ctx, "q4.acc", n_acc, &d_acc));
    FG_TRY(pinned_get_t(ctx, "q4.acc", n_acc, &h_acc));
    FG_HIP(ctx, hipMemsetAsync(d_acc, 0, sizeof(unsigned long long) * n_acc, ctx->stream));
    if (s.st_a.n_tiles > 0) {
        LaunchScope ls(ctx, "q4_category_kernel");
        hipLaunchKernelGGL(q4_category_kernel, dim3((unsigned)s.st_a.n_tiles), dim3(kBlock), 0, ctx->stream, auction->category,
                           s.final_price, auction->rows, s.st_a, d_cs, d_acc);
    }
    FG_TRY(check_launch(ctx, "q4_category_kernel"));
    FG_HIP(ctx, hipMemcpyAsync(h_acc, d_acc, sizeof(unsigned long long) * n_acc, hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> &offs = ctx->host_i64["q4.win_out_offsets"];
    offs.assign((size_t)n_win + 1, 0);
    std::vector<int32_t> cats;
    std::vector<double> avgs;
    for (int w = 0; w < n_win; ++w) {
        for (int k = 0; k < kMaxCategories; ++k) {
            const unsigned long long cnt = h_acc[((size_t)w * kMaxCategories + k) * 2 + 1];
            if (!cnt) continue;
            const long long sum = (long long)h_acc[((size_t)w * kMaxCategories + k) * 2];
            cats.push_back(h_cs[w] + k);
            avgs.push_back((double)sum / (double)cnt);  // AVG state {count UInt64, sum Float64}: exact below 2^53
        }
        offs[w + 1] = (int64_t)cats.size();
    }
    const size_t n_out = cats.size();
    int32_t *d_cat = nullptr, *h_cat = nullptr;
    double *d_avg = nullptr, *h_avg = nullptr;
    FG_TRY(arena_get_t(ctx, "q4.out_category", n_out + 1, &d_cat));
    FG_TRY(arena_get_t(ctx, "q4.out_avg", n_out + 1, &d_avg));
    FG_TRY(pinned_get_t(ctx, "q4.out_category", n_out + 1, &h_cat));
    FG_TRY(pinned_get_t(ctx, "q4.out_avg", n_out + 1, &h_avg));
    std::copy(cats.begin(), cats.end(), h_cat);
    std::copy(avgs.begin(), avgs.end(), h_avg);
    if (n_out) {
        FG_HIP(ctx, hipMemcpyAsync(d_cat, h_cat, sizeof(int32_t) * n_out, hipMemcpyHostToDevice, ctx->stream));
        FG_HIP(ctx, hipMemcpyAsync(d_avg, h_avg, sizeof(double) * n_out, hipMemcpyHostToDevice, ctx->stream));
    }
    out->category = d_cat;
    out->avg_final = d_avg;
    out->win_out_offsets = offs.data();
    out->rows = (int64_t)n_out;
    return FLOCKGPU_OK;
}

}  // extern "C"
