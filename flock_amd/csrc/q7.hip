// NEXMark q7 "highest bid" for gfx950: per Tumbling(10 s) window
//   SELECT auction, price, bidder, b_date_time FROM bid JOIN (SELECT MAX(price) AS maxprice FROM bid) B1 ON price = maxprice
// (benchmarks/src/nexmark/query/q7.sql, q7_plan.fmt; window: benchmarks/src/nexmark/main.rs:119).
// First of SURVEY.md section 8(f)'s "next" queries; it reuses the q2 machinery (flag tiles, count -> scan -> emit).
//
// HBM-bound integer work, no MFMA.  ONE streaming pass over `price` (4 B / bid):
//   max  : per tile the maximum of its rows (kept: tile_max[tile]) and, by atomicMax, the window's maximum
//   flag : only tiles whose maximum IS the window's maximum can hold result rows -- every other tile writes zero
//          counts without reading a byte; the few remaining tiles re-read their 32 KiB and flag `price = maxprice`
//   scan / emit : as q2 -- the four columns of the surviving rows (gather.hip: emit_bids_kernel), input order kept,
//                 ties all returned
// An empty window has MAX = NULL and the inner join emits nothing.
#include <algorithm>

#include "gather.hpp"

using namespace flockgpu;

namespace {

constexpr int32_t kNoMax = (int32_t)0x80000000;

// Persistent: block b reduces tiles b, b + G, ...; the NEXT tile's descriptor is requested before the current tile's
// rows are reduced, so the column loads of a tile never wait for a dependent descriptor load.
__global__ __launch_bounds__(kBlock) void q7_max_kernel(const int32_t *__restrict__ price, int64_t n_rows, SegTiles st,
                                                        int32_t *__restrict__ tile_max) {
    __shared__ int32_t s_red[2][kWavesPerBlock];
    int32_t tile = (int32_t)blockIdx.x;
    if (tile >= st.n_tiles) return;
    TileRange tr = locate_tile(st, tile, kFlagTile);
    const int32_t rel0 = flag_rel0();
    int par = 0;
#pragma unroll 1
    for (;;) {
        int32_t a[kFlagIters][4];
        load_flag_tile(price, n_rows, tr, a);
        const int32_t next = tile + (int32_t)gridDim.x;
        TileRange trn = tr;
        if (next < st.n_tiles) trn = locate_tile(st, next, kFlagTile);
        const int32_t rel_lo = (int32_t)(tr.lo - tr.tile_begin), rel_hi = (int32_t)(tr.hi - tr.tile_begin);
        int32_t mx = kNoMax;
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int32_t rel = rel0 + it * 256 + j;
                if (rel >= rel_lo && rel < rel_hi) mx = max(mx, a[it][j]);
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
        if (lane_id() == 0) s_red[par][threadIdx.x >> 6] = mx;
        __syncthreads();  // (s_red is double-buffered: one barrier per tile is enough)
        if (threadIdx.x == 0) {
            mx = max(max(s_red[par][0], s_red[par][1]), max(s_red[par][2], s_red[par][3]));
            tile_max[tile] = mx;
        }
        if (next >= st.n_tiles) break;
        tile = next;
        tr = trn;
        par ^= 1;
    }
}

// win_max[w] = maximum over the window's tiles: one workgroup per window over tile_max (a window's ~1100 tiles of a
// 1e9-bid run would otherwise be as many atomics on one address, issued by workgroups that run at the same time).
__global__ __launch_bounds__(kBlock) void q7_window_max_kernel(const int32_t *__restrict__ tile_max, const int32_t *__restrict__ tile_first,
                                                               int32_t *__restrict__ win_max) {
    __shared__ int32_t s_red[kWavesPerBlock];
    const int32_t w = (int32_t)blockIdx.x;
    int32_t mx = kNoMax;
    for (int32_t t = tile_first[w] + (int32_t)threadIdx.x; t < tile_first[w + 1]; t += kBlock) mx = max(mx, tile_max[t]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    if (lane_id() == 0) s_red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) win_max[w] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
}

__global__ __launch_bounds__(kBlock) void q7_flag_kernel(const int32_t *__restrict__ price, int64_t n_rows, SegTiles st,
                                                         const int32_t *__restrict__ tile_max,
                                                         const int32_t *__restrict__ win_max,
                                                         uint32_t *__restrict__ flag_words, uint32_t *__restrict__ counts) {
    const int32_t tile = (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const int32_t mx = win_max[tr.seg];
    if (tr.hi <= tr.lo || tile_max[tile] != mx) {  // block-uniform: no row of this tile reaches the maximum
        if (threadIdx.x < kWavesPerBlock) counts[(size_t)tile * kWavesPerBlock + threadIdx.x] = 0;
        return;
    }
    int32_t a[kFlagIters][4];
    load_flag_tile(price, n_rows, tr, a);
    const int32_t rel_lo = (int32_t)(tr.lo - tr.tile_begin), rel_hi = (int32_t)(tr.hi - tr.tile_begin);
    const int32_t rel0 = flag_rel0();
    uint32_t flags = 0;
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t rel = rel0 + it * 256 + j;
            const bool f = a[it][j] == mx && rel >= rel_lo && rel < rel_hi;
            flags |= (f ? 1u : 0u) << (it * 4 + j);
        }
    store_flags_and_counts(flags, tile, flag_words, counts);
}

__global__ __launch_bounds__(kBlock) void fill_i32_kernel(int32_t *p, int32_t v, int32_t n) {
    const int32_t i = (int32_t)(blockIdx.x * kBlock + threadIdx.x);
    if (i < n) p[i] = v;
}

}  // namespace

extern "C" {

int flockgpu_q7_highest_bid(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                            flockgpu_q7_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!bid || !out || bid->rows < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "q7: null argument");
    FG_TRY(check_windows(ctx, win, bid->rows, "q7"));
    if (bid->rows > 0 && (!bid->auction || !bid->price || !bid->bidder || !bid->b_date_time))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q7: null column (the projection keeps all four bid columns)");
    if (reinterpret_cast<uintptr_t>(bid->price) & 15) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q7: price column must be 16-byte aligned");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_win = win->n_windows;
    std::vector<int64_t> sb(n_win), se(n_win);
    for (int w = 0; w < n_win; ++w) {
        sb[w] = win->pane_row_offsets[win->win_pane_lo[w]];
        se[w] = win->pane_row_offsets[win->win_pane_hi[w]];
    }
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, "q7", sb.data(), se.data(), n_win, kFlagTile, &st));
    int32_t *tile_max = nullptr, *d_wmax = nullptr, *h_wmax = nullptr;
    uint32_t *flag_words = nullptr, *counts = nullptr;
    uint64_t *tile_base = nullptr;
    int64_t *d_off = nullptr, *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, "q7.tile_max", (size_t)st.n_tiles + 1, &tile_max));
    FG_TRY(arena_get_t(ctx, "q7.win_max", (size_t)n_win + 1, &d_wmax));
    FG_TRY(pinned_get_t(ctx, "q7.win_max", (size_t)n_win + 1, &h_wmax));
    FG_TRY(arena_get_t(ctx, "q7.flag_words", (size_t)st.n_tiles * kBlock, &flag_words));
    FG_TRY(arena_get_t(ctx, "q7.counts", (size_t)st.n_tiles * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, "q7.tile_base", (size_t)st.n_tiles + 1, &tile_base));
    FG_TRY(arena_get_t(ctx, "q7.seg_out_off", (size_t)n_win + 1, &d_off));
    FG_TRY(pinned_get_t(ctx, "q7.seg_out_off", (size_t)n_win + 1, &h_off));
    if (n_win > 0) {
        hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)div_up(n_win, kBlock)), dim3(kBlock), 0, ctx->stream, d_wmax, kNoMax, n_win);
        FG_TRY(check_launch(ctx, "fill_i32_kernel"));
    }
    if (st.n_tiles > 0) {
        {
            LaunchScope ls(ctx, "q7_max_kernel");
            const unsigned grid = (unsigned)std::min<int64_t>(st.n_tiles, (int64_t)ctx->num_cus * kStreamBlocksPerCu);
            hipLaunchKernelGGL(q7_max_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, bid->price, bid->rows, st, tile_max);
        }
        FG_TRY(check_launch(ctx, "q7_max_kernel"));
        hipLaunchKernelGGL(q7_window_max_kernel, dim3((unsigned)n_win), dim3(kBlock), 0, ctx->stream, tile_max, st.tile_first, d_wmax);
        FG_TRY(check_launch(ctx, "q7_window_max_kernel"));
        {
            LaunchScope ls(ctx, "q7_flag_kernel");
            hipLaunchKernelGGL(q7_flag_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, bid->price, bid->rows, st,
                               tile_max, d_wmax, flag_words, counts);
        }
        FG_TRY(check_launch(ctx, "q7_flag_kernel"));
    }
    FG_TRY(launch_tile_scan(ctx, counts, st.n_tiles, tile_base, st.tile_first, st.n_seg, d_off));
    FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * ((size_t)n_win + 1), hipMemcpyDeviceToHost, ctx->stream));
    if (n_win > 0) FG_HIP(ctx, hipMemcpyAsync(h_wmax, d_wmax, sizeof(int32_t) * n_win, hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> &offs = ctx->host_i64["q7.win_out_offsets"];
    offs.assign(h_off, h_off + n_win + 1);
    std::vector<int64_t> &wmax = ctx->host_i64["q7.win_max"];
    wmax.assign(h_wmax, h_wmax + n_win);
    const int64_t n_out = offs[n_win];
    int32_t *o_a = nullptr, *o_p = nullptr, *o_b = nullptr;
    int64_t *o_t = nullptr;
    FG_TRY(arena_get_t(ctx, "q7.out_auction", (size_t)n_out + 1, &o_a));
    FG_TRY(arena_get_t(ctx, "q7.out_price", (size_t)n_out + 1, &o_p));
    FG_TRY(arena_get_t(ctx, "q7.out_bidder", (size_t)n_out + 1, &o_b));
    FG_TRY(arena_get_t(ctx, "q7.out_time", (size_t)n_out + 1, &o_t));
    if (n_out > 0) FG_TRY(emit_flagged_bids(ctx, st, flag_words, counts, tile_base, *bid, o_a, o_p, o_b, o_t));
    out->auction = o_a;
    out->price = o_p;
    out->bidder = o_b;
    out->b_date_time = o_t;
    out->win_out_offsets = offs.data();
    out->win_max = wmax.data();
    out->rows = n_out;
    return FLOCKGPU_OK;
}

}  // extern "C"
