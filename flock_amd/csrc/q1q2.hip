// NEXMark q1 (projection) and q2 (filter -> projection) for gfx950.
//   q1: ProjectionExec [auction, bidder, 0.908 * CAST(price AS Float64), b_date_time]   (planner.rs:90)
//   q2: FilterExec CAST(auction AS Int64) % 123 = 0 -> CoalesceBatches -> Projection     (planner.rs:120-124)
// Both are HBM-bound integer/byte scans: 16-byte coalesced lane loads, wave64 ballot + mbcnt ranks,
// single-pass chained scan for the stable (input-order) compaction.  No MFMA, no LDS staging needed.
#include <algorithm>

#include "scan.hpp"

using namespace flockgpu;

namespace {

constexpr int kQ2Iters = 8;                      // 8 x (4 rows per lane) = 32 rows per thread
constexpr int kQ2Tile = kBlock * 4 * kQ2Iters;   // 8192 rows per tile
constexpr int kQ2WaveRows = kQ2Tile / kWavesPerBlock;

// `CAST(a AS Int64) % m == 0` (truncated remainder; the plan compares with the literal 0, planner.rs:122)
// without a divide: a is divisible by m  <=>  |a| is divisible by d = |m| = 2^k * o (o odd)
// <=>  ror32(|a| * inverse(o) mod 2^32, k) <= floor((2^32 - 1) / d)   (Granlund-Montgomery / Lemire-Kaser).
// One 32-bit multiply per row instead of the 128-bit product a fastmod remainder needs.
struct ModPred {
    uint32_t inv;    // o^-1 mod 2^32
    uint32_t lim;    // floor((2^32 - 1) / d)
    uint32_t shift;  // k
    int32_t wide;    // |m| >= 2^32  ->  a % m == a, zero only for a == 0
};

__device__ __forceinline__ bool mod_is_zero(int32_t a, const ModPred &p) {
    const uint32_t n = a < 0 ? 0u - (uint32_t)a : (uint32_t)a;
    if (p.wide) return n == 0;
    const uint32_t q = n * p.inv;
    return __funnelshift_r(q, q, p.shift) <= p.lim;
}

__device__ __forceinline__ void load4(const int32_t *__restrict__ col, int64_t r0, int64_t n_rows, int32_t (&v)[4]) {
    if (r0 >= 0 && r0 + 4 <= n_rows) {
        const int4 t = *reinterpret_cast<const int4 *>(col + r0);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (r0 + j >= 0 && r0 + j < n_rows) ? col[r0 + j] : 0;
    }
}

// Loads the tile's `auction` values: lane l of wave w holds rows  w*2048 + it*256 + 4l .. +3  (relative to
// tile_begin), i.e. every load instruction of a wave covers 1 KiB of consecutive bytes.
__device__ __forceinline__ void q2_load_tile(const int32_t *__restrict__ auction, int64_t n_rows, const TileRange &tr,
                                             int32_t (&a)[kQ2Iters][4]) {
    const int64_t wbase = tr.tile_begin + (int64_t)(threadIdx.x >> 6) * kQ2WaveRows + lane_id() * 4;
    if (tr.tile_begin + kQ2Tile <= n_rows) {  // block-uniform: no row of the tile is past the column
#pragma unroll
        for (int it = 0; it < kQ2Iters; ++it) {
            const int4 t = stream_load4(auction + wbase + it * 256);
            a[it][0] = t.x; a[it][1] = t.y; a[it][2] = t.z; a[it][3] = t.w;
        }
    } else {
#pragma unroll
        for (int it = 0; it < kQ2Iters; ++it) load4(auction, wbase + it * 256, n_rows, a[it]);
    }
}

// q2 runs as count -> scan -> emit (scan.hpp):
//   q2_flag_kernel : streams `auction` once (the only pass over the column), evaluates the predicate and leaves
//                    ONE 32-bit word of row flags per lane (bit it*4+j, 1 KiB per 8192-row tile, coalesced) plus one
//                    count per wave.  No LDS, no barrier, no dependence between workgroups: a pure HBM stream.
//   tile_scan      : tile bases + per-window output offsets.
//   q2_emit_kernel : per tile with survivors: flag words -> ranks -> the survivors' row numbers into an LDS list
//                    in row order -> the block copies (auction, price) of the listed rows, lane i taking survivor
//                    i, so the gathers are issued back to back and the stores are coalesced.  `price` is only
//                    ever touched for surviving rows (their `auction` values are L2 / MALL hits).
__global__ __launch_bounds__(kBlock) void q2_flag_kernel(const int32_t *__restrict__ auction, int64_t n_rows, SegTiles st,
                                                         ModPred pred, uint32_t *__restrict__ flag_words,
                                                         uint32_t *__restrict__ counts) {
    int32_t tile = (int32_t)blockIdx.x;
    if (tile >= st.n_tiles) return;
    TileRange tr = locate_tile(st, tile, kQ2Tile);
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int32_t rel0 = wave * kQ2WaveRows + lane * 4;
#pragma unroll 1
    for (;;) {  // tiles b, b + G, ...: the next descriptor is requested while this tile's rows are in flight (scan.hpp)
        int32_t a[kQ2Iters][4];
        q2_load_tile(auction, n_rows, tr, a);
        const int32_t next = tile + (int32_t)gridDim.x;
        TileRange trn = tr;
        if (next < st.n_tiles) trn = locate_tile(st, next, kQ2Tile);
        // rows of the tile that belong to the window, relative to tile_begin: [rel_lo, rel_hi)
        const int32_t rel_lo = (int32_t)(tr.lo - tr.tile_begin), rel_hi = (int32_t)(tr.hi - tr.tile_begin);
        uint32_t flags = 0;
#pragma unroll
        for (int it = 0; it < kQ2Iters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int32_t rel = rel0 + it * 256 + j;
                const bool f = mod_is_zero(a[it][j], pred) && rel >= rel_lo && rel < rel_hi;
                flags |= (f ? 1u : 0u) << (it * 4 + j);
            }
        flag_words[(size_t)tile * kBlock + threadIdx.x] = flags;
        const uint32_t incl = wave_incl_scan_u32((uint32_t)__popc(flags));
        if (lane == 63) counts[(size_t)tile * kWavesPerBlock + wave] = incl;
        if (next >= st.n_tiles) break;
        tile = next;
        tr = trn;
    }
}

__global__ __launch_bounds__(kBlock) void q2_emit_kernel(const int32_t *__restrict__ auction,
                                                         const int32_t *__restrict__ price, SegTiles st,
                                                         const uint32_t *__restrict__ flag_words,
                                                         const uint32_t *__restrict__ counts,
                                                         const uint64_t *__restrict__ tile_base,
                                                         int32_t *__restrict__ out_auction,
                                                         int32_t *__restrict__ out_price) {
    __shared__ uint16_t s_list[kQ2Tile];
    const int32_t tile = (int32_t)blockIdx.x;
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)tile * kWavesPerBlock);
    const uint32_t total = wc.x + wc.y + wc.z + wc.w;
    if (total == 0) return;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t flags = flag_words[(size_t)tile * kBlock + threadIdx.x];
    uint32_t p = (wave > 0 ? wc.x : 0u) + (wave > 1 ? wc.y : 0u) + (wave > 2 ? wc.z : 0u);
    const int32_t rel0 = wave * kQ2WaveRows + lane * 4;
#pragma unroll
    for (int it = 0; it < kQ2Iters; ++it) {
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
    __syncthreads();
    const TileRange tr = locate_tile(st, tile, kQ2Tile);
    const uint64_t base = tile_base[tile];
    for (uint32_t i = threadIdx.x; i < total; i += kBlock) {
        const int64_t r = tr.tile_begin + s_list[i];
        out_auction[base + i] = auction[r];
        out_price[base + i] = price[r];
    }
}

// q1: one IEEE-754 f64 multiply per row (i32 -> f64 cast is exact).  8 rows per thread per step.
__global__ __launch_bounds__(kBlock) void q1_project_kernel(const int32_t *__restrict__ price, int64_t n_rows,
                                                            double factor, double *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * kBlock * 4;
    for (int64_t r0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4; r0 < n_rows; r0 += stride) {
        if (r0 + 4 <= n_rows) {
            const int4 t = *reinterpret_cast<const int4 *>(price + r0);
            double2 lo, hi;
            lo.x = factor * (double)t.x; lo.y = factor * (double)t.y;
            hi.x = factor * (double)t.z; hi.y = factor * (double)t.w;
            *reinterpret_cast<double2 *>(out + r0) = lo;
            *reinterpret_cast<double2 *>(out + r0 + 2) = hi;
        } else {
            for (int64_t r = r0; r < n_rows; ++r) out[r] = factor * (double)price[r];
        }
    }
}

}  // namespace

extern "C" {

int flockgpu_q1_project(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, double factor, double *out_price) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!bid || bid->rows < 0 || (bid->rows > 0 && (!bid->price || !out_price)))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q1: null column");
    if (bid->rows == 0) return FLOCKGPU_OK;
    if ((reinterpret_cast<uintptr_t>(bid->price) & 15) || (reinterpret_cast<uintptr_t>(out_price) & 15))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q1: columns must be 16-byte aligned");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    int64_t blocks = div_up(bid->rows, (int64_t)kBlock * 4);
    const int64_t cap = (int64_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    {
        LaunchScope ls(ctx, "q1_project_kernel");
        hipLaunchKernelGGL(q1_project_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, bid->price, bid->rows,
                           factor, out_price);
    }
    return check_launch(ctx, "q1_project_kernel");
}

int flockgpu_q2_filter(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win, int64_t modulus,
                       flockgpu_q2_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!bid || !out || bid->rows < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "q2: null argument");
    if (modulus == 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "q2: modulus 0 (DataFusion raises divide-by-zero)");
    FG_TRY(check_windows(ctx, win, bid->rows, "q2"));
    if (bid->rows > 0 && (!bid->auction || !bid->price)) return fail(ctx, FLOCKGPU_ERR_INVALID, "q2: null column");
    if (reinterpret_cast<uintptr_t>(bid->auction) & 15)
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q2: auction column must be 16-byte aligned");
    FG_HIP(ctx, hipSetDevice(ctx->device));

    const int n_win = win->n_windows;
    std::vector<int64_t> sb(n_win), se(n_win);
    int64_t worst = 0;
    for (int w = 0; w < n_win; ++w) {
        sb[w] = win->pane_row_offsets[win->win_pane_lo[w]];
        se[w] = win->pane_row_offsets[win->win_pane_hi[w]];
        worst += se[w] - sb[w];
    }
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, "q2", sb.data(), se.data(), n_win, kQ2Tile, &st));

    int32_t *o_a = nullptr, *o_p = nullptr;
    FG_TRY(arena_get_t(ctx, "q2.out_auction", (size_t)worst, &o_a));
    FG_TRY(arena_get_t(ctx, "q2.out_price", (size_t)worst, &o_p));
    uint32_t *flag_words = nullptr, *counts = nullptr;
    uint64_t *tile_base = nullptr;
    FG_TRY(arena_get_t(ctx, "q2.flag_words", (size_t)st.n_tiles * kBlock, &flag_words));
    FG_TRY(arena_get_t(ctx, "q2.counts", (size_t)st.n_tiles * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, "q2.tile_base", (size_t)st.n_tiles + 1, &tile_base));
    int64_t *d_off = nullptr, *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, "q2.seg_out_off", (size_t)n_win + 1, &d_off));
    FG_TRY(pinned_get_t(ctx, "q2.seg_out_off", (size_t)n_win + 1, &h_off));

    ModPred pred{};
    const uint64_t am = modulus < 0 ? (uint64_t)0 - (uint64_t)modulus : (uint64_t)modulus;
    if (am >> 32) {
        pred.wide = 1;
    } else {
        uint32_t d = (uint32_t)am, k = 0;
        while (!(d & 1u)) { d >>= 1; ++k; }
        uint32_t inv = d;  // Newton: 3 correct bits, doubled per step
        for (int i = 0; i < 5; ++i) inv *= 2u - d * inv;
        pred.inv = inv;
        pred.shift = k;
        pred.lim = 0xFFFFFFFFu / (uint32_t)am;
    }
    if (st.n_tiles > 0) {
        LaunchScope ls(ctx, "q2_flag_kernel");
        const unsigned grid = (unsigned)std::min<int64_t>(st.n_tiles, (int64_t)ctx->num_cus * kStreamBlocksPerCu);
        hipLaunchKernelGGL(q2_flag_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, bid->auction, bid->rows, st, pred,
                           flag_words, counts);
    }
    FG_TRY(check_launch(ctx, "q2_flag_kernel"));
    FG_TRY(launch_tile_scan(ctx, counts, st.n_tiles, tile_base, st.tile_first, st.n_seg, d_off));
    if (st.n_tiles > 0) {
        LaunchScope ls(ctx, "q2_emit_kernel");
        hipLaunchKernelGGL(q2_emit_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, bid->auction,
                           bid->price, st, flag_words, counts, tile_base, o_a, o_p);
    }
    FG_TRY(check_launch(ctx, "q2_emit_kernel"));
    FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * ((size_t)n_win + 1), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> &offs = ctx->host_i64["q2.win_out_offsets"];
    offs.assign(h_off, h_off + n_win + 1);
    out->auction = o_a;
    out->price = o_p;
    out->win_out_offsets = offs.data();
    out->rows = offs[n_win];
    return FLOCKGPU_OK;
}

}  // extern "C"
