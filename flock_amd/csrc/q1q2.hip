// NEXMark q1 (projection) and q2 (filter -> projection) for gfx950.
//   q1: ProjectionExec [auction, bidder, 0.908 * CAST(price AS Float64), b_date_time]   (planner.rs:90)
//   q2: FilterExec CAST(auction AS Int64) % 123 = 0 -> CoalesceBatches -> Projection     (planner.rs:120-124)
// Both are HBM-bound integer/byte scans: 16-byte coalesced lane loads, wave64 ballot + mbcnt ranks,
// single-pass chained scan for the stable (input-order) compaction.  No MFMA, no LDS staging needed.
#include "scan.hpp"

using namespace flockgpu;

namespace {

constexpr int kQ2Iters = 4;                      // 4 x (4 rows per lane) = 16 rows per thread
constexpr int kQ2Tile = kBlock * 4 * kQ2Iters;   // 4096 rows per workgroup
constexpr int kQ2WaveRows = kQ2Tile / kWavesPerBlock;

// Truncated remainder `CAST(a AS Int64) % m == rem` without a hardware divide:
// Lemire's fastmod on |a| (32 bit) with a 64-bit magic, sign restored afterwards.
struct ModPred {
    uint64_t magic;  // floor((2^64 - 1) / d) + 1
    uint32_t d;      // |m| when it fits 32 bits
    int32_t rem;
    int32_t wide;    // |m| >= 2^32  ->  a % m == a
    int32_t never;   // rem does not fit Int32 / has impossible sign  ->  always false
};

__device__ __forceinline__ bool mod_eq(int32_t a, const ModPred &p) {
    if (p.wide) return a == p.rem;
    const uint32_t n = a < 0 ? (uint32_t)(-(int64_t)a) : (uint32_t)a;
    const uint64_t low = p.magic * (uint64_t)n;
    const uint32_t r = (uint32_t)__umul64hi(low, (uint64_t)p.d);
    const int32_t sr = a < 0 ? -(int32_t)r : (int32_t)r;
    return sr == p.rem;
}

__device__ __forceinline__ void load4(const int32_t *__restrict__ col, int64_t r0, int64_t n_rows, int32_t (&v)[4]) {
    if (r0 + 4 <= n_rows) {
        const int4 t = *reinterpret_cast<const int4 *>(col + r0);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (r0 + j < n_rows) ? col[r0 + j] : 0;
    }
}

__global__ __launch_bounds__(kBlock) void q2_filter_kernel(const int32_t *__restrict__ auction,
                                                           const int32_t *__restrict__ price, int64_t n_rows,
                                                           SegTiles st, ModPred pred, uint64_t *status,
                                                           uint32_t *ticket, int32_t *__restrict__ out_auction,
                                                           int32_t *__restrict__ out_price, int64_t *seg_out_off) {
    __shared__ uint64_t s_scan[kWavesPerBlock + 1];
    __shared__ int32_t s_tile;
    const int32_t tile = take_ticket(ticket, &s_tile);
    const TileRange tr = locate_tile(st, tile, kQ2Tile);
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int64_t wbase = tr.tile_begin + (int64_t)wave * kQ2WaveRows + lane * 4;

    int32_t a[kQ2Iters][4];
#pragma unroll
    for (int it = 0; it < kQ2Iters; ++it) load4(auction, wbase + it * 256, n_rows, a[it]);

    uint32_t flags = 0;               // bit (it*4 + j)
    uint32_t lane_rank[kQ2Iters];     // selected rows of this iteration in lower lanes
    uint32_t it_total[kQ2Iters];
    uint32_t wave_total = 0;
#pragma unroll
    for (int it = 0; it < kQ2Iters; ++it) {
        const int64_t r0 = wbase + it * 256;
        uint32_t rank = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + j;
            const bool f = r >= tr.lo && r < tr.hi && !pred.never && mod_eq(a[it][j], pred);
            const uint64_t b = __ballot(f);
            rank += mbcnt(b);
            total += (uint32_t)__popcll((unsigned long long)b);
            flags |= (f ? 1u : 0u) << (it * 4 + j);
        }
        lane_rank[it] = rank;
        it_total[it] = total;
        wave_total += total;
    }

    uint64_t tile_base, tile_total;
    uint64_t pos = block_chained_offset(status, tile, wave_total, s_scan, &tile_base, &tile_total);
    if (threadIdx.x == 0) {
        if (tile == st.tile_first[tr.seg]) seg_out_off[tr.seg] = (int64_t)tile_base;
        if (tile == st.n_tiles - 1) seg_out_off[st.n_seg] = (int64_t)(tile_base + tile_total);
    }
    if (wave_total == 0) return;
#pragma unroll
    for (int it = 0; it < kQ2Iters; ++it) {
        uint64_t p = pos + lane_rank[it];
        const int64_t r0 = wbase + it * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (flags & (1u << (it * 4 + j))) {
                out_auction[p] = a[it][j];
                out_price[p] = price[r0 + j];   // price is only touched for surviving rows
                ++p;
            }
        }
        pos += it_total[it];
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
    uint64_t *status = nullptr;
    FG_TRY(arena_get_t(ctx, "q2.status", (size_t)st.n_tiles + 2, &status));  // [n_tiles] doubles as the ticket
    int64_t *d_off = nullptr, *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, "q2.seg_out_off", (size_t)n_win + 1, &d_off));
    FG_TRY(pinned_get_t(ctx, "q2.seg_out_off", (size_t)n_win + 1, &h_off));
    FG_HIP(ctx, hipMemsetAsync(status, 0, sizeof(uint64_t) * ((size_t)st.n_tiles + 2), ctx->stream));
    FG_HIP(ctx, hipMemsetAsync(d_off, 0xFF, sizeof(int64_t) * ((size_t)n_win + 1), ctx->stream));

    ModPred pred{};
    const uint64_t am = modulus < 0 ? (uint64_t)0 - (uint64_t)modulus : (uint64_t)modulus;
    pred.rem = 0;  // the plan compares with literal 0 (planner.rs:122)
    if (am >> 32) {
        pred.wide = 1;
    } else {
        pred.d = (uint32_t)am;
        pred.magic = 0xFFFFFFFFFFFFFFFFull / am + 1;
    }
    if (st.n_tiles > 0) {
        LaunchScope ls(ctx, "q2_filter_kernel");
        hipLaunchKernelGGL(q2_filter_kernel, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, bid->auction,
                           bid->price, bid->rows, st, pred, status, reinterpret_cast<uint32_t *>(status + st.n_tiles),
                           o_a, o_p, d_off);
    }
    FG_TRY(check_launch(ctx, "q2_filter_kernel"));
    FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * ((size_t)n_win + 1), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> &offs = ctx->host_i64["q2.win_out_offsets"];
    offs.assign(h_off, h_off + n_win + 1);
    if (st.n_tiles == 0) offs[n_win] = 0;
    for (int w = n_win - 1; w >= 0; --w)
        if (offs[w] < 0) offs[w] = offs[w + 1];  // empty windows have no tile to write their offset
    out->auction = o_a;
    out->price = o_p;
    out->win_out_offsets = offs.data();
    out->rows = offs[n_win];
    return FLOCKGPU_OK;
}

}  // extern "C"
