// NEXMark q8 for gfx950: per Tumbling(10 s) window
//   (SELECT p_id, name FROM person GROUP BY p_id, name) P  JOIN  (SELECT seller FROM auction GROUP BY seller) A
//   ON p_id = seller  ->  [p_id, name]
// (benchmarks/src/nexmark/query/q8.sql, q8_plan.fmt:1-10, playground/.../nexmark/q8.dag).
//
//   sellers : auctions -> DISTINCT seller as a hash set per window.  75 % of the rows of a tile carry the
//             current hot seller (event.rs:255-259): each wave drops lanes whose key equals the wave's first
//             key, and an already-present key costs one L2 load, no atomic.
//   persons : persons -> DISTINCT (p_id, name): claim a slot keyed by p_id with the row index; a loser of the
//             claim compares its full key (p_id and name bytes) with the winner's and is dropped when equal.
//             Surviving rows probe the seller set; survivors are compacted in row order by the chained scan.
//   gather  : take() of p_id and name.
#include <algorithm>

#include "gather.hpp"
#include "hashtab.hpp"

using namespace flockgpu;

namespace {

constexpr int kSellerIters = 4;
constexpr int kSellerTile = kBlock * 4 * kSellerIters;  // 4096 auctions per workgroup
constexpr int kPersonItems = 8;
constexpr int kPersonTile = kBlock * kPersonItems;      // 2048 persons per workgroup
constexpr int kPersonWaveRows = kPersonTile / kWavesPerBlock;

__global__ __launch_bounds__(kBlock) void q8_sellers_kernel(const int32_t *__restrict__ seller, int64_t n_rows, SegTiles st,
                                                            uint64_t *sets, uint32_t cap, uint32_t *err) {
    const TileRange tr = locate_tile(st, (int32_t)blockIdx.x, kSellerTile);
    uint64_t *set = sets + (size_t)tr.seg * cap;
#pragma unroll
    for (int it = 0; it < kSellerIters; ++it) {
        const int64_t r0 = tr.tile_begin + it * (kBlock * 4) + threadIdx.x * 4;
        int32_t k[4];
        if (r0 + 4 <= n_rows) {
            const int4 t = *reinterpret_cast<const int4 *>(seller + r0);
            k[0] = t.x; k[1] = t.y; k[2] = t.z; k[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) k[j] = (r0 + j < n_rows) ? seller[r0 + j] : 0;
        }
        bool v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (r0 + j >= tr.lo) && (r0 + j < tr.hi);
        // wave collapse of the hot seller: only the first live lane keeps it
        const uint64_t live = __ballot(v[0]);
        if (live) {
            const int src = __ffsll((unsigned long long)live) - 1;
            const int32_t hot = __shfl(k[0], src, 64);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (v[j] && k[j] == hot && !(lane_id() == src && j == 0)) v[j] = false;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j)
                if (v[i] && v[j] && k[i] == k[j]) v[j] = false;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (v[j] && set_insert(set, cap, k[j], (int32_t)(r0 + j)) < 0) atomicOr(err, 1u);
    }
}

__device__ __forceinline__ bool same_person(const int32_t *__restrict__ p_id, const int32_t *__restrict__ name_off,
                                            const uint8_t *__restrict__ name, int64_t a, int64_t b) {
    if (p_id[a] != p_id[b]) return false;
    const int32_t ab = name_off[a], ae = name_off[a + 1], bb = name_off[b], be = name_off[b + 1];
    if (ae - ab != be - bb) return false;
    for (int32_t k = 0; k < ae - ab; ++k)
        if (name[ab + k] != name[bb + k]) return false;
    return true;
}

__global__ __launch_bounds__(kBlock) void q8_persons_kernel(const int32_t *__restrict__ p_id,
                                                            const int32_t *__restrict__ name_off,
                                                            const uint8_t *__restrict__ name, SegTiles st,
                                                            uint32_t *ptabs, uint32_t pcap, const uint64_t *sets,
                                                            uint32_t scap, uint64_t *status,
                                                            int32_t *__restrict__ out_person_row, int64_t *seg_out_off,
                                                            uint32_t *err) {
    __shared__ uint64_t s_scan[2 * kWavesPerBlock];
    StripedScan sc;
#pragma unroll 1
    for (int32_t tile = (int32_t)blockIdx.x; tile < st.n_tiles; tile += (int32_t)gridDim.x) {
    const TileRange tr = locate_tile(st, tile, kPersonTile);
    uint32_t *ptab = ptabs + (size_t)tr.seg * pcap;
    const uint64_t *set = sets + (size_t)tr.seg * scap;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int64_t wbase = tr.tile_begin + (int64_t)wave * kPersonWaveRows + lane;

    uint32_t flags = 0, lane_rank[kPersonItems], it_total[kPersonItems], wave_total = 0;
#pragma unroll
    for (int it = 0; it < kPersonItems; ++it) {
        const int64_t r = wbase + it * 64;
        bool keep = false;
        if (r >= tr.lo && r < tr.hi) {
            const int32_t key = p_id[r];
            // DISTINCT (p_id, name): first claimant of a slot represents its key
            uint32_t s = slot_of((uint32_t)key, pcap);
            bool unique = false, done = false;
#pragma unroll 1
            for (uint32_t probe = 0, lim = probe_limit(pcap); probe < lim && !done; ++probe) {
                uint32_t cur = __hip_atomic_load(&ptab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == kEmpty32) {
                    uint32_t expected = kEmpty32;
                    if (__hip_atomic_compare_exchange_strong(&ptab[s], &expected, (uint32_t)r, __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        unique = true;
                        done = true;
                        break;
                    }
                    cur = expected;
                }
                if (same_person(p_id, name_off, name, r, (int64_t)cur)) {
                    done = true;  // duplicate of an earlier claimant
                    break;
                }
                s = (s + 1 == pcap) ? 0 : s + 1;
            }
            if (!done) atomicOr(err, 1u);
            keep = unique && multimap_find(set, scap, key) >= 0;
        }
        const uint64_t b = __ballot(keep);
        lane_rank[it] = mbcnt(b);
        it_total[it] = (uint32_t)__popcll((unsigned long long)b);
        wave_total += it_total[it];
        flags |= (keep ? 1u : 0u) << it;
    }
    uint64_t tile_base, tile_total;
    uint64_t pos = block_striped_offset(status, sc, tile, wave_total, s_scan, &tile_base, &tile_total, err);
    if (threadIdx.x == 0) {
        if (tile == st.tile_first[tr.seg]) seg_out_off[tr.seg] = (int64_t)tile_base;
        if (tile == st.n_tiles - 1) seg_out_off[st.n_seg] = (int64_t)(tile_base + tile_total);
    }
#pragma unroll
    for (int it = 0; it < kPersonItems; ++it) {
        if (flags & (1u << it)) out_person_row[pos + lane_rank[it]] = (int32_t)(wbase + it * 64);
        pos += it_total[it];
    }
    }  // tile loop
}

}  // namespace

extern "C" {

int flockgpu_q8_join(flockgpu_ctx *ctx, const flockgpu_person_cols *person, const flockgpu_windows *person_win,
                     const flockgpu_auction_cols *auction, const flockgpu_windows *auction_win, flockgpu_q8_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!auction || !person || !out || auction->rows < 0 || person->rows < 0)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q8: null argument");
    FG_TRY(check_windows(ctx, auction_win, auction->rows, "q8.auction"));
    FG_TRY(check_windows(ctx, person_win, person->rows, "q8.person"));
    if (auction_win->n_windows != person_win->n_windows)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q8: auction and person schedules differ in window count");
    if (auction->rows >= (int64_t(1) << 31) || person->rows >= (int64_t(1) << 31) - 1)
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q8: relations are limited to 2^31 rows per call");
    if (auction->rows > 0 && !auction->seller) return fail(ctx, FLOCKGPU_ERR_INVALID, "q8: null seller column");
    if (person->rows > 0 && (!person->p_id || !person->name.offsets || !person->name.data))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q8: null person column");
    if (reinterpret_cast<uintptr_t>(auction->seller) & 15)
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q8: seller column must be 16-byte aligned");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_win = auction_win->n_windows;

    std::vector<int64_t> ab(n_win), ae(n_win), pb(n_win), pe(n_win);
    int64_t max_p = 0, max_a = 0, out_cap = 16;
    for (int w = 0; w < n_win; ++w) {
        ab[w] = auction_win->pane_row_offsets[auction_win->win_pane_lo[w]];
        ae[w] = auction_win->pane_row_offsets[auction_win->win_pane_hi[w]];
        pb[w] = person_win->pane_row_offsets[person_win->win_pane_lo[w]];
        pe[w] = person_win->pane_row_offsets[person_win->win_pane_hi[w]];
        max_p = std::max(max_p, pe[w] - pb[w]);
        max_a = std::max(max_a, ae[w] - ab[w]);
        out_cap += pe[w] - pb[w];
    }
    SegTiles st_a, st_p;
    FG_TRY(build_seg_tiles(ctx, "q8.auction", ab.data(), ae.data(), n_win, kSellerTile, &st_a));
    FG_TRY(build_seg_tiles(ctx, "q8.person", pb.data(), pe.data(), n_win, kPersonTile, &st_p));

    const uint64_t pcap64 = std::max<uint64_t>(64, (uint64_t)max_p * 3 / 2 + 8);
    if (pcap64 >= (uint64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q8: window too large");
    const uint32_t pcap = (uint32_t)pcap64;
    uint64_t *status = nullptr;
    FG_TRY(arena_get_t(ctx, "q8.status", (size_t)st_p.n_tiles + 3, &status));  // + spare, err
    uint32_t *d_err = reinterpret_cast<uint32_t *>(status + st_p.n_tiles + 1);
    int64_t *d_off = nullptr, *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, "q8.seg_out_off", (size_t)n_win + 1, &d_off));
    FG_TRY(pinned_get_t(ctx, "q8.seg_out_off", (size_t)n_win + 2, &h_off));
    uint32_t *ptabs = nullptr;
    FG_TRY(arena_get_t(ctx, "q8.person_tables", (size_t)pcap * std::max(n_win, 1), &ptabs));
    int32_t *o_pr = nullptr;
    FG_TRY(arena_get_t(ctx, "q8.out_person_row", (size_t)out_cap, &o_pr));

    // the seller sets are sized from the distinct-seller density seen last time; a full set redoes the window batch
    uint64_t scap64 = std::max<uint64_t>(64, (uint64_t)((double)max_a / std::max(1.0, ctx->q8_rows_per_seller) * 2.0) + 64);
    int64_t n_out = 0;
    for (int attempt = 0;; ++attempt) {
        if (attempt > 6 || scap64 >= (uint64_t(1) << 31))
            return fail(ctx, FLOCKGPU_ERR_CAPACITY, "q8: seller set capacity %llu still overflows", (unsigned long long)scap64);
        const uint32_t scap = (uint32_t)scap64;
        uint64_t *sets = nullptr;
        FG_TRY(arena_get_t(ctx, "q8.seller_sets", (size_t)scap * std::max(n_win, 1), &sets));
        FG_HIP(ctx, hipMemsetAsync(sets, 0xFF, sizeof(uint64_t) * (size_t)scap * n_win, ctx->stream));
        FG_HIP(ctx, hipMemsetAsync(ptabs, 0xFF, sizeof(uint32_t) * (size_t)pcap * n_win, ctx->stream));
        FG_HIP(ctx, hipMemsetAsync(status, 0, sizeof(uint64_t) * ((size_t)st_p.n_tiles + 3), ctx->stream));
        FG_HIP(ctx, hipMemsetAsync(d_off, 0xFF, sizeof(int64_t) * ((size_t)n_win + 1), ctx->stream));
        if (st_a.n_tiles > 0) {
            LaunchScope ls(ctx, "q8_sellers_kernel");
            hipLaunchKernelGGL(q8_sellers_kernel, dim3((unsigned)st_a.n_tiles), dim3(kBlock), 0, ctx->stream, auction->seller,
                               auction->rows, st_a, sets, scap, d_err);
        }
        FG_TRY(check_launch(ctx, "q8_sellers_kernel"));
        if (st_p.n_tiles > 0) {
            unsigned grid = 1;
            FG_TRY(persistent_grid(ctx, q8_persons_kernel, "q8_persons_kernel", st_p.n_tiles, &grid));
            LaunchScope ls(ctx, "q8_persons_kernel");
            hipLaunchKernelGGL(q8_persons_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, person->p_id,
                               person->name.offsets, person->name.data, st_p, ptabs, pcap, sets, scap, status, o_pr, d_off,
                               d_err);
        }
        FG_TRY(check_launch(ctx, "q8_persons_kernel"));
        FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * ((size_t)n_win + 1), hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipMemcpyAsync(h_off + n_win + 1, d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (const uint32_t h_err = *reinterpret_cast<uint32_t *>(h_off + n_win + 1)) {
            if (h_err & 2u) return fail(ctx, FLOCKGPU_ERR_HIP, "q8: chained scan stalled");
            scap64 *= 4;
            continue;
        }
        n_out = st_p.n_tiles == 0 ? 0 : h_off[n_win];
        if (attempt > 0) ctx->q8_rows_per_seller = std::max(1.0, (double)max_a * 2.0 / (double)scap64);
        break;
    }
    std::vector<int64_t> &offs = ctx->host_i64["q8.win_out_offsets"];
    offs.assign(h_off, h_off + n_win + 1);
    if (st_p.n_tiles == 0) offs[n_win] = 0;
    for (int w = n_win - 1; w >= 0; --w)
        if (offs[w] < 0) offs[w] = offs[w + 1];

    int32_t *o_pid = nullptr;
    FG_TRY(arena_get_t(ctx, "q8.out_p_id", (size_t)n_out + 1, &o_pid));
    FG_TRY(gather_i32(ctx, person->p_id, o_pr, n_out, o_pid));
    FG_TRY(gather_utf8(ctx, "q8.out_name", person->name, o_pr, n_out, &out->name, &out->name_bytes));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->p_id = o_pid;
    out->person_row = o_pr;
    out->win_out_offsets = offs.data();
    out->rows = n_out;
    return FLOCKGPU_OK;
}

}  // extern "C"
