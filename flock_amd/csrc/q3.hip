// NEXMark q3 for gfx950: per ElementWise window (1-s epoch)
//   Filter(auction.category = 10)  JOIN  Filter(person.state = 'or' OR 'id' OR 'ca')  ON seller = p_id
//   -> Projection [name, city, state, a_id]
// (benchmarks/src/nexmark/query/q3.sql, q3_plan.fmt:1-6, flock/src/distributed_plan/planner.rs:152-171).
//
// All windows of a schedule run in THREE launches (build, probe, gathers), not three per epoch: a tile never
// straddles a window and every window owns a region of one global hash table, so per-epoch work of a few
// hundred KB does not become launch-latency bound (SURVEY.md section 7 "hard parts").
//   build : persons -> state filter (byte compare on the Utf8 buffers) -> multimap insert keyed p_id
//   probe : auctions -> category filter -> lookup seller -> order-preserving expansion of the matching
//           (auction_row, person_row) pairs via the single-pass chained scan
//   gather: take() of a_id and of the three Utf8 columns
// DataFusion builds on the LEFT (auction) side; which side is hashed is unobservable in the result multiset,
// so the smaller, key-unique side is built here while duplicates on either side still produce every pair.
#include <algorithm>

#include "gather.hpp"
#include "hashtab.hpp"

using namespace flockgpu;

namespace {

constexpr int kMaxLits = 8;
struct Utf8Lits {  // literals of the `state = lit OR ...` chain, each <= 8 bytes
    uint64_t bytes[kMaxLits];
    uint32_t len[kMaxLits];
    int32_t n;
};

constexpr int kBuildItems = 8;
constexpr int kBuildTile = kBlock * kBuildItems;  // 2048 persons per workgroup
constexpr int kProbeIters = 2;
constexpr int kProbeTile = kBlock * 4 * kProbeIters;  // 2048 auctions per workgroup
constexpr int kProbeWaveRows = kProbeTile / kWavesPerBlock;

__device__ __forceinline__ bool utf8_in(const int32_t *__restrict__ off, const uint8_t *__restrict__ data, int64_t row,
                                        const Utf8Lits &lits) {
    const int32_t b = off[row], e = off[row + 1];
    const uint32_t len = (uint32_t)(e - b);
    if (len > 8) return false;
    uint64_t v = 0;
    for (uint32_t k = 0; k < len; ++k) v |= (uint64_t)data[b + k] << (8 * k);
    bool hit = false;
#pragma unroll
    for (int l = 0; l < kMaxLits; ++l) hit = hit || (l < lits.n && lits.len[l] == len && lits.bytes[l] == v);
    return hit;
}

__global__ __launch_bounds__(kBlock) void q3_build_kernel(const int32_t *__restrict__ p_id,
                                                          const int32_t *__restrict__ state_off,
                                                          const uint8_t *__restrict__ state_data, SegTiles st, Utf8Lits lits,
                                                          uint64_t *tables, uint32_t cap, int32_t *next, uint32_t *err) {
    const TileRange tr = locate_tile(st, (int32_t)blockIdx.x, kBuildTile);
    uint64_t *tab = tables + (size_t)tr.seg * cap;
#pragma unroll
    for (int it = 0; it < kBuildItems; ++it) {
        const int64_t r = tr.tile_begin + it * kBlock + threadIdx.x;
        if (r < tr.lo || r >= tr.hi) continue;
        if (!utf8_in(state_off, state_data, r, lits)) continue;
        if (!multimap_insert(tab, cap, next, p_id[r], (int32_t)r)) atomicOr(err, 1u);
    }
}

__global__ __launch_bounds__(kBlock) void q3_probe_kernel(const int32_t *__restrict__ seller,
                                                          const int32_t *__restrict__ category, int64_t n_rows,
                                                          int64_t category_lit, SegTiles st, const uint64_t *tables,
                                                          uint32_t cap, const int32_t *__restrict__ next, uint64_t *status,
                                                          uint32_t *err, int32_t *__restrict__ out_auction_row,
                                                          int32_t *__restrict__ out_person_row, uint64_t out_cap,
                                                          int64_t *seg_out_off) {
    __shared__ uint64_t s_scan[2 * kWavesPerBlock];
    StripedScan sc;
    const int wave = threadIdx.x >> 6, lane = lane_id();
#pragma unroll 1
    for (int32_t tile = (int32_t)blockIdx.x; tile < st.n_tiles; tile += (int32_t)gridDim.x) {
    const TileRange tr = locate_tile(st, tile, kProbeTile);
    const uint64_t *tab = tables + (size_t)tr.seg * cap;
    const int64_t wbase = tr.tile_begin + (int64_t)wave * kProbeWaveRows + lane * 4;

    int32_t head[kProbeIters][4];
    uint32_t cnt[kProbeIters][4];
    uint32_t lane_rank[kProbeIters], it_total[kProbeIters], wave_total = 0;
#pragma unroll
    for (int it = 0; it < kProbeIters; ++it) {
        const int64_t r0 = wbase + it * 256;
        int32_t s4[4], c4[4];
        if (r0 + 4 <= n_rows) {
            const int4 a = *reinterpret_cast<const int4 *>(seller + r0);
            const int4 b = *reinterpret_cast<const int4 *>(category + r0);
            s4[0] = a.x; s4[1] = a.y; s4[2] = a.z; s4[3] = a.w;
            c4[0] = b.x; c4[1] = b.y; c4[2] = b.z; c4[3] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s4[j] = (r0 + j < n_rows) ? seller[r0 + j] : 0;
                c4[j] = (r0 + j < n_rows) ? category[r0 + j] : 0;
            }
        }
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + j;
            head[it][j] = -1;
            cnt[it][j] = 0;
            if (r >= tr.lo && r < tr.hi && (int64_t)c4[j] == category_lit) {
                const int32_t h = multimap_find(tab, cap, s4[j]);
                head[it][j] = h;
                uint32_t n = 0;
                for (int32_t p = h; p >= 0; p = next[p]) ++n;
                cnt[it][j] = n;
                mine += n;
            }
        }
        const uint32_t incl = wave_incl_scan_u32(mine);
        lane_rank[it] = incl - mine;
        it_total[it] = __shfl(incl, 63, 64);
        wave_total += it_total[it];
    }
    uint64_t tile_base, tile_total;
    uint64_t pos = block_striped_offset(status, sc, tile, wave_total, s_scan, &tile_base, &tile_total, err);
    if (threadIdx.x == 0) {
        if (tile == st.tile_first[tr.seg]) seg_out_off[tr.seg] = (int64_t)tile_base;
        if (tile == st.n_tiles - 1) seg_out_off[st.n_seg] = (int64_t)(tile_base + tile_total);
    }
    if (wave_total == 0) continue;
#pragma unroll
    for (int it = 0; it < kProbeIters; ++it) {
        uint64_t p = pos + lane_rank[it];
        const int64_t r0 = wbase + it * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            for (int32_t q = head[it][j]; q >= 0; q = next[q]) {
                if (p < out_cap) {  // a too-small pair buffer is detected by the host from the scan total
                    out_auction_row[p] = (int32_t)(r0 + j);
                    out_person_row[p] = q;
                }
                ++p;
            }
        }
        pos += it_total[it];
    }
    }  // tile loop
}

}  // namespace

extern "C" {

int flockgpu_q3_join(flockgpu_ctx *ctx, const flockgpu_auction_cols *auction, const flockgpu_windows *auction_win,
                     const flockgpu_person_cols *person, const flockgpu_windows *person_win, int64_t category_lit,
                     const char *const *state_lits, int n_state_lits, flockgpu_q3_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!auction || !person || !out || auction->rows < 0 || person->rows < 0)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q3: null argument");
    FG_TRY(check_windows(ctx, auction_win, auction->rows, "q3.auction"));
    FG_TRY(check_windows(ctx, person_win, person->rows, "q3.person"));
    if (auction_win->n_windows != person_win->n_windows)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q3: auction and person schedules differ in window count");
    if (auction->rows >= (int64_t(1) << 31) || person->rows >= (int64_t(1) << 31))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q3: relations are limited to 2^31 rows per call");
    if (auction->rows > 0 && (!auction->a_id || !auction->seller || !auction->category))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q3: null auction column");
    if (person->rows > 0 && (!person->p_id || !person->state.offsets || !person->name.offsets || !person->city.offsets))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "q3: null person column");
    if ((reinterpret_cast<uintptr_t>(auction->seller) & 15) || (reinterpret_cast<uintptr_t>(auction->category) & 15))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q3: auction columns must be 16-byte aligned");
    if (n_state_lits < 0 || n_state_lits > kMaxLits) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q3: more than 8 literals");
    Utf8Lits lits{};
    lits.n = n_state_lits;
    for (int l = 0; l < n_state_lits; ++l) {
        const size_t len = std::strlen(state_lits[l]);
        if (len > 8) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q3: Utf8 literal longer than 8 bytes");
        lits.len[l] = (uint32_t)len;
        for (size_t k = 0; k < len; ++k) lits.bytes[l] |= (uint64_t)(uint8_t)state_lits[l][k] << (8 * k);
    }
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_win = auction_win->n_windows;

    std::vector<int64_t> ab(n_win), ae(n_win), pb(n_win), pe(n_win);
    int64_t max_person_rows = 0;
    for (int w = 0; w < n_win; ++w) {
        ab[w] = auction_win->pane_row_offsets[auction_win->win_pane_lo[w]];
        ae[w] = auction_win->pane_row_offsets[auction_win->win_pane_hi[w]];
        pb[w] = person_win->pane_row_offsets[person_win->win_pane_lo[w]];
        pe[w] = person_win->pane_row_offsets[person_win->win_pane_hi[w]];
        max_person_rows = std::max(max_person_rows, pe[w] - pb[w]);
    }
    SegTiles st_a, st_p;
    FG_TRY(build_seg_tiles(ctx, "q3.auction", ab.data(), ae.data(), n_win, kProbeTile, &st_a));
    FG_TRY(build_seg_tiles(ctx, "q3.person", pb.data(), pe.data(), n_win, kBuildTile, &st_p));

    const uint64_t cap64 = std::max<uint64_t>(64, (uint64_t)max_person_rows * 3 / 2 + 8);
    if (cap64 >= (uint64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q3: window too large for one table region");
    const uint32_t cap = (uint32_t)cap64;
    uint64_t *tables = nullptr;
    int32_t *next = nullptr;
    FG_TRY(arena_get_t(ctx, "q3.tables", (size_t)cap * std::max(n_win, 1), &tables));
    FG_TRY(arena_get_t(ctx, "q3.next", (size_t)person->rows + 1, &next));
    uint64_t *status = nullptr;
    FG_TRY(arena_get_t(ctx, "q3.status", (size_t)st_a.n_tiles + 4, &status));  // + spare, err, pair total
    int64_t *d_off = nullptr, *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, "q3.seg_out_off", (size_t)n_win + 1, &d_off));
    FG_TRY(pinned_get_t(ctx, "q3.seg_out_off", (size_t)n_win + 2, &h_off));
    FG_HIP(ctx, hipMemsetAsync(tables, 0xFF, sizeof(uint64_t) * (size_t)cap * n_win, ctx->stream));
    FG_HIP(ctx, hipMemsetAsync(status, 0, sizeof(uint64_t) * ((size_t)st_a.n_tiles + 4), ctx->stream));
    FG_HIP(ctx, hipMemsetAsync(d_off, 0xFF, sizeof(int64_t) * ((size_t)n_win + 1), ctx->stream));
    uint32_t *d_err = reinterpret_cast<uint32_t *>(status + st_a.n_tiles + 1);

    if (st_p.n_tiles > 0) {
        LaunchScope ls(ctx, "q3_build_kernel");
        hipLaunchKernelGGL(q3_build_kernel, dim3((unsigned)st_p.n_tiles), dim3(kBlock), 0, ctx->stream, person->p_id,
                           person->state.offsets, person->state.data, st_p, lits, tables, cap, next, d_err);
    }
    FG_TRY(check_launch(ctx, "q3_build_kernel"));
    // The pair buffers are sized optimistically (one match per auction row: p_id is unique in NEXMark); the
    // chained scan's grand total tells the host when a hot build key needed more, and the probe is redone.
    uint64_t out_cap = 16;
    for (int w = 0; w < n_win; ++w) out_cap += (uint64_t)(ae[w] - ab[w]);
    int32_t *o_ar = nullptr, *o_pr = nullptr, *o_aid = nullptr;
    uint64_t n_pairs = 0;
    for (int attempt = 0;; ++attempt) {
        FG_TRY(arena_get_t(ctx, "q3.out_auction_row", (size_t)out_cap, &o_ar));
        FG_TRY(arena_get_t(ctx, "q3.out_person_row", (size_t)out_cap, &o_pr));
        if (attempt > 0) {
            FG_HIP(ctx, hipMemsetAsync(status, 0, sizeof(uint64_t) * ((size_t)st_a.n_tiles + 1), ctx->stream));
            FG_HIP(ctx, hipMemsetAsync(d_off, 0xFF, sizeof(int64_t) * ((size_t)n_win + 1), ctx->stream));
        }
        if (st_a.n_tiles > 0) {
            unsigned grid = 1;
            FG_TRY(persistent_grid(ctx, q3_probe_kernel, "q3_probe_kernel", st_a.n_tiles, &grid));
            LaunchScope ls(ctx, "q3_probe_kernel");
            hipLaunchKernelGGL(q3_probe_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, auction->seller,
                               auction->category, auction->rows, category_lit, st_a, tables, cap, next, status, d_err, o_ar,
                               o_pr, out_cap, d_off);
        }
        FG_TRY(check_launch(ctx, "q3_probe_kernel"));
        FG_HIP(ctx, hipMemcpyAsync(h_off, d_off, sizeof(int64_t) * ((size_t)n_win + 1), hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipMemcpyAsync(h_off + n_win + 1, d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (const uint32_t h_err = *reinterpret_cast<uint32_t *>(h_off + n_win + 1)) {
            if (h_err & 2u) return fail(ctx, FLOCKGPU_ERR_HIP, "q3: chained scan stalled");
            return fail(ctx, FLOCKGPU_ERR_CAPACITY, "q3: build table overflow (cap %u)", cap);
        }
        n_pairs = st_a.n_tiles == 0 ? 0 : (uint64_t)h_off[n_win];
        if (n_pairs >= (uint64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q3: join output exceeds 2^31 rows");
        if (n_pairs <= out_cap) break;
        if (attempt > 0) return fail(ctx, FLOCKGPU_ERR_HIP, "q3: pair count changed between probes");
        out_cap = n_pairs + 16;
    }
    FG_TRY(arena_get_t(ctx, "q3.out_a_id", (size_t)n_pairs + 1, &o_aid));
    FG_TRY(gather_i32(ctx, auction->a_id, o_ar, (int64_t)n_pairs, o_aid));
    FG_TRY(gather_utf8(ctx, "q3.out_name", person->name, o_pr, (int64_t)n_pairs, &out->name, &out->name_bytes));
    FG_TRY(gather_utf8(ctx, "q3.out_city", person->city, o_pr, (int64_t)n_pairs, &out->city, &out->city_bytes));
    FG_TRY(gather_utf8(ctx, "q3.out_state", person->state, o_pr, (int64_t)n_pairs, &out->state, &out->state_bytes));
    std::vector<int64_t> &offs = ctx->host_i64["q3.win_out_offsets"];
    offs.assign(h_off, h_off + n_win + 1);  // (read before the gathers reuse nothing of h_off)
    if (st_a.n_tiles == 0) offs[n_win] = 0;
    for (int w = n_win - 1; w >= 0; --w)
        if (offs[w] < 0) offs[w] = offs[w + 1];
    out->a_id = o_aid;
    out->auction_row = o_ar;
    out->person_row = o_pr;
    out->win_out_offsets = offs.data();
    out->rows = (int64_t)n_pairs;
    return FLOCKGPU_OK;
}

}  // extern "C"
