// NEXMark q5 "hot items" for gfx950: per hopping window
//   COUNT(*) GROUP BY auction  ->  MAX(num)  ->  rows with num = maxn (ties kept)
// (benchmarks/src/nexmark/query/q5.sql, q5_plan.fmt:1-13, playground/.../nexmark/q5.dag).
//
// HBM-bound integer work, no MFMA.  One pass over the `auction` column (4 B / bid):
//   * rows are cut into 8192-row tiles that never straddle a pane (pane = gcd(window, hop) seconds);
//   * each workgroup pre-aggregates its tile in an LDS open-addressing table (packed {key:32,count:32}
//     slots, ds_cmpst_b64 / ds_add_u64); before touching LDS every wave collapses the current hot key
//     with ballot + popcount (half of all bids hit one auction id, event.rs:355-359);
//   * the tile's distinct (key, count) pairs are flushed with one 64-bit global atomic each into the
//     table of EVERY window that contains the pane (pane sharing: a bid is read once although it
//     belongs to window/hop windows);  MAX(num) falls out of the flush for free: counts only grow, so
//     the maximum over all fetch_add results is the final maximum;
//   * a second small kernel scans the window tables for count == max.
#include <algorithm>

#include "scan.hpp"

using namespace flockgpu;

namespace {

constexpr int kQ5Iters = 8;
constexpr int kQ5Tile = kBlock * 4 * kQ5Iters;  // 8192 rows
constexpr int kSlotBits = 11;
constexpr int kSlots = 1 << kSlotBits;          // 2048 x 8 B = 16 KiB LDS
constexpr int kLdsMaxProbe = 24;
constexpr uint32_t kFib = 0x9E3779B1u;

__device__ __forceinline__ bool lds_insert(uint64_t *tab, uint32_t key, uint32_t c) {
    uint32_t s = (key * kFib) >> (32 - kSlotBits);
    const uint64_t mine = ((uint64_t)key << 32) | c;
#pragma unroll 1
    for (int probe = 0; probe < kLdsMaxProbe; ++probe) {
        uint64_t cur = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == 0) {
            cur = atomicCAS(reinterpret_cast<unsigned long long *>(&tab[s]), 0ull, (unsigned long long)mine);
            if (cur == 0) return true;
        }
        if ((uint32_t)(cur >> 32) == key) {
            atomicAdd(reinterpret_cast<unsigned long long *>(&tab[s]), (unsigned long long)c);
            return true;
        }
        s = (s + 1) & (kSlots - 1);
    }
    return false;
}

// Returns the key's new count in this window (0 on table overflow, which also raises *err).
__device__ __forceinline__ uint32_t global_insert(uint64_t *tab, uint32_t cap, uint32_t key, uint32_t c, uint32_t *err) {
    uint32_t s = (uint32_t)(((uint64_t)(key * kFib) * cap) >> 32);
    const uint64_t mine = ((uint64_t)key << 32) | c;
#pragma unroll 1
    for (uint32_t probe = 0, lim = cap < 2048u ? cap : 2048u; probe < lim; ++probe) {
        uint64_t cur = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0) {
            uint64_t expected = 0;
            if (__hip_atomic_compare_exchange_strong(&tab[s], &expected, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT))
                return c;
            cur = expected;
        }
        if ((uint32_t)(cur >> 32) == key) {
            const uint64_t old = __hip_atomic_fetch_add(&tab[s], (uint64_t)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return (uint32_t)old + c;
        }
        s = (s + 1 == cap) ? 0 : s + 1;
    }
    atomicOr(err, 1u);
    return 0;
}

template <bool COLLAPSE>
__global__ __launch_bounds__(kBlock) void q5_count_kernel(const int32_t *__restrict__ auction, int64_t n_rows, SegTiles st,
                                                          const int32_t *__restrict__ pane_win_ptr,
                                                          const int32_t *__restrict__ pane_win_idx, uint64_t *tables,
                                                          uint32_t cap, uint64_t *win_max, uint32_t *err) {
    __shared__ uint64_t lds[kSlots];
    __shared__ uint32_t s_spill;  // keys that did not fit the LDS table go straight to the global tables
    for (int s = threadIdx.x; s < kSlots; s += kBlock) lds[s] = 0;
    if (threadIdx.x == 0) s_spill = 0;
    const TileRange tr = locate_tile(st, (int32_t)blockIdx.x, kQ5Tile);
    const int32_t wp0 = pane_win_ptr[tr.seg], wp1 = pane_win_ptr[tr.seg + 1];
    __syncthreads();
    if (wp0 == wp1) return;  // pane belongs to no (full) window

    const int lane = lane_id();
    int32_t k[kQ5Iters][4];
#pragma unroll
    for (int it = 0; it < kQ5Iters; ++it) {
        const int64_t r0 = tr.tile_begin + it * (kBlock * 4) + threadIdx.x * 4;
        if (r0 + 4 <= n_rows) {
            const int4 t = *reinterpret_cast<const int4 *>(auction + r0);
            k[it][0] = t.x; k[it][1] = t.y; k[it][2] = t.z; k[it][3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) k[it][j] = (r0 + j < n_rows) ? auction[r0 + j] : 0;
        }
    }
#pragma unroll
    for (int it = 0; it < kQ5Iters; ++it) {
        const int64_t r0 = tr.tile_begin + it * (kBlock * 4) + threadIdx.x * 4;
        bool v[4];
        uint32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = (r0 + j >= tr.lo) && (r0 + j < tr.hi);
            c[j] = 1;
        }
        if (COLLAPSE) {
            // wave-level collapse of the hot key: take the first live key of the wave as candidate
            const uint64_t live = __ballot(v[0]);
            if (live) {
                const int src = __ffsll((unsigned long long)live) - 1;
                const int32_t hot = __shfl(k[it][0], src, 64);
                uint32_t cnt = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool m = v[j] && k[it][j] == hot;
                    cnt += (uint32_t)__popcll((unsigned long long)__ballot(m));
                    v[j] = v[j] && !m;
                }
                if (lane == src && !lds_insert(lds, (uint32_t)hot, cnt)) {
                    // cannot happen for a sane table size, but stay correct: spill
                    atomicAdd(&s_spill, 1u);
                    for (int wi = wp0; wi < wp1; ++wi) {
                        const int32_t w = pane_win_idx[wi];
                        const uint32_t nc = global_insert(tables + (size_t)w * cap, cap, (uint32_t)hot, cnt, err);
                        atomicMax(reinterpret_cast<unsigned long long *>(&win_max[w]), (unsigned long long)nc);
                    }
                }
            }
        }
        // lane-local merge of equal keys
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j)
                if (v[i] && v[j] && k[it][i] == k[it][j]) {
                    c[i] += c[j];
                    v[j] = false;
                }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (v[j] && !lds_insert(lds, (uint32_t)k[it][j], c[j])) {
                atomicAdd(&s_spill, 1u);
                for (int wi = wp0; wi < wp1; ++wi) {
                    const int32_t w = pane_win_idx[wi];
                    const uint32_t nc = global_insert(tables + (size_t)w * cap, cap, (uint32_t)k[it][j], c[j], err);
                    atomicMax(reinterpret_cast<unsigned long long *>(&win_max[w]), (unsigned long long)nc);
                }
            }
        }
    }
    __syncthreads();
    // flush the tile's distinct keys into every window table that contains this pane
    for (int wi = wp0; wi < wp1; ++wi) {
        const int32_t w = pane_win_idx[wi];
        uint64_t *tab = tables + (size_t)w * cap;
        uint32_t best = 0;
#pragma unroll
        for (int s = threadIdx.x; s < kSlots; s += kBlock) {
            const uint64_t e = lds[s];
            if (e) {
                const uint32_t nc = global_insert(tab, cap, (uint32_t)(e >> 32), (uint32_t)e, err);
                best = nc > best ? nc : best;
            }
        }
        best = wave_max_u32(best);
        if (lane == 0 && best > 0) {
            const uint64_t seen = __hip_atomic_load(&win_max[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint64_t)best > seen) atomicMax(reinterpret_cast<unsigned long long *>(&win_max[w]), (unsigned long long)best);
        }
    }
}

// Scan every window table: count groups, append (window, key) of the rows whose count equals the window max.
__global__ __launch_bounds__(kBlock) void q5_select_kernel(const uint64_t *__restrict__ tables, uint32_t cap,
                                                           const uint64_t *__restrict__ win_max, uint64_t *win_groups,
                                                           uint32_t *cursor, uint32_t out_cap, int32_t *out_win,
                                                           int32_t *out_key) {
    const int32_t w = blockIdx.y;
    const uint64_t *tab = tables + (size_t)w * cap;
    const uint32_t mx = (uint32_t)win_max[w];
    uint32_t groups = 0;
    for (uint32_t s = blockIdx.x * kBlock + threadIdx.x; s < cap; s += gridDim.x * kBlock) {
        const uint64_t e = tab[s];
        if (e) {
            ++groups;
            if ((uint32_t)e == mx) {
                const uint32_t p = atomicAdd(cursor, 1u);
                if (p < out_cap) {
                    out_win[p] = w;
                    out_key[p] = (int32_t)(e >> 32);
                }
            }
        }
    }
    const uint64_t g = wave_sum_u64(groups);
    if (lane_id() == 0 && g) atomicAdd(reinterpret_cast<unsigned long long *>(&win_groups[w]), (unsigned long long)g);
}

}  // namespace

extern "C" {

int flockgpu_q5_hot_items(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                          flockgpu_q5_result *out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!bid || !out || bid->rows < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "q5: null argument");
    FG_TRY(check_windows(ctx, win, bid->rows, "q5"));
    if (bid->rows > 0 && !bid->auction) return fail(ctx, FLOCKGPU_ERR_INVALID, "q5: null auction column");
    if (reinterpret_cast<uintptr_t>(bid->auction) & 15)
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q5: auction column must be 16-byte aligned");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const int n_win = win->n_windows, n_panes = win->n_panes;

    // pane -> windows CSR, window row counts
    std::vector<int32_t> ptr(n_panes + 1, 0), idx;
    int64_t max_win_rows = 0;
    for (int w = 0; w < n_win; ++w) {
        for (int p = win->win_pane_lo[w]; p < win->win_pane_hi[w]; ++p) ++ptr[p + 1];
        const int64_t rows = win->pane_row_offsets[win->win_pane_hi[w]] - win->pane_row_offsets[win->win_pane_lo[w]];
        max_win_rows = std::max(max_win_rows, rows);
    }
    if (max_win_rows >= (int64_t(1) << 32))
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "q5: a window holds >= 2^32 rows (packed 32-bit counters)");
    for (int p = 0; p < n_panes; ++p) ptr[p + 1] += ptr[p];
    idx.resize(ptr[n_panes]);
    {
        std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
        for (int w = 0; w < n_win; ++w)
            for (int p = win->win_pane_lo[w]; p < win->win_pane_hi[w]; ++p) idx[fill[p]++] = w;
    }
    std::vector<int64_t> sb(n_panes), se(n_panes);
    for (int p = 0; p < n_panes; ++p) {
        sb[p] = win->pane_row_offsets[p];
        se[p] = win->pane_row_offsets[p + 1];
    }
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, "q5", sb.data(), se.data(), n_panes, kQ5Tile, &st));

    int32_t *d_ptr = nullptr, *d_idx = nullptr, *h_ptr = nullptr, *h_idx = nullptr;
    FG_TRY(arena_get_t(ctx, "q5.pane_win_ptr", ptr.size(), &d_ptr));
    FG_TRY(arena_get_t(ctx, "q5.pane_win_idx", idx.size() + 1, &d_idx));
    FG_TRY(pinned_get_t(ctx, "q5.pane_win_ptr", ptr.size(), &h_ptr));
    FG_TRY(pinned_get_t(ctx, "q5.pane_win_idx", idx.size() + 1, &h_idx));
    std::copy(ptr.begin(), ptr.end(), h_ptr);
    std::copy(idx.begin(), idx.end(), h_idx);
    FG_HIP(ctx, hipMemcpyAsync(d_ptr, h_ptr, ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (!idx.empty())
        FG_HIP(ctx, hipMemcpyAsync(d_idx, h_idx, idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));

    // device scalars: [0 .. n_win) win_max, [n_win .. 2 n_win) win_groups, then cursor + err (as 2 x u32)
    const size_t n_meta = (size_t)2 * n_win + 1;
    uint64_t *d_meta = nullptr, *h_meta = nullptr;
    FG_TRY(arena_get_t(ctx, "q5.meta", n_meta, &d_meta));
    FG_TRY(pinned_get_t(ctx, "q5.meta", n_meta, &h_meta));
    uint32_t *d_cursor = reinterpret_cast<uint32_t *>(d_meta + 2 * n_win), *d_err = d_cursor + 1;

    double rpg = ctx->q5_rows_per_group < 1.0 ? 1.0 : ctx->q5_rows_per_group;
    uint64_t cap64 = std::max<uint64_t>(1024, (uint64_t)((double)max_win_rows / rpg * 2.0) + 64);
    uint32_t out_cap = 1u << 16;
    std::vector<int32_t> h_win, h_key;
    uint32_t n_sel = 0;
    for (int attempt = 0;; ++attempt) {
        if (attempt > 8 || cap64 >= (uint64_t(1) << 31))
            return fail(ctx, FLOCKGPU_ERR_CAPACITY, "q5: hash table capacity %llu still overflows", (unsigned long long)cap64);
        const uint32_t cap = (uint32_t)cap64;
        uint64_t *tables = nullptr;
        FG_TRY(arena_get_t(ctx, "q5.tables", (size_t)cap * std::max(n_win, 1), &tables));
        int32_t *o_win = nullptr, *o_key = nullptr;
        FG_TRY(arena_get_t(ctx, "q5.sel_win", out_cap, &o_win));
        FG_TRY(arena_get_t(ctx, "q5.sel_key", out_cap, &o_key));
        FG_HIP(ctx, hipMemsetAsync(tables, 0, sizeof(uint64_t) * (size_t)cap * n_win, ctx->stream));
        FG_HIP(ctx, hipMemsetAsync(d_meta, 0, sizeof(uint64_t) * n_meta, ctx->stream));
        if (st.n_tiles > 0 && n_win > 0) {
            LaunchScope ls(ctx, "q5_count_kernel");
            hipLaunchKernelGGL(q5_count_kernel<true>, dim3((unsigned)st.n_tiles), dim3(kBlock), 0, ctx->stream, bid->auction,
                               bid->rows, st, d_ptr, d_idx, tables, cap, d_meta, d_err);
        }
        FG_TRY(check_launch(ctx, "q5_count_kernel"));
        if (n_win > 0) {
            LaunchScope ls(ctx, "q5_select_kernel");
            const unsigned gx = (unsigned)std::min<int64_t>(div_up(cap, kBlock * 8), 256);
            hipLaunchKernelGGL(q5_select_kernel, dim3(gx, (unsigned)n_win), dim3(kBlock), 0, ctx->stream, tables, cap, d_meta,
                               d_meta + n_win, d_cursor, out_cap, o_win, o_key);
        }
        FG_TRY(check_launch(ctx, "q5_select_kernel"));
        FG_HIP(ctx, hipMemcpyAsync(h_meta, d_meta, sizeof(uint64_t) * n_meta, hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const uint32_t *tail = reinterpret_cast<const uint32_t *>(h_meta + 2 * n_win);
        n_sel = tail[0];
        if (tail[1]) {  // a window table filled up: the group-count hint was too optimistic
            cap64 *= 4;
            continue;
        }
        if (n_sel > out_cap) {  // many ties: grow the winner buffer and redo (rare)
            out_cap = n_sel + 1024;
            continue;
        }
        h_win.resize(n_sel);
        h_key.resize(n_sel);
        if (n_sel) {
            FG_HIP(ctx, hipMemcpy(h_win.data(), o_win, sizeof(int32_t) * n_sel, hipMemcpyDeviceToHost));
            FG_HIP(ctx, hipMemcpy(h_key.data(), o_key, sizeof(int32_t) * n_sel, hipMemcpyDeviceToHost));
        }
        break;
    }
    // remember how dense the groups were so the next call sizes its tables right away
    {
        double best = 1e30;
        for (int w = 0; w < n_win; ++w) {
            const int64_t rows = win->pane_row_offsets[win->win_pane_hi[w]] - win->pane_row_offsets[win->win_pane_lo[w]];
            const uint64_t g = h_meta[n_win + w];
            if (g) best = std::min(best, (double)rows / (double)g);
        }
        if (best < 1e30) ctx->q5_rows_per_group = best;
    }
    // order the winners by (window, auction) and hand them back on the device
    std::vector<uint32_t> order(n_sel);
    for (uint32_t i = 0; i < n_sel; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return h_win[a] != h_win[b] ? h_win[a] < h_win[b] : h_key[a] < h_key[b];
    });
    std::vector<int64_t> &offs = ctx->host_i64["q5.win_out_offsets"];
    std::vector<uint64_t> &wmax = ctx->host_u64["q5.win_max"], &wgrp = ctx->host_u64["q5.win_groups"];
    offs.assign((size_t)n_win + 1, 0);
    wmax.assign(h_meta, h_meta + n_win);
    wgrp.assign(h_meta + n_win, h_meta + 2 * n_win);
    int32_t *h_oa = nullptr;
    uint64_t *h_on = nullptr;
    FG_TRY(pinned_get_t(ctx, "q5.out_auction", (size_t)n_sel + 1, &h_oa));
    FG_TRY(pinned_get_t(ctx, "q5.out_num", (size_t)n_sel + 1, &h_on));
    for (uint32_t i = 0; i < n_sel; ++i) {
        const uint32_t s = order[i];
        h_oa[i] = h_key[s];
        h_on[i] = wmax[h_win[s]];
        offs[h_win[s] + 1] += 1;
    }
    for (int w = 0; w < n_win; ++w) offs[w + 1] += offs[w];
    int32_t *d_oa = nullptr;
    uint64_t *d_on = nullptr;
    FG_TRY(arena_get_t(ctx, "q5.out_auction", (size_t)n_sel + 1, &d_oa));
    FG_TRY(arena_get_t(ctx, "q5.out_num", (size_t)n_sel + 1, &d_on));
    if (n_sel) {
        FG_HIP(ctx, hipMemcpyAsync(d_oa, h_oa, sizeof(int32_t) * n_sel, hipMemcpyHostToDevice, ctx->stream));
        FG_HIP(ctx, hipMemcpyAsync(d_on, h_on, sizeof(uint64_t) * n_sel, hipMemcpyHostToDevice, ctx->stream));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    out->auction = d_oa;
    out->num = d_on;
    out->win_out_offsets = offs.data();
    out->win_max = wmax.data();
    out->win_groups = wgrp.data();
    out->rows = n_sel;
    return FLOCKGPU_OK;
}

}  // extern "C"
