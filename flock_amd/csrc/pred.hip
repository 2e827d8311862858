// One-pass FilterExec predicates (pred.hpp): a postfix program over leaf comparisons, evaluated per 8192-row flag tile.
// Every leaf reads its column(s) with the tile's 16-byte lane loads and yields two 32-bit words per lane -- which of the
// lane's 32 rows are TRUE, which are NULL --, AND / OR / NOT combine those words (SQL three-valued logic), and the tile's
// flag words + wave counts are what count -> scan -> emit (scan.hpp) takes from there.  HBM traffic = each referenced
// column once + 1 KiB of flags per tile.
#include <algorithm>

#include "gather.hpp"
#include "pred.hpp"

using namespace flockgpu;

namespace {

struct TN {
    uint32_t t, n;   // per row of the lane: TRUE / NULL (never both); FALSE = neither
};

// the lane's rows of iteration `it`: r0 .. r0 + 3, r0 a multiple of 4
__device__ __forceinline__ void rows4_i32(const int32_t *__restrict__ col, int64_t r0, int64_t n_rows, bool full, bool stream, int32_t (&v)[4]) {
    if (full) {
        const int4 t = stream ? stream_load4(col + r0) : *reinterpret_cast<const int4 *>(col + r0);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        load4_i32(col, r0, n_rows, v);
    }
}
__device__ __forceinline__ void rows4_i64(const int64_t *__restrict__ col, int64_t r0, int64_t n_rows, bool full, int64_t (&v)[4]) {
    if (full) {
        const int4 a = *reinterpret_cast<const int4 *>(col + r0), b = *reinterpret_cast<const int4 *>(col + r0 + 2);
        v[0] = (int64_t)(((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x);
        v[1] = (int64_t)(((uint64_t)(uint32_t)a.w << 32) | (uint32_t)a.z);
        v[2] = (int64_t)(((uint64_t)(uint32_t)b.y << 32) | (uint32_t)b.x);
        v[3] = (int64_t)(((uint64_t)(uint32_t)b.w << 32) | (uint32_t)b.z);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (r0 + j >= 0 && r0 + j < n_rows) ? col[r0 + j] : 0;
    }
}
// any integer column's rows as int64 (Int32 sign-extended; UInt64 as its bit pattern)
__device__ __forceinline__ void rows4_int(const PredCol &c, int64_t r0, int64_t n_rows, bool full, int64_t (&v)[4]) {
    if (c.type == (int32_t)ColType::I32) {
        int32_t t[4];
        rows4_i32(static_cast<const int32_t *>(c.values), r0, n_rows, full, false, t);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = t[j];
    } else {
        rows4_i64(static_cast<const int64_t *>(c.values), r0, n_rows, full, v);
    }
}

// (sub >= 0: only iteration `sub` of the tile's eight is this workgroup's -- pred_flag_kernel's kSplit instance; the other rows' bits stay 0)
__device__ __forceinline__ uint32_t valid_bits(const uint8_t *__restrict__ valid, int64_t wbase, int64_t n_rows, bool full, int sub = -1) {
    if (!valid) return ~0u;
    uint32_t bits = 0;
#pragma unroll
    for (int it = 0; it < kFlagIters; ++it) {
        if (sub >= 0 && it != sub) continue;
        const int64_t r0 = wbase + it * 256;
        uint32_t w = 0;
        if (full) {
            w = *reinterpret_cast<const uint32_t *>(valid + r0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (r0 + j >= 0 && r0 + j < n_rows) w |= (uint32_t)valid[r0 + j] << (8 * j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) bits |= (uint32_t)(((w >> (8 * j)) & 0xffu) != 0) << (it * 4 + j);
    }
    return bits;
}

__device__ __forceinline__ uint64_t load_u64_any(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// (lt, eq[, gt]) row masks -> the comparison's mask: the operator is applied to whole words, once per lane, not per row
__device__ __forceinline__ uint32_t cmp_from_masks(int32_t op, uint32_t lt, uint32_t eq, uint32_t gt) {
    switch (op) {   // (uniform)
        case 0: return eq;
        case 1: return ~eq;
        case 2: return lt;
        case 3: return lt | eq;
        case 4: return gt;
        default: return gt | eq;
    }
}

template <bool kWide>
__device__ __forceinline__ TN leaf_cmp_int_lit(const PredProgram &P, const PredLeafDesc &L, int64_t wbase, int64_t n_rows, bool full, int sub) {
    const PredCol &c = P.cols[L.a];
    uint32_t lt = 0, eq = 0;
    if (c.type == (int32_t)ColType::I32) {
        // (the host folded literals outside the Int32 range into constants -- a remainder by |m| < 2^31 is an Int32, too: 32-bit comparisons)
        const int32_t *col = static_cast<const int32_t *>(c.values);
        const int32_t lit = (int32_t)L.lit;
        int32_t a[kFlagIters][4];
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it) {
            if (sub >= 0 && it != sub) { a[it][0] = a[it][1] = a[it][2] = a[it][3] = 0; continue; }
            rows4_i32(col, wbase + it * 256, n_rows, full, c.uses == 1, a[it]);
        }
        if (L.mod_kind == 1) {   // remainder and comparison row by row: the 32 values die as they are used
            const UMod32 mm = L.mod;
#pragma unroll
            for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int32_t r = smod32_apply(a[it][j], mm);
                    lt |= (uint32_t)(r < lit) << (it * 4 + j);
                    eq |= (uint32_t)(r == lit) << (it * 4 + j);
                }
        } else
        if (kWide && L.mod_kind == 2) {   // |m| beyond 2^31: a 64-bit software division per row, kept out of the common cases' registers
#pragma unroll 1
            for (int k = 0; k < kFlagIters * 4; ++k) {
                int32_t x = 0;
#pragma unroll
                for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) x = (k == it * 4 + j) ? a[it][j] : x;
                const int64_t r = (int64_t)x % L.modulus;
                lt |= (uint32_t)(r < L.lit) << k;
                eq |= (uint32_t)(r == L.lit) << k;
            }
        } else {
#pragma unroll
            for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lt |= (uint32_t)(a[it][j] < lit) << (it * 4 + j);
                    eq |= (uint32_t)(a[it][j] == lit) << (it * 4 + j);
                }
        }
    } else if (kWide) {
        const int64_t *col = static_cast<const int64_t *>(c.values);
        const uint64_t flip = L.uns ? 0ull : (1ull << 63);   // signed order as unsigned order
        const uint64_t lit = (uint64_t)L.lit ^ flip;
#pragma unroll 1
        for (int it = 0; it < kFlagIters; ++it) {
            if (sub >= 0 && it != sub) continue;
            int64_t v[4];
            rows4_i64(col, wbase + it * 256, n_rows, full, v);
            if (L.mod_kind == 2) {   // ONE software division in flight (four side by side take ~160 registers)
#pragma unroll 1
                for (int j = 0; j < 4; ++j) {
                    const int64_t x = j == 0 ? v[0] : j == 1 ? v[1] : j == 2 ? v[2] : v[3];
                    const uint64_t u = (uint64_t)(x % L.modulus) ^ flip;
                    lt |= (uint32_t)(u < lit) << (it * 4 + j);
                    eq |= (uint32_t)(u == lit) << (it * 4 + j);
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t u = (uint64_t)v[j] ^ flip;
                lt |= (uint32_t)(u < lit) << (it * 4 + j);
                eq |= (uint32_t)(u == lit) << (it * 4 + j);
            }
        }
    }
    const uint32_t valid = valid_bits(c.valid, wbase, n_rows, full, sub);
    return TN{cmp_from_masks(L.cmp, lt, eq, ~(lt | eq)) & valid, ~valid};
}

__device__ __forceinline__ TN leaf_cmp_f64_lit(const PredProgram &P, const PredLeafDesc &L, int64_t wbase, int64_t n_rows, bool full, int sub) {
    const PredCol &c = P.cols[L.a];
    const int64_t *col = static_cast<const int64_t *>(c.values);
    const double lit = __longlong_as_double(L.lit);
    uint32_t lt = 0, eq = 0, gt = 0;   // IEEE: a NaN is none of the three, so only != holds for it
#pragma unroll 1
    for (int it = 0; it < kFlagIters; ++it) {
        if (sub >= 0 && it != sub) continue;
        int64_t v[4];
        rows4_i64(col, wbase + it * 256, n_rows, full, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double x = __longlong_as_double(v[j]);
            lt |= (uint32_t)(x < lit) << (it * 4 + j);
            eq |= (uint32_t)(x == lit) << (it * 4 + j);
            gt |= (uint32_t)(x > lit) << (it * 4 + j);
        }
    }
    const uint32_t valid = valid_bits(c.valid, wbase, n_rows, full, sub);
    return TN{cmp_from_masks(L.cmp, lt, eq, gt) & valid, ~valid};
}

__device__ __forceinline__ TN leaf_cmp_col(const PredProgram &P, const PredLeafDesc &L, int64_t wbase, int64_t n_rows, bool full, bool f64, int sub) {
    const PredCol &ca = P.cols[L.a], &cb = P.cols[L.b];
    const uint64_t flip = L.uns ? 0ull : (1ull << 63);
    uint32_t lt = 0, eq = 0, gt = 0;
#pragma unroll 1
    for (int it = 0; it < kFlagIters; ++it) {
        if (sub >= 0 && it != sub) continue;
        int64_t x[4], y[4];
        rows4_int(ca, wbase + it * 256, n_rows, full, x);
        rows4_int(cb, wbase + it * 256, n_rows, full, y);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (f64) {
                const double p = __longlong_as_double(x[j]), q = __longlong_as_double(y[j]);
                lt |= (uint32_t)(p < q) << (it * 4 + j);
                eq |= (uint32_t)(p == q) << (it * 4 + j);
                gt |= (uint32_t)(p > q) << (it * 4 + j);
            } else {
                const uint64_t p = (uint64_t)x[j] ^ flip, q = (uint64_t)y[j] ^ flip;
                lt |= (uint32_t)(p < q) << (it * 4 + j);
                eq |= (uint32_t)(p == q) << (it * 4 + j);
                gt |= (uint32_t)(p > q) << (it * 4 + j);
            }
        }
    }
    const uint32_t valid = valid_bits(ca.valid, wbase, n_rows, full, sub) & valid_bits(cb.valid, wbase, n_rows, full, sub);
    return TN{cmp_from_masks(L.cmp, lt, eq, gt) & valid, ~valid};
}

// Utf8 column = literal: the length first, then the literal's bytes eight at a time -- each word read from an address clamped into
// the byte buffer (never under a per-row branch: DESIGN section 3) and shifted into place.
__device__ __forceinline__ TN leaf_utf8_eq(const PredProgram &P, const PredLeafDesc &L, int64_t wbase, int64_t n_rows, bool full, int sub) {
    const PredCol &c = P.cols[L.a];
    const int32_t *__restrict__ off = c.offsets;
    const uint8_t *__restrict__ bytes = static_cast<const uint8_t *>(c.values);
    const int32_t len = L.lit_len;
    const int64_t total = c.bytes;
    uint32_t bits = 0;
#pragma unroll 1
    for (int it = 0; it < kFlagIters; ++it) {
        if (sub >= 0 && it != sub) continue;
        const int64_t r0 = wbase + it * 256;
        int32_t o[5];
        if (full) {   // rows r0 .. r0 + 3 exist, so offsets r0 .. r0 + 4 do
            const int4 t = *reinterpret_cast<const int4 *>(off + r0);
            o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
            o[4] = off[r0 + 4];
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int64_t r = r0 + j < 0 ? 0 : (r0 + j > n_rows ? n_rows : r0 + j);
                o[j] = off[r];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t b0 = o[j];
            bool eq = (o[j + 1] - o[j]) == len;
            if (total >= 8) {
                for (int32_t k = 0; k < len; k += 8) {   // (uniform trip count)
                    const int64_t at = b0 + k, cl = at < total - 8 ? at : total - 8;
                    const int64_t sh = at - cl > 7 ? 7 : at - cl;
                    const uint64_t w = load_u64_any(bytes + cl) >> (8 * sh);
                    const uint64_t lw = load_u64_any(P.pool + L.lit_off + k);
                    const int nb = len - k < 8 ? len - k : 8;
                    const uint64_t mask = nb == 8 ? ~uint64_t(0) : ((uint64_t(1) << (8 * nb)) - 1);
                    eq = eq && ((w ^ lw) & mask) == 0;
                }
            } else if (eq) {   // a column of fewer than eight bytes in all
                for (int32_t k = 0; k < len; ++k) eq = eq && bytes[b0 + k] == P.pool[L.lit_off + k];
            }
            bits |= (uint32_t)(eq != (L.negate != 0)) << (it * 4 + j);
        }
    }
    const uint32_t valid = valid_bits(c.valid, wbase, n_rows, full, sub);
    return TN{bits & valid, ~valid};
}

// kWide: the program holds a leaf over 64-bit values (Int64 / UInt64 / Timestamp / Float64 columns, column-to-column comparisons, `%` by
// more than 2^31).  Programs without one -- Int32 and Utf8 columns against literals, what NEXMark's filters are -- run the instance
// that does not carry those paths' registers (78 against 190 VGPRs: occupancy is what a streaming pass lives on).
template <bool kWide>
__device__ __forceinline__ TN eval_leaf(const PredProgram &P, int which, int64_t wbase, int64_t n_rows, bool full, int sub) {
    const PredLeafDesc &L = P.leaves[which];
    switch (L.kind) {   // (uniform)
        case (uint8_t)PredLeafKind::CmpIntLit: return leaf_cmp_int_lit<kWide>(P, L, wbase, n_rows, full, sub);
        case (uint8_t)PredLeafKind::CmpF64Lit: return kWide ? leaf_cmp_f64_lit(P, L, wbase, n_rows, full, sub) : TN{0u, 0u};
        case (uint8_t)PredLeafKind::CmpIntCol: return kWide ? leaf_cmp_col(P, L, wbase, n_rows, full, false, sub) : TN{0u, 0u};
        case (uint8_t)PredLeafKind::CmpF64Col: return kWide ? leaf_cmp_col(P, L, wbase, n_rows, full, true, sub) : TN{0u, 0u};
        case (uint8_t)PredLeafKind::Utf8Eq: return leaf_utf8_eq(P, L, wbase, n_rows, full, sub);
        case (uint8_t)PredLeafKind::IsNull: {
            const uint32_t valid = valid_bits(P.cols[L.a].valid, wbase, n_rows, full, sub);
            return TN{L.negate ? valid : ~valid, 0u};
        }
        default: return TN{L.lit == 1 ? ~0u : 0u, L.lit == 2 ? ~0u : 0u};
    }
}

// kFull: every tile of the launch lies inside the relation (the grid's tiles first_tile .. ); the relation's last, ragged tile is a
// launch of its own with the guarded loads -- carrying both load paths in one kernel doubled its registers.
// kSplit (small relations; kFull = false): the grid is (tiles, 8) and workgroup (t, s) evaluates only iteration s of tile t -- 1024 rows, four per
// lane -- and ORs / adds its share into the tile's flag words / wave counts (cleared by the host).  A leaf's eight iterations are eight
// DEPENDENT rounds of loads (offsets, then bytes; a 64-bit software division at a time), which a streaming pass hides behind its other
// workgroups and a relation of a few tiles does not: q3's two filters over 6e4 auctions and 2e4 persons took 21 us per launch.
template <bool kFull, bool kWide, bool kSplit = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kFull ? (kWide ? 3 : 4) : 1))) void pred_flag_kernel(const PredProgram P, int64_t n_rows, SegTiles st, int32_t first_tile,
                                                           uint32_t *__restrict__ flag_words, uint32_t *__restrict__ counts) {
    const int sub = kSplit ? (int)blockIdx.y : -1;
    // operands waiting below the top of the stack: one column of words per thread, touched by that thread only (no barrier)
    __shared__ uint32_t s_t[kPredMaxStack][kBlock], s_n[kPredMaxStack][kBlock];
    const int32_t tile = first_tile + (int32_t)blockIdx.x;
    const TileRange tr = locate_tile(st, tile, kFlagTile);
    const int64_t wbase = tr.tile_begin + flag_rel0();
    constexpr bool full = kFull;
    TN top{0u, 0u};
    int sp = 0;   // operands on the stack, the top one in registers
    for (int i = 0; i < P.n_ops; ++i) {
        const uint8_t op = P.op[i];
        if (op == (uint8_t)PredOpKind::Leaf) {
            if (sp > 0) {
                s_t[sp - 1][threadIdx.x] = top.t;
                s_n[sp - 1][threadIdx.x] = top.n;
            }
            top = eval_leaf<kWide>(P, P.arg[i], wbase, n_rows, full, sub);
            ++sp;
        } else if (op == (uint8_t)PredOpKind::Not) {
            top.t = ~top.t & ~top.n;
        } else {
            const TN a{s_t[sp - 2][threadIdx.x], s_n[sp - 2][threadIdx.x]};
            const uint32_t fa = ~a.t & ~a.n, fb = ~top.t & ~top.n;
            uint32_t t, f;
            if (op == (uint8_t)PredOpKind::And) {
                t = a.t & top.t;
                f = fa | fb;
            } else {
                t = a.t | top.t;
                f = fa & fb;
            }
            top = TN{t, ~t & ~f};
            --sp;
        }
    }
    uint32_t flags = top.t;
    if (!kFull || !(tr.lo <= tr.tile_begin && tr.hi >= tr.tile_begin + kFlagTile)) {   // a tile at the relation's edge: rows outside [lo, hi) select nothing
        uint32_t in = 0;
#pragma unroll
        for (int it = 0; it < kFlagIters; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = wbase + it * 256 + j;
                in |= (uint32_t)(r >= tr.lo && r < tr.hi) << (it * 4 + j);
            }
        flags &= in;
    }
    if (kSplit) {
        flags &= 0xFu << (4 * sub);
        if (flags) atomicOr(&flag_words[(size_t)tile * kBlock + threadIdx.x], flags);
        const uint32_t c = (uint32_t)wave_sum_u64((uint64_t)__popc(flags));
        if (lane_id() == 0 && c) atomicAdd(&counts[(size_t)tile * kWavesPerBlock + (threadIdx.x >> 6)], c);
        return;
    }
    store_flags_and_counts(flags, tile, flag_words, counts);
}

}  // namespace

namespace flockgpu {

int pred_to_rows(flockgpu_ctx *ctx, const char *name, const PredProgram &prog, int64_t rows, int32_t **out_rows, int64_t *n_out) {
    const std::string base = name;
    int32_t *o_rows = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".rows").c_str(), (size_t)std::max<int64_t>(rows, 0) + 4, &o_rows));
    *out_rows = o_rows;
    *n_out = 0;
    if (rows <= 0) return FLOCKGPU_OK;
    if (rows >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than 2^31 rows", name);
    if (prog.n_ops < 1 || prog.max_stack > kPredMaxStack) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: malformed predicate program", name);
    int64_t sb = 0, se = rows;
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, (base + ".tiles").c_str(), &sb, &se, 1, kFlagTile, &st));
    uint32_t *flags = nullptr, *counts = nullptr;
    uint64_t *tile_base = nullptr;
    int64_t *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".flags").c_str(), (size_t)st.n_tiles * kBlock + 4, &flags));
    FG_TRY(arena_get_t(ctx, (base + ".counts").c_str(), (size_t)st.n_tiles * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, (base + ".base").c_str(), (size_t)st.n_tiles + 1, &tile_base));
    FG_TRY(pinned_get_t(ctx, (base + ".off").c_str(), 2, &h_off));
    pinned_pending(reinterpret_cast<uint64_t *>(h_off), 2);   // (wait_pinned below)
    bool wide = false;
    for (int i = 0; i < prog.n_leaves; ++i) {
        const PredLeafDesc &l = prog.leaves[i];
        const PredLeafKind k = (PredLeafKind)l.kind;
        wide = wide || k == PredLeafKind::CmpF64Lit || k == PredLeafKind::CmpIntCol || k == PredLeafKind::CmpF64Col ||
               (k == PredLeafKind::CmpIntLit && (l.mod_kind == 2 || prog.cols[l.a].type != (int32_t)ColType::I32));
    }
    // the tiles that lie inside the relation, then -- a launch of its own -- the ragged last one (one segment: there is at most one)
    const int32_t n_full = (int32_t)(rows / kFlagTile);
    if (st.n_tiles < 4 * ctx->num_cus) {   // a few tiles only: every tile's eight iterations on workgroups of their own (kSplit)
        FG_TRY(fill_words(ctx, FillList().add(flags, 0u, (uint64_t)st.n_tiles * kBlock).add(counts, 0u, (uint64_t)st.n_tiles * kWavesPerBlock)));
        LaunchScope ls(ctx, "pred_flag_kernel");
        if (wide) hipLaunchKernelGGL((pred_flag_kernel<false, true, true>), dim3((unsigned)st.n_tiles, kFlagIters), dim3(kBlock), 0, ctx->stream, prog, rows, st, 0, flags, counts);
        else hipLaunchKernelGGL((pred_flag_kernel<false, false, true>), dim3((unsigned)st.n_tiles, kFlagIters), dim3(kBlock), 0, ctx->stream, prog, rows, st, 0, flags, counts);
    } else {
        LaunchScope ls(ctx, "pred_flag_kernel");
        if (n_full > 0) {
            if (wide) hipLaunchKernelGGL((pred_flag_kernel<true, true>), dim3((unsigned)n_full), dim3(kBlock), 0, ctx->stream, prog, rows, st, 0, flags, counts);
            else hipLaunchKernelGGL((pred_flag_kernel<true, false>), dim3((unsigned)n_full), dim3(kBlock), 0, ctx->stream, prog, rows, st, 0, flags, counts);
        }
        if (st.n_tiles > n_full) {
            if (wide) hipLaunchKernelGGL((pred_flag_kernel<false, true>), dim3((unsigned)(st.n_tiles - n_full)), dim3(kBlock), 0, ctx->stream, prog, rows, st, n_full, flags, counts);
            else hipLaunchKernelGGL((pred_flag_kernel<false, false>), dim3((unsigned)(st.n_tiles - n_full)), dim3(kBlock), 0, ctx->stream, prog, rows, st, n_full, flags, counts);
        }
    }
    FG_TRY(check_launch(ctx, "pred_flag_kernel"));
    if (st.n_tiles <= 2048) {   // a relation of up to 1.7e7 rows: the emit sums the lower tiles' counts itself (one launch less; 16 B per lower tile from L2)
        FG_TRY(emit_flagged_rows_self(ctx, st, flags, counts, o_rows, h_off));
    } else {
        FG_TRY(launch_tile_scan(ctx, counts, st.n_tiles, tile_base, st.tile_first, st.n_seg, h_off));   // (the selected-row count goes straight into pinned memory)
        FG_TRY(emit_flagged_rows(ctx, st, flags, counts, tile_base, o_rows));
    }
    FG_TRY(wait_pinned(ctx, reinterpret_cast<const uint64_t *>(h_off), 2));
    *n_out = h_off[1];
    return FLOCKGPU_OK;
}

}  // namespace flockgpu
