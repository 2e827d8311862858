// The general expression evaluator of the plan layer: a postfix program per expression, interpreted per row (valprog.hpp).
#include <type_traits>

#include "valprog.hpp"

using namespace flockgpu;

namespace {

constexpr int kPre = 2;         // Int32 columns (the program's first) requested at the top of every pass
constexpr int kValGroups = 4;   // 16-byte groups of rows per pass of the program (see valprog_kernel; 1 / 2 / 4 measured: profiles/r06/expr_groups_ab.txt)
constexpr uint32_t kErrDivZero = 1u, kErrCast = 2u, kErrNull = 4u, kErrOverflow = 8u;

__device__ __forceinline__ double as_f64(uint64_t b) { return __longlong_as_double((long long)b); }
__device__ __forceinline__ uint64_t f64_bits(double d) { return (uint64_t)__double_as_longlong(d); }

// Values on the stack: Int32 sign-extended to 64 bits, Int64 / UInt64 as they are, Float64 as its bits, BOOL 0 / 1.
__device__ __forceinline__ uint64_t load_value(const ValCol &c, int64_t i) {
    if (c.type == (int32_t)ColType::I32) return (uint64_t)(int64_t)static_cast<const int32_t *>(c.values)[i];
    return static_cast<const uint64_t *>(c.values)[i];
}

__device__ __forceinline__ bool cast_value(uint64_t v, uint8_t from, uint8_t to, uint64_t *out) {
    const uint8_t I32 = (uint8_t)ValType::I32, I64 = (uint8_t)ValType::I64, U64 = (uint8_t)ValType::U64, F64 = (uint8_t)ValType::F64;
    *out = v;
    if (from == to) return true;
    if (from == I32 || from == I64) {
        const int64_t s = (int64_t)v;
        if (to == I32) return s >= INT32_MIN && s <= INT32_MAX;
        if (to == I64) return true;
        if (to == U64) return s >= 0;
        *out = f64_bits((double)s);
        return true;
    }
    if (from == U64) {
        if (to == I32) return v <= (uint64_t)INT32_MAX;
        if (to == I64) return v <= (uint64_t)INT64_MAX;
        *out = f64_bits((double)v);
        return true;
    }
    const double t = trunc(as_f64(v));   // (NaN stays NaN: every comparison below is false)
    if (to == I32) {
        if (!(t >= -2147483648.0 && t <= 2147483647.0)) return false;
        *out = (uint64_t)(int64_t)t;
        return true;
    }
    if (to == I64) {
        if (!(t >= -9223372036854775808.0 && t < 9223372036854775808.0)) return false;
        *out = (uint64_t)(int64_t)t;
        return true;
    }
    if (to == U64) {
        if (!(t >= 0.0 && t < 18446744073709551616.0)) return false;   // (-0.9 truncates to -0.0, which fits)
        *out = (uint64_t)t;
        return true;
    }
    return false;
}

__device__ __forceinline__ uint64_t wrap_to(uint8_t type, uint64_t r) { return type == (uint8_t)ValType::I32 ? (uint64_t)(int64_t)(int32_t)(uint32_t)r : r; }

__device__ __forceinline__ uint64_t arith(uint8_t kind, uint8_t type, uint64_t a, uint64_t b, uint32_t *bad) {
    const uint8_t ADD = (uint8_t)ValOpKind::Add, SUB = (uint8_t)ValOpKind::Sub, MUL = (uint8_t)ValOpKind::Mul, DIV = (uint8_t)ValOpKind::Div;
    if (type == (uint8_t)ValType::F64) {
        const double x = as_f64(a), y = as_f64(b);
        if ((kind == DIV || kind == (uint8_t)ValOpKind::Mod) && y == 0.0) {   // (A-V3: arrow-rs tests is_zero() for floats, too; -0.0 == 0.0)
            *bad |= kErrDivZero;
            return 0;
        }
        return f64_bits(kind == ADD ? x + y : kind == SUB ? x - y : kind == MUL ? x * y : kind == DIV ? x / y : fmod(x, y));
    }
    if (kind == ADD || kind == SUB || kind == MUL) return wrap_to(type, kind == ADD ? a + b : kind == SUB ? a - b : a * b);   // wraps at the type's width
    if (b == 0) {
        *bad |= kErrDivZero;
        return 0;
    }
    if (type == (uint8_t)ValType::U64) return kind == DIV ? a / b : a % b;
    const int64_t x = (int64_t)a, y = (int64_t)b;
    if (y == -1) {   // (A-V4: INT_MIN / -1 and INT_MIN % -1 overflow; any other x: -x, 0)
        if (x == (type == (uint8_t)ValType::I32 ? (int64_t)INT32_MIN : INT64_MIN)) *bad |= kErrOverflow;
        return kind == DIV ? wrap_to(type, 0 - a) : 0;
    }
    if (type == (uint8_t)ValType::I32) {
        const int32_t x32 = (int32_t)x, y32 = (int32_t)y;
        return (uint64_t)(int64_t)(kind == DIV ? x32 / y32 : x32 % y32);
    }
    return (uint64_t)(kind == DIV ? x / y : x % y);
}

__device__ __forceinline__ bool compare(uint8_t kind, uint8_t type, uint64_t a, uint64_t b) {
    bool lt, eq;
    if (type == (uint8_t)ValType::F64) {
        const double x = as_f64(a), y = as_f64(b);
        lt = x < y;
        eq = x == y;
        if (x != x || y != y) return kind == (uint8_t)ValOpKind::Ne;   // IEEE: only <> holds against NaN
    } else if (type == (uint8_t)ValType::U64) {
        lt = a < b;
        eq = a == b;
    } else {
        lt = (int64_t)a < (int64_t)b;
        eq = a == b;
    }
    switch ((ValOpKind)kind) {
        case ValOpKind::Eq: return eq;
        case ValOpKind::Ne: return !eq;
        case ValOpKind::Lt: return lt;
        case ValOpKind::Le: return lt || eq;
        case ValOpKind::Gt: return !lt && !eq;
        default: return !lt;   // Ge
    }
}

// x / c and x % c for an integer literal c != 0 through the reciprocal the host made (ValBuilder::fuse_immediate): unsigned quotient of the
// absolute values, then the signs (truncation towards zero; the remainder takes the dividend's sign).  Everything that depends on the LITERAL
// only (power of two, the 65-bit multiplier form, the divisor's sign) is uniform and arrives as template / scalar arguments, so the per-value code
// is straight-line: selects, no branches (a branch per value was an exec-mask save / restore per value -- the kernel was bound by its SCALAR
// instruction stream: 1120 SALU against 860 VALU instructions per pass of `expr_filter`, profiles/r06/expr_groups_ab.txt).
// kForm: 0 = power of two (shift), 1 = multiplier, 2 = multiplier with the add step.
template <uint8_t KIND, uint8_t TYPE, int kForm>
__device__ __forceinline__ uint64_t div_recip64(uint64_t a, uint64_t c, uint64_t ad, bool neg_c, uint64_t magic, uint32_t shift) {
    constexpr bool sgn = TYPE != (uint8_t)ValType::U64;
    const bool neg_a = sgn && (int64_t)a < 0;
    const uint64_t ax = neg_a ? 0 - a : a;
    uint64_t q;
    if (kForm == 0) {
        q = ax >> shift;
    } else {
        const uint64_t hi = __umul64hi(ax, magic);
        q = kForm == 2 ? (((ax - hi) >> 1) + hi) >> shift : hi >> shift;
    }
    if (KIND == (uint8_t)ValOpKind::Div) return wrap_to(TYPE, sgn && (neg_a != neg_c) ? 0 - q : q);
    const uint64_t r = ax - q * ad;
    return neg_a ? 0 - r : r;
}
template <uint8_t KIND, uint8_t TYPE, int kForm>
__device__ __forceinline__ uint64_t div_recip32(uint64_t a, uint32_t d, bool neg_c, uint32_t magic, uint32_t shift) {
    constexpr bool sgn = TYPE != (uint8_t)ValType::U64;
    const bool neg_a = sgn && (int64_t)a < 0;
    const uint32_t n = (uint32_t)(neg_a ? 0 - a : a);
    uint32_t q;
    if (kForm == 0) {
        q = n >> shift;
    } else {
        const uint32_t hi = __umulhi(n, magic);
        q = kForm == 2 ? (((n - hi) >> 1) + hi) >> shift : hi >> shift;
    }
    if (KIND == (uint8_t)ValOpKind::Div) return wrap_to(TYPE, sgn && (neg_a != neg_c) ? 0 - (uint64_t)q : (uint64_t)q);
    const uint32_t r = n - q * d;
    return neg_a ? 0 - (uint64_t)r : (uint64_t)r;
}

// Operator kind and operand type are the same for every lane and every row of a pass: the interpreter branches on them ONCE per operator and
// pass (scalar branches) into code in which both are compile-time constants -- the per-value functions above then fold to the few
// instructions of the one case (a dispatch per VALUE made the kernel issue-bound: 0.39 ms for `price * 2 + 1` over 9.2e7 rows).
template <uint8_t V> using U8 = std::integral_constant<uint8_t, V>;
template <class F> __device__ __forceinline__ void with_type(uint8_t t, F &&f) {
    switch (t) {
        case (uint8_t)ValType::I32: f(U8<(uint8_t)ValType::I32>{}); break;
        case (uint8_t)ValType::I64: f(U8<(uint8_t)ValType::I64>{}); break;
        case (uint8_t)ValType::U64: f(U8<(uint8_t)ValType::U64>{}); break;
        default: f(U8<(uint8_t)ValType::F64>{}); break;
    }
}
template <class F> __device__ __forceinline__ void with_arith(uint8_t k, F &&f) {
    switch ((ValOpKind)k) {
        case ValOpKind::Add: f(U8<(uint8_t)ValOpKind::Add>{}); break;
        case ValOpKind::Sub: f(U8<(uint8_t)ValOpKind::Sub>{}); break;
        case ValOpKind::Mul: f(U8<(uint8_t)ValOpKind::Mul>{}); break;
        case ValOpKind::Div: f(U8<(uint8_t)ValOpKind::Div>{}); break;
        default: f(U8<(uint8_t)ValOpKind::Mod>{}); break;
    }
}
template <class F> __device__ __forceinline__ void with_cmp(uint8_t k, F &&f) {
    switch ((ValOpKind)k) {
        case ValOpKind::Eq: f(U8<(uint8_t)ValOpKind::Eq>{}); break;
        case ValOpKind::Ne: f(U8<(uint8_t)ValOpKind::Ne>{}); break;
        case ValOpKind::Lt: f(U8<(uint8_t)ValOpKind::Lt>{}); break;
        case ValOpKind::Le: f(U8<(uint8_t)ValOpKind::Le>{}); break;
        case ValOpKind::Gt: f(U8<(uint8_t)ValOpKind::Gt>{}); break;
        default: f(U8<(uint8_t)ValOpKind::Ge>{}); break;
    }
}

// One pass of the program over FOUR consecutive rows per lane.  Values on the stack: 4 x 64 bits + 4 validity bits; the top of the stack in
// registers (tv / tok), what waits below it in the lane's own columns of `s_v` / `s_ok` (no bank conflicts, no barrier).  r0: the lane's
// first row; rows at or beyond n load nothing and come out NULL.
struct Vec4 {
    uint64_t v[4];
    uint32_t ok;   // bit j: row j holds a value
};

__device__ __forceinline__ Vec4 load_col4(const ValCol &c, int64_t r0, int64_t n, bool whole) {
    Vec4 x;
    x.ok = 0;
    if (whole) {   // (block-uniform) the tile lies inside the relation: 16-byte loads (columns are 16-byte aligned: arena / Arrow buffers)
        if (c.type == (int32_t)ColType::I32) {
            const int4 t = stream_load4(static_cast<const int32_t *>(c.values) + r0);
            x.v[0] = (uint64_t)(int64_t)t.x; x.v[1] = (uint64_t)(int64_t)t.y; x.v[2] = (uint64_t)(int64_t)t.z; x.v[3] = (uint64_t)(int64_t)t.w;
        } else {
            const uint4 lo = stream_load4u(reinterpret_cast<const uint32_t *>(static_cast<const uint64_t *>(c.values) + r0));
            const uint4 hi = stream_load4u(reinterpret_cast<const uint32_t *>(static_cast<const uint64_t *>(c.values) + r0 + 2));
            x.v[0] = ((uint64_t)lo.y << 32) | lo.x; x.v[1] = ((uint64_t)lo.w << 32) | lo.z;
            x.v[2] = ((uint64_t)hi.y << 32) | hi.x; x.v[3] = ((uint64_t)hi.w << 32) | hi.z;
        }
        if (c.valid) {
            const uint32_t vb = *reinterpret_cast<const uint32_t *>(c.valid + r0);   // (r0 is a multiple of 4)
            x.ok = (vb & 0xFFu ? 1u : 0u) | (vb & 0xFF00u ? 2u : 0u) | (vb & 0xFF0000u ? 4u : 0u) | (vb & 0xFF000000u ? 8u : 0u);
        } else {
            x.ok = 15u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + j;
            const bool ok = r < n && (!c.valid || c.valid[r]);
            x.v[j] = ok ? load_value(c, r) : 0;
            x.ok |= (ok ? 1u : 0u) << j;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (!((x.ok >> j) & 1u)) x.v[j] = 0;   // (the slot of a NULL holds 0: no arithmetic on garbage, no spurious division by zero)
    return x;
}

// kG: 16-byte groups per pass -- a pass of the program evaluates 4 kG rows per lane (kG = 4: a lane has four loads per column operand in
// flight and one scalar dispatch per operator and sixteen rows; kG = 1 for programs whose operand stack would not fit the LDS otherwise).
template <bool kMask, int kG>
__global__ __launch_bounds__(kBlock) void valprog_kernel(ValProgram p, int64_t n, int32_t n_tiles, void *__restrict__ out_values, uint8_t *__restrict__ out_valid,
                                                         int32_t out_type, uint32_t *__restrict__ flag_words, uint32_t *__restrict__ counts, uint32_t *err) {
    extern __shared__ __attribute__((aligned(16))) uint64_t s_dyn[];   // [max_stack - 1][kG][4][kBlock] values, then [max_stack - 1][kG][kBlock] validity nibbles
    const int t = threadIdx.x;
    const int below = p.max_stack > 1 ? p.max_stack - 1 : 0;
    uint64_t *s_v = s_dyn;
    uint8_t *s_ok = reinterpret_cast<uint8_t *>(s_dyn + (size_t)below * kG * 4 * kBlock);
    uint32_t bad = 0;
    for (int32_t tile = (int32_t)blockIdx.x; tile < n_tiles; tile += (int32_t)gridDim.x) {
        const int64_t tile_begin = (int64_t)tile * kFlagTile;
        const bool whole = tile_begin + kFlagTile <= n;   // (block-uniform)
        uint32_t flags = 0;
#pragma unroll 1
        for (int it0 = 0; it0 < kFlagIters; it0 += kG) {
            const int64_t r0 = tile_begin + flag_rel0() + it0 * 256;   // group g: rows r0 + 256 g .. + 3
            Vec4 top[kG];
#pragma unroll
            for (int g = 0; g < kG; ++g) {
                top[g].ok = 0;
                top[g].v[0] = top[g].v[1] = top[g].v[2] = top[g].v[3] = 0;
            }
            // every Int32 column among the program's first kPre is ASKED FOR here, before the first operator: a pass that reads two columns
            // then has both in flight together instead of waiting for them one after the other (expr_filter: 0.50 -> see profiles/r06)
            int4 raw[kPre][kG];
            if (whole) {
#pragma unroll
                for (int c = 0; c < kPre; ++c)
                    if (c < p.n_cols && p.cols[c].type == (int32_t)ColType::I32) {
#pragma unroll
                        for (int g = 0; g < kG; ++g) raw[c][g] = stream_load4(static_cast<const int32_t *>(p.cols[c].values) + r0 + g * 256);
                    }
            }
            auto prefetched = [&](int c, int g) -> Vec4 {   // column c < kPre, Int32, tile inside the relation
                const int4 t4 = c == 0 ? raw[0][g] : raw[kPre - 1][g];
                Vec4 x;
                x.v[0] = (uint64_t)(int64_t)t4.x; x.v[1] = (uint64_t)(int64_t)t4.y; x.v[2] = (uint64_t)(int64_t)t4.z; x.v[3] = (uint64_t)(int64_t)t4.w;
                x.ok = 15u;
                const uint8_t *valid = p.cols[c].valid;
                if (valid) {
                    const uint32_t vb = *reinterpret_cast<const uint32_t *>(valid + r0 + g * 256);
                    x.ok = (vb & 0xFFu ? 1u : 0u) | (vb & 0xFF00u ? 2u : 0u) | (vb & 0xFF0000u ? 4u : 0u) | (vb & 0xFF000000u ? 8u : 0u);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!((x.ok >> j) & 1u)) x.v[j] = 0;
                }
                return x;
            };
            int sp = 0;   // operands on the stack, the top one in `top`
            auto spill = [&]() {   // a push over a live top: it goes to its slot below
                if (sp > 0) {
#pragma unroll
                    for (int g = 0; g < kG; ++g) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) s_v[(((size_t)(sp - 1) * kG + g) * 4 + j) * kBlock + t] = top[g].v[j];
                        s_ok[((size_t)(sp - 1) * kG + g) * kBlock + t] = (uint8_t)top[g].ok;
                    }
                }
            };
            auto second = [&](int g) -> Vec4 {   // group g of the operand below the top
                Vec4 a;
#pragma unroll
                for (int j = 0; j < 4; ++j) a.v[j] = s_v[(((size_t)(sp - 2) * kG + g) * 4 + j) * kBlock + t];
                a.ok = s_ok[((size_t)(sp - 2) * kG + g) * kBlock + t];
                return a;
            };
#pragma unroll 1
            for (int o = 0; o < p.n_ops; ++o) {
                const ValOp op = p.ops[o];
                const uint32_t at = op.arg & 31u;   // (fetching the NEXT operator and its constants ahead of its turn measured slower: profiles/r06/expr_groups_ab.txt)
                const uint64_t k0 = p.consts[at], k1 = p.consts[at + 1], k2 = p.consts[at + 2], k3 = p.consts[at + 3];
                const bool imm = op.to == kValImm || op.to == kValImm32;
                switch ((ValOpKind)op.kind) {
                    case ValOpKind::Col:
                        spill();
                        if (whole && op.arg < kPre && p.cols[op.arg].type == (int32_t)ColType::I32) {
#pragma unroll
                            for (int g = 0; g < kG; ++g) top[g] = prefetched(op.arg, g);
                        } else {
#pragma unroll
                            for (int g = 0; g < kG; ++g) top[g] = load_col4(p.cols[op.arg], r0 + g * 256, n, whole);
                        }
                        ++sp;
                        break;
                    case ValOpKind::Const:
                        spill();
#pragma unroll
                        for (int g = 0; g < kG; ++g) {
                            top[g].v[0] = top[g].v[1] = top[g].v[2] = top[g].v[3] = k0;
                            top[g].ok = 15u;
                        }
                        ++sp;
                        break;
                    case ValOpKind::Null:
                        spill();
#pragma unroll
                        for (int g = 0; g < kG; ++g) {
                            top[g].v[0] = top[g].v[1] = top[g].v[2] = top[g].v[3] = 0;
                            top[g].ok = 0;
                        }
                        ++sp;
                        break;
                    case ValOpKind::Add: case ValOpKind::Sub: case ValOpKind::Mul: case ValOpKind::Div: case ValOpKind::Mod: {
                        with_type(op.type, [&](auto T) {
                            with_arith(op.kind, [&](auto K) {
                                constexpr uint8_t ty = decltype(T)::value, kd = decltype(K)::value;
                                constexpr bool by_recip = (kd == (uint8_t)ValOpKind::Div || kd == (uint8_t)ValOpKind::Mod) && ty != (uint8_t)ValType::F64;
                                // f(value) for every row of the pass, NULL rows kept at 0: straight-line code, no branch per value
                                auto each = [&](auto &&f) {
#pragma unroll
                                    for (int g = 0; g < kG; ++g)
#pragma unroll
                                        for (int j = 0; j < 4; ++j) top[g].v[j] = f(top[g].v[j]) & (0 - (uint64_t)((top[g].ok >> j) & 1u));
                                };
                                if (imm && by_recip) {   // left = top, right = a non-zero integer literal: multiply by its reciprocal
                                    const uint64_t c = k0;
                                    const bool neg_c = ty != (uint8_t)ValType::U64 && (int64_t)c < 0;
                                    const uint64_t ad = neg_c ? 0 - c : c;
                                    if (neg_c && ad == 1) {   // (uniform) c = -1: INT_MIN / -1 and INT_MIN % -1 overflow (A-V4)
                                        const uint64_t lowest = ty == (uint8_t)ValType::I32 ? (uint64_t)(int64_t)INT32_MIN : (uint64_t)INT64_MIN;
#pragma unroll
                                        for (int g = 0; g < kG; ++g)
#pragma unroll
                                            for (int j = 0; j < 4; ++j) bad |= (top[g].v[j] == lowest && ((top[g].ok >> j) & 1u)) ? kErrOverflow : 0u;
                                    }
                                    if (op.to == kValImm32) {   // the dividend is known to fit Int32
                                        const uint64_t r32 = k3;
                                        const uint32_t magic = (uint32_t)r32, shift = (uint32_t)(r32 >> 32) & 31u, d = (uint32_t)ad;
                                        if (magic == 0) each([&](uint64_t a) { return div_recip32<kd, ty, 0>(a, d, neg_c, magic, shift); });
                                        else if ((r32 >> 40) & 1u) each([&](uint64_t a) { return div_recip32<kd, ty, 2>(a, d, neg_c, magic, shift); });
                                        else each([&](uint64_t a) { return div_recip32<kd, ty, 1>(a, d, neg_c, magic, shift); });
                                    } else {
                                        const uint64_t magic = k1;
                                        const uint32_t sh = (uint32_t)k2, shift = sh & 63u;
                                        if (magic == 0) each([&](uint64_t a) { return div_recip64<kd, ty, 0>(a, c, ad, neg_c, magic, shift); });
                                        else if (sh >> 8) each([&](uint64_t a) { return div_recip64<kd, ty, 2>(a, c, ad, neg_c, magic, shift); });
                                        else each([&](uint64_t a) { return div_recip64<kd, ty, 1>(a, c, ad, neg_c, magic, shift); });
                                    }
                                } else if (imm && (kd == (uint8_t)ValOpKind::Add || kd == (uint8_t)ValOpKind::Sub || kd == (uint8_t)ValOpKind::Mul)) {
                                    const uint64_t c = k0;
                                    uint32_t none = 0;   // (+ - * raise nothing)
                                    each([&](uint64_t a) { return arith(kd, ty, a, c, &none); });
                                } else if (imm) {   // Float64 / and % by a literal (a zero literal: the error, for the valid rows)
                                    const uint64_t c = k0;
#pragma unroll
                                    for (int g = 0; g < kG; ++g)
#pragma unroll
                                        for (int j = 0; j < 4; ++j)
                                            if ((top[g].ok >> j) & 1u) top[g].v[j] = arith(kd, ty, top[g].v[j], c, &bad);
                                } else {
#pragma unroll
                                    for (int g = 0; g < kG; ++g) {
                                        const Vec4 a = second(g);
                                        const uint32_t ok = a.ok & top[g].ok;
                                        if (kd == (uint8_t)ValOpKind::Add || kd == (uint8_t)ValOpKind::Sub || kd == (uint8_t)ValOpKind::Mul) {
                                            uint32_t none = 0;
#pragma unroll
                                            for (int j = 0; j < 4; ++j) top[g].v[j] = arith(kd, ty, a.v[j], top[g].v[j], &none) & (0 - (uint64_t)((ok >> j) & 1u));
                                        } else {
#pragma unroll
                                            for (int j = 0; j < 4; ++j) top[g].v[j] = (ok >> j) & 1u ? arith(kd, ty, a.v[j], top[g].v[j], &bad) : 0;
                                        }
                                        top[g].ok = ok;
                                    }
                                }
                            });
                        });
                        if (!imm) --sp;
                        break;
                    }
                    case ValOpKind::Neg:
                        with_type(op.type, [&](auto T) {
                            constexpr uint8_t ty = decltype(T)::value;
#pragma unroll
                            for (int g = 0; g < kG; ++g)
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const uint64_t a = top[g].v[j];
                                    top[g].v[j] = ty == (uint8_t)ValType::F64 ? ((top[g].ok >> j) & 1u ? f64_bits(-as_f64(a)) : 0)
                                                  : ty == (uint8_t)ValType::I32 ? (uint64_t)(int64_t)(int32_t)(0u - (uint32_t)a) : 0 - a;
                                }
                        });
                        break;
                    case ValOpKind::Cast: case ValOpKind::TryCast: {
                        const bool checked = (ValOpKind)op.kind == ValOpKind::Cast;
                        with_type(op.type, [&](auto F) {
                            with_type(op.to, [&](auto T) {
                                constexpr uint8_t from = decltype(F)::value, to = decltype(T)::value;
#pragma unroll
                                for (int g = 0; g < kG; ++g)
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
                                        if ((top[g].ok >> j) & 1u) {
                                            uint64_t r = 0;
                                            if (cast_value(top[g].v[j], from, to, &r)) {
                                                top[g].v[j] = r;
                                            } else {
                                                top[g].v[j] = 0;
                                                top[g].ok &= ~(1u << j);
                                                if (checked) bad |= kErrCast;
                                            }
                                        }
                            });
                        });
                        break;
                    }
                    case ValOpKind::Eq: case ValOpKind::Ne: case ValOpKind::Lt: case ValOpKind::Le: case ValOpKind::Gt: case ValOpKind::Ge: {
                        with_type(op.type, [&](auto T) {
                            with_cmp(op.kind, [&](auto K) {
                                constexpr uint8_t ty = decltype(T)::value, kd = decltype(K)::value;
                                if (imm) {   // (results as integers and bit arithmetic: Boolean `&&` of lane conditions is scalar-unit work)
                                    const uint64_t c = k0;
#pragma unroll
                                    for (int g = 0; g < kG; ++g)
#pragma unroll
                                        for (int j = 0; j < 4; ++j) top[g].v[j] = (compare(kd, ty, top[g].v[j], c) ? 1u : 0u) & (top[g].ok >> j);
                                } else {
#pragma unroll
                                    for (int g = 0; g < kG; ++g) {
                                        const Vec4 a = second(g);
                                        const uint32_t ok = a.ok & top[g].ok;
#pragma unroll
                                        for (int j = 0; j < 4; ++j) top[g].v[j] = (compare(kd, ty, a.v[j], top[g].v[j]) ? 1u : 0u) & (ok >> j);
                                        top[g].ok = ok;
                                    }
                                }
                            });
                        });
                        if (!imm) --sp;
                        break;
                    }
                    case ValOpKind::And: case ValOpKind::Or: {   // Kleene, on nibbles: bit j = row j (values are 0 / 1)
#pragma unroll
                        for (int g = 0; g < kG; ++g) {
                            const Vec4 a = second(g);
                            const uint32_t va = ((uint32_t)a.v[0] & 1u) | (((uint32_t)a.v[1] & 1u) << 1) | (((uint32_t)a.v[2] & 1u) << 2) | (((uint32_t)a.v[3] & 1u) << 3);
                            const uint32_t vb = ((uint32_t)top[g].v[0] & 1u) | (((uint32_t)top[g].v[1] & 1u) << 1) | (((uint32_t)top[g].v[2] & 1u) << 2) | (((uint32_t)top[g].v[3] & 1u) << 3);
                            const uint32_t oa = a.ok, ob = top[g].ok, ta = va & oa, tb = vb & ob, fa = ~va & oa, fb = ~vb & ob;   // TRUE / FALSE per side
                            uint32_t v, ok;
                            if ((ValOpKind)op.kind == ValOpKind::And) {
                                v = ta & tb;
                                ok = fa | fb | (oa & ob);
                            } else {
                                v = ta | tb;
                                ok = v | (oa & ob);
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) top[g].v[j] = (v >> j) & 1u;
                            top[g].ok = ok & 15u;
                        }
                        --sp;
                        break;
                    }
                    case ValOpKind::Not:
#pragma unroll
                        for (int g = 0; g < kG; ++g)
#pragma unroll
                            for (int j = 0; j < 4; ++j) top[g].v[j] = ((uint32_t)top[g].v[j] ^ 1u) & (top[g].ok >> j) & 1u;
                        break;
                    case ValOpKind::IsNull: case ValOpKind::IsNotNull: {
                        const uint32_t flip = (ValOpKind)op.kind == ValOpKind::IsNotNull ? 0u : 1u;
#pragma unroll
                        for (int g = 0; g < kG; ++g) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) top[g].v[j] = ((top[g].ok >> j) & 1u) ^ flip;
                            top[g].ok = 15u;
                        }
                        break;
                    }
                    case ValOpKind::Select: {   // [.. ELSE WHEN THEN]: THEN = top, WHEN below it, ELSE below that
#pragma unroll
                        for (int g = 0; g < kG; ++g) {
                            const Vec4 when = second(g);
                            --sp;
                            const Vec4 els = second(g);
                            ++sp;
                            uint32_t ok_out = 0;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const bool take = ((when.ok >> j) & 1u) && when.v[j];
                                if (!take) top[g].v[j] = els.v[j];
                                ok_out |= ((take ? top[g].ok : els.ok) >> j & 1u) << j;
                            }
                            top[g].ok = ok_out;
                        }
                        sp -= 2;
                        break;
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < kG; ++g) {
                const int64_t rg = r0 + g * 256;
                if (kMask) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) flags |= (uint32_t)(((top[g].ok >> j) & 1u) && top[g].v[j] && rg + j < n) << ((it0 + g) * 4 + j);
                } else if (whole) {
                    if (out_type == (int32_t)ColType::I32) {
                        *reinterpret_cast<int4 *>(static_cast<int32_t *>(out_values) + rg) = make_int4((int32_t)top[g].v[0], (int32_t)top[g].v[1], (int32_t)top[g].v[2], (int32_t)top[g].v[3]);
                    } else {
                        uint4 *dst = reinterpret_cast<uint4 *>(static_cast<uint64_t *>(out_values) + rg);
                        dst[0] = make_uint4((uint32_t)top[g].v[0], (uint32_t)(top[g].v[0] >> 32), (uint32_t)top[g].v[1], (uint32_t)(top[g].v[1] >> 32));
                        dst[1] = make_uint4((uint32_t)top[g].v[2], (uint32_t)(top[g].v[2] >> 32), (uint32_t)top[g].v[3], (uint32_t)(top[g].v[3] >> 32));
                    }
                    if (out_valid) *reinterpret_cast<uint32_t *>(out_valid + rg) = (top[g].ok & 1u) | ((top[g].ok & 2u) << 7) | ((top[g].ok & 4u) << 14) | ((top[g].ok & 8u) << 21);
                    else if (top[g].ok != 15u) bad |= kErrNull;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t r = rg + j;
                        if (r >= n) continue;
                        const bool ok = (top[g].ok >> j) & 1u;
                        if (out_type == (int32_t)ColType::I32) static_cast<int32_t *>(out_values)[r] = (int32_t)top[g].v[j];
                        else static_cast<uint64_t *>(out_values)[r] = top[g].v[j];
                        if (out_valid) out_valid[r] = ok;
                        else if (!ok) bad |= kErrNull;
                    }
                }
            }
        }
        if (kMask) store_flags_and_counts(flags, tile, flag_words, counts);
    }
    if (bad) atomicOr(err, bad);
}

size_t stack_bytes(const ValProgram &prog);
// groups per pass: four (two) while the operand stack below the top fits half the 64 KB a launch gets without asking
int groups_of(const ValProgram &prog) {
    static const int forced = exp_env("FLOCKGPU_VALPROG_GROUPS") ? atoi(exp_env("FLOCKGPU_VALPROG_GROUPS")) : 0;   // (A/B knob)
    const int g = forced ? forced : kValGroups;
    return prog.max_stack <= 2 ? g : prog.max_stack <= 3 && g > 2 ? 2 : 1;
}
template <bool kMask>
void launch(flockgpu_ctx *ctx, const ValProgram &prog, unsigned grid, int64_t rows, int32_t n_tiles, void *out_values, uint8_t *out_valid, int32_t out_type, uint32_t *flags,
            uint32_t *counts, uint32_t *d_err) {
    const int g = groups_of(prog);
    const size_t lds = stack_bytes(prog);
    if (g == 4) hipLaunchKernelGGL((valprog_kernel<kMask, 4>), dim3(grid), dim3(kBlock), lds, ctx->stream, prog, rows, n_tiles, out_values, out_valid, out_type, flags, counts, d_err);
    else if (g == 2) hipLaunchKernelGGL((valprog_kernel<kMask, 2>), dim3(grid), dim3(kBlock), lds, ctx->stream, prog, rows, n_tiles, out_values, out_valid, out_type, flags, counts, d_err);
    else hipLaunchKernelGGL((valprog_kernel<kMask, 1>), dim3(grid), dim3(kBlock), lds, ctx->stream, prog, rows, n_tiles, out_values, out_valid, out_type, flags, counts, d_err);
}
size_t stack_bytes(const ValProgram &prog) {
    const size_t below = prog.max_stack > 1 ? (size_t)prog.max_stack - 1 : 0;
    return below * (size_t)groups_of(prog) * (4 * kBlock * sizeof(uint64_t) + kBlock) + 16;
}

int report(flockgpu_ctx *ctx, const char *name, uint32_t e) {
    if (e & kErrDivZero) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: division by zero", name);
    if (e & kErrOverflow) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: INT_MIN / -1 overflows (the reference's arithmetic panics there)", name);
    if (e & kErrCast) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: a value does not fit the type it is cast to", name);
    if (e & kErrNull) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: a NULL result where the expression was taken not to produce one", name);
    return FLOCKGPU_OK;
}

}  // namespace

namespace flockgpu {

int valprog_to_column(flockgpu_ctx *ctx, const char *name, const ValProgram &prog, int64_t rows, ColType out_type, void *out_values, uint8_t *out_valid) {
    if (out_type == ColType::UTF8) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: a computed Utf8 column", name);
    if (prog.n_ops < 1 || prog.max_stack > kValMaxStack) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: malformed expression program", name);
    if (rows <= 0) return FLOCKGPU_OK;
    if (rows >= (int64_t(1) << 44)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: too many rows", name);
    const std::string base = name;
    uint32_t *d_err = nullptr, *h_err = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".err").c_str(), 4, &d_err));
    FG_TRY(pinned_get_t(ctx, (base + ".err").c_str(), 4, &h_err));
    FG_TRY(fill_words(ctx, FillList().add(d_err, 0u, 1)));
    const int32_t n_tiles = (int32_t)div_up(rows, (int64_t)kFlagTile);
    const unsigned grid = (unsigned)std::min<int64_t>(n_tiles, (int64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "valprog_kernel");
        launch<false>(ctx, prog, grid, rows, n_tiles, out_values, out_valid, (int32_t)out_type, nullptr, nullptr, d_err);
    }
    FG_TRY(check_launch(ctx, "valprog_kernel"));
    pinned_pending32(h_err, 1);
    FG_TRY(publish_words(ctx, PublishList().add(h_err, d_err, 1)));
    FG_TRY(wait_pinned32(ctx, h_err, 1));
    return report(ctx, name, *h_err);
}

int valprog_to_rows(flockgpu_ctx *ctx, const char *name, const ValProgram &prog, int64_t rows, int32_t **out_rows, int64_t *n_out) {
    const std::string base = name;
    int32_t *o_rows = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".rows").c_str(), (size_t)std::max<int64_t>(rows, 0) + 4, &o_rows));
    *out_rows = o_rows;
    *n_out = 0;
    if (prog.n_ops < 1 || prog.max_stack > kValMaxStack) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: malformed expression program", name);
    if (rows <= 0) return FLOCKGPU_OK;
    if (rows >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: more than 2^31 rows", name);
    int64_t sb = 0, se = rows;
    SegTiles st;
    FG_TRY(build_seg_tiles(ctx, (base + ".tiles").c_str(), &sb, &se, 1, kFlagTile, &st));   // (one segment from row 0: tile t = rows [8192 t, 8192 (t + 1)), as the kernel walks them)
    uint32_t *flags = nullptr, *counts = nullptr, *d_err = nullptr, *h_err = nullptr;
    uint64_t *tile_base = nullptr;
    int64_t *h_off = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".flags").c_str(), (size_t)st.n_tiles * kBlock + 4, &flags));
    FG_TRY(arena_get_t(ctx, (base + ".counts").c_str(), (size_t)st.n_tiles * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, (base + ".base").c_str(), (size_t)st.n_tiles + 1, &tile_base));
    FG_TRY(arena_get_t(ctx, (base + ".err").c_str(), 4, &d_err));
    FG_TRY(pinned_get_t(ctx, (base + ".err").c_str(), 4, &h_err));
    FG_TRY(pinned_get_t(ctx, (base + ".off").c_str(), 2, &h_off));
    pinned_pending(reinterpret_cast<uint64_t *>(h_off), 2);   // (wait_pinned below)
    FG_TRY(fill_words(ctx, FillList().add(d_err, 0u, 1)));
    const unsigned grid = (unsigned)std::min<int64_t>(st.n_tiles, (int64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "valprog_kernel");
        launch<true>(ctx, prog, grid, rows, st.n_tiles, nullptr, nullptr, 0, flags, counts, d_err);
    }
    FG_TRY(check_launch(ctx, "valprog_kernel"));
    pinned_pending32(h_err, 1);
    FG_TRY(publish_words(ctx, PublishList().add(h_err, d_err, 1)));
    if (st.n_tiles <= 2048) {
        FG_TRY(emit_flagged_rows_self(ctx, st, flags, counts, o_rows, h_off));
    } else {
        FG_TRY(launch_tile_scan(ctx, counts, st.n_tiles, tile_base, st.tile_first, st.n_seg, h_off));
        FG_TRY(emit_flagged_rows(ctx, st, flags, counts, tile_base, o_rows));
    }
    FG_TRY(wait_pinned(ctx, reinterpret_cast<const uint64_t *>(h_off), 2));
    FG_TRY(wait_pinned32(ctx, h_err, 1));   // (published before the emit was queued: there by now)
    FG_TRY(report(ctx, name, *h_err));
    *n_out = h_off[1];
    return FLOCKGPU_OK;
}

}  // namespace flockgpu
