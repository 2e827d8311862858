// The general expression evaluator of the plan layer: a postfix program per expression, interpreted per row (valprog.hpp).
#include "valprog.hpp"

using namespace flockgpu;

namespace {

constexpr uint32_t kErrDivZero = 1u, kErrCast = 2u, kErrNull = 4u;

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

__device__ __forceinline__ uint64_t arith(uint8_t kind, uint8_t type, uint64_t a, uint64_t b, uint32_t *bad) {
    const uint8_t ADD = (uint8_t)ValOpKind::Add, SUB = (uint8_t)ValOpKind::Sub, MUL = (uint8_t)ValOpKind::Mul, DIV = (uint8_t)ValOpKind::Div;
    if (type == (uint8_t)ValType::F64) {
        const double x = as_f64(a), y = as_f64(b);
        return f64_bits(kind == ADD ? x + y : kind == SUB ? x - y : kind == MUL ? x * y : kind == DIV ? x / y : fmod(x, y));
    }
    if (kind == ADD || kind == SUB || kind == MUL) {
        const uint64_t r = kind == ADD ? a + b : kind == SUB ? a - b : a * b;
        return type == (uint8_t)ValType::I32 ? (uint64_t)(int64_t)(int32_t)(uint32_t)r : r;   // wraps at the type's width
    }
    if (b == 0) {
        *bad |= kErrDivZero;
        return 0;
    }
    if (type == (uint8_t)ValType::U64) return kind == DIV ? a / b : a % b;
    const int64_t x = (int64_t)a, y = (int64_t)b;
    if (y == -1) {   // (INT_MIN / -1 wraps; x % -1 is 0)
        if (kind != DIV) return 0;
        const uint64_t r = 0 - a;
        return type == (uint8_t)ValType::I32 ? (uint64_t)(int64_t)(int32_t)(uint32_t)r : r;
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

// kMask: out_values is the byte mask (1 = TRUE); else a value column of `out_type` + out_valid (null: a NULL result is an error)
template <bool kMask>
__global__ __launch_bounds__(kBlock) void valprog_kernel(ValProgram p, int64_t n, void *__restrict__ out_values, uint8_t *__restrict__ out_valid,
                                                         int32_t out_type, uint32_t *err) {
    __shared__ uint64_t s_v[kValMaxStack][kBlock];
    __shared__ uint8_t s_ok[kValMaxStack][kBlock];
    const int t = threadIdx.x;
    uint32_t bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + t; i < n; i += (int64_t)gridDim.x * kBlock) {
        int sp = 0;
        for (int o = 0; o < p.n_ops; ++o) {
            const ValOp op = p.ops[o];
            switch ((ValOpKind)op.kind) {
                case ValOpKind::Col: {
                    const ValCol &c = p.cols[op.arg];
                    const bool ok = !c.valid || c.valid[i];
                    s_v[sp][t] = ok ? load_value(c, i) : 0;   // (the slot of a NULL holds an unspecified value)
                    s_ok[sp][t] = ok;
                    ++sp;
                    break;
                }
                case ValOpKind::Const:
                    s_v[sp][t] = p.consts[op.arg];
                    s_ok[sp][t] = 1;
                    ++sp;
                    break;
                case ValOpKind::Null:
                    s_v[sp][t] = 0;
                    s_ok[sp][t] = 0;
                    ++sp;
                    break;
                case ValOpKind::Add: case ValOpKind::Sub: case ValOpKind::Mul: case ValOpKind::Div: case ValOpKind::Mod: {
                    --sp;
                    const bool ok = s_ok[sp - 1][t] && s_ok[sp][t];
                    s_v[sp - 1][t] = ok ? arith(op.kind, op.type, s_v[sp - 1][t], s_v[sp][t], &bad) : 0;
                    s_ok[sp - 1][t] = ok;
                    break;
                }
                case ValOpKind::Neg: {
                    const uint64_t a = s_v[sp - 1][t];
                    s_v[sp - 1][t] = op.type == (uint8_t)ValType::F64 ? f64_bits(-as_f64(a))
                                     : op.type == (uint8_t)ValType::I32 ? (uint64_t)(int64_t)(int32_t)(0u - (uint32_t)a) : 0 - a;
                    break;
                }
                case ValOpKind::Cast: case ValOpKind::TryCast: {
                    if (s_ok[sp - 1][t]) {
                        uint64_t r = 0;
                        if (cast_value(s_v[sp - 1][t], op.type, op.to, &r)) {
                            s_v[sp - 1][t] = r;
                        } else {
                            s_v[sp - 1][t] = 0;
                            s_ok[sp - 1][t] = 0;
                            if ((ValOpKind)op.kind == ValOpKind::Cast) bad |= kErrCast;
                        }
                    }
                    break;
                }
                case ValOpKind::Eq: case ValOpKind::Ne: case ValOpKind::Lt: case ValOpKind::Le: case ValOpKind::Gt: case ValOpKind::Ge: {
                    --sp;
                    const bool ok = s_ok[sp - 1][t] && s_ok[sp][t];
                    s_v[sp - 1][t] = ok && compare(op.kind, op.type, s_v[sp - 1][t], s_v[sp][t]) ? 1 : 0;
                    s_ok[sp - 1][t] = ok;
                    break;
                }
                case ValOpKind::And: case ValOpKind::Or: {
                    --sp;
                    const bool oa = s_ok[sp - 1][t], ob = s_ok[sp][t], va = oa && s_v[sp - 1][t], vb = ob && s_v[sp][t];
                    bool v, ok;
                    if ((ValOpKind)op.kind == ValOpKind::And) {
                        const bool is_false = (oa && !va) || (ob && !vb);
                        ok = is_false || (oa && ob);
                        v = !is_false && oa && ob;
                    } else {
                        const bool is_true = va || vb;
                        ok = is_true || (oa && ob);
                        v = is_true;
                    }
                    s_v[sp - 1][t] = v ? 1 : 0;
                    s_ok[sp - 1][t] = ok;
                    break;
                }
                case ValOpKind::Not:
                    s_v[sp - 1][t] = s_ok[sp - 1][t] && !s_v[sp - 1][t] ? 1 : 0;
                    break;
                case ValOpKind::IsNull: case ValOpKind::IsNotNull:
                    s_v[sp - 1][t] = (s_ok[sp - 1][t] != 0) == ((ValOpKind)op.kind == ValOpKind::IsNotNull) ? 1 : 0;
                    s_ok[sp - 1][t] = 1;
                    break;
                case ValOpKind::Select: {   // [.. ELSE WHEN THEN]
                    sp -= 2;
                    const bool take = s_ok[sp][t] && s_v[sp][t];
                    if (take) {
                        s_v[sp - 1][t] = s_v[sp + 1][t];
                        s_ok[sp - 1][t] = s_ok[sp + 1][t];
                    }
                    break;
                }
            }
        }
        const bool ok = s_ok[0][t];
        const uint64_t v = ok ? s_v[0][t] : 0;
        if (kMask) {
            static_cast<uint8_t *>(out_values)[i] = ok && v ? 1 : 0;
        } else {
            if (out_type == (int32_t)ColType::I32) static_cast<int32_t *>(out_values)[i] = (int32_t)v;
            else static_cast<uint64_t *>(out_values)[i] = v;
            if (out_valid) out_valid[i] = ok;
            else if (!ok) bad |= kErrNull;
        }
    }
    if (bad) atomicOr(err, bad);
}

int run(flockgpu_ctx *ctx, const char *name, const ValProgram &prog, int64_t rows, bool mask, ColType out_type, void *out_values, uint8_t *out_valid) {
    if (prog.n_ops < 1 || prog.max_stack > kValMaxStack) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: malformed expression program", name);
    if (rows <= 0) return FLOCKGPU_OK;
    const std::string base = name;
    uint32_t *d_err = nullptr, *h_err = nullptr;
    FG_TRY(arena_get_t(ctx, (base + ".err").c_str(), 4, &d_err));
    FG_TRY(pinned_get_t(ctx, (base + ".err").c_str(), 4, &h_err));
    FG_TRY(fill_words(ctx, FillList().add(d_err, 0u, 1)));
    const unsigned grid = (unsigned)std::min<int64_t>(div_up(rows, kBlock), (int64_t)ctx->num_cus * 8);
    {
        LaunchScope ls(ctx, "valprog_kernel");
        if (mask) hipLaunchKernelGGL(valprog_kernel<true>, dim3(grid), dim3(kBlock), 0, ctx->stream, prog, rows, out_values, (uint8_t *)nullptr, 0, d_err);
        else hipLaunchKernelGGL(valprog_kernel<false>, dim3(grid), dim3(kBlock), 0, ctx->stream, prog, rows, out_values, out_valid, (int32_t)out_type, d_err);
    }
    FG_TRY(check_launch(ctx, "valprog_kernel"));
    pinned_pending32(h_err, 1);
    FG_TRY(publish_words(ctx, PublishList().add(h_err, d_err, 1)));
    FG_TRY(wait_pinned32(ctx, h_err, 1));
    if (*h_err & kErrDivZero) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: division by zero", name);
    if (*h_err & kErrCast) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: a value does not fit the type it is cast to", name);
    if (*h_err & kErrNull) return fail(ctx, FLOCKGPU_ERR_INVALID, "%s: a NULL result where the expression was taken not to produce one", name);
    return FLOCKGPU_OK;
}

}  // namespace

namespace flockgpu {

int valprog_to_column(flockgpu_ctx *ctx, const char *name, const ValProgram &prog, int64_t rows, ColType out_type, void *out_values, uint8_t *out_valid) {
    if (out_type == ColType::UTF8) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "%s: a computed Utf8 column", name);
    return run(ctx, name, prog, rows, false, out_type, out_values, out_valid);
}

int valprog_to_mask(flockgpu_ctx *ctx, const char *name, const ValProgram &prog, int64_t rows, uint8_t *mask) {
    return run(ctx, name, prog, rows, true, ColType::I32, mask, nullptr);
}

}  // namespace flockgpu
