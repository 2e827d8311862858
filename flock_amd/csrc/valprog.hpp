// Computed expressions of ProjectionExec / FilterExec in general (valprog.hip): what neither the column pass-through, q1's
// `literal * Int32 column` kernel nor the one-pass predicate program (pred.hpp) covers -- arithmetic over columns (+ - * / %, unary -),
// CAST / TRY_CAST between the numeric types, comparisons of computed values, CASE WHEN ... THEN ... ELSE ... END -- runs as ONE kernel
// per expression: the tree is flattened on the host into a postfix program (passed by value) and interpreted in the FLAG-TILE geometry
// of the selecting kernels (scan.hpp: 8192-row tiles, a lane's 32 rows = eight 16-byte loads per Int32 column).  Round 6: a pass of
// the program evaluates FOUR rows per lane (one 16-byte load per column operand, one scalar dispatch per operator and four rows), the top
// of the operand stack lives in registers (LDS only holds what waits below it), an operator whose right operand is a literal takes it as
// an immediate (no push, no pop), and division / remainder by a literal is a multiply-high by a host-made reciprocal (a 64-bit division is
// ~150 instructions per row on gfx950).  The filter form writes the flag words and wave counts the row-emitting machinery reads
// (no byte mask, no pass over one): profiles/r06/expr_*.
//
// Semantics restated from upstream DataFusion ~6 / arrow-rs ~8 (the fork's expressions/*.rs are not in the
// reference tree: ASSUMPTIONS, not pinned by reference-held vectors -- tests/test_plan_round5b.py checks them against the oracle's
// twin, oracle/generic_ops.py: eval_typed, and q1's reference-held projection stays on its own kernel):
//   A-V1 both operands of a binary operator have ONE type (the planner inserts the casts); an untyped literal takes the other side's;
//   A-V2 + - * and unary - on integers wrap at the type's width (arrow's unchecked kernels `add` / `subtract` / `multiply`); Float64 + - *
//        are IEEE (no contraction: -ffp-contract=off);
//   A-V3 / and %: arrow-rs `divide` / `modulus` = `math_checked_divide_op` (and `divide_scalar` / `modulus_scalar`, and the `simd` feature's
//        `simd_checked_divide` the fork enables, flock/Cargo.toml:12) test `is_zero()` on the divisor of every VALID row for EVERY native
//        type: a zero divisor -- integer 0, Float64 0.0 or -0.0 -- fails the whole call with ArrowError::DivideByZero.  (Arrow C++ /
//        pyarrow return +-inf / NaN for floats: a difference between the two Arrows, the fork runs arrow-rs.)  A NULL operand makes the
//        row NULL before the divisor is looked at.  Integer / and % truncate towards zero;
//   A-V4 INT_MIN / -1 and INT_MIN % -1 (Int32 and Int64): Rust's `/` and `%` panic ("attempt to divide with overflow"), and the fork's
//        release profile aborts on panic (Cargo.toml:27): the invocation dies.  Here: the call fails with FLOCKGPU_ERR_INVALID;
//   A-V5 CAST fails the call when a valid value does not fit the target (DataFusion casts with safe = false), TRY_CAST yields NULL;
//        Float64 -> integer truncates towards zero, NaN does not fit; integer -> Float64 rounds to nearest even;
//   A-V6 comparisons yield NULL when an operand is NULL; AND / OR / NOT are Kleene; IS [NOT] NULL never yields NULL;
//   A-V7 CASE evaluates every branch for every row (as the fork's CaseExpr does over the whole batch) and picks the first WHEN that is
//        TRUE, else ELSE, else NULL.
#pragma once
#include "relops.hpp"
#include "divmagic.hpp"

namespace flockgpu {

constexpr int kValMaxOps = 96;
constexpr int kValMaxCols = 8;
constexpr int kValMaxConsts = 32;
constexpr int kValMaxStack = 8;

enum class ValType : uint8_t { I32 = 0, I64 = 1, U64 = 2, F64 = 3, BOOL = 5, NONE = 255 };   // (0..3 = ColType)
enum class ValOpKind : uint8_t {
    Col = 0, Const, Null,                         // push
    Add, Sub, Mul, Div, Mod, Neg,                 // arithmetic in `type`
    Cast, TryCast,                                // `type` -> `to`
    Eq, Ne, Lt, Le, Gt, Ge,                       // compare in `type` -> BOOL
    And, Or, Not, IsNull, IsNotNull,              // BOOL (IsNull: any type)
    Select                                        // pops THEN, WHEN, ELSE (pushed in the order ELSE, WHEN, THEN)
};
struct ValOp {
    uint8_t kind, type, to, arg;   // ValOpKind; operand ValType; Cast target -- or, on a binary operator, kValImm: the right operand is consts[arg]
};                                 // (Div / Mod with kValImm: consts[arg] = divisor, [arg + 1] = reciprocal, [arg + 2] = shift | add << 8, [arg + 3]: see kValImm32; ValBuilder::fuse_immediate)
constexpr uint8_t kValImm = 0x80;
constexpr uint8_t kValImm32 = 0x81;   // Div / Mod by a literal |c| < 2^32 whose dividend is KNOWN to fit Int32 (ValBuilder::narrow): consts[arg + 3] = the 32-bit reciprocal
struct ValCol {
    const void *values;
    const uint8_t *valid;
    int32_t type;   // ColType
    int32_t pad;
};
struct ValProgram {
    ValCol cols[kValMaxCols];
    uint64_t consts[kValMaxConsts + 3];   // (+ 3: the interpreter fetches consts[arg .. arg + 3] for every operator ahead of its turn)
    ValOp ops[kValMaxOps];
    int32_t n_cols = 0, n_consts = 0, n_ops = 0, max_stack = 0;
};

// Host-side assembly (plan.hip walks the expression tree).  Every add returns false when the program is full.
struct ValBuilder {
    ValProgram p{};
    int depth = 0;
    uint32_t narrow = 0;   // bit i: the value in stack slot i is known to lie in [-2^31, 2^31) whatever its type (an Int32 column behind the planner's CAST to Int64, a small literal, a remainder of such)
    int add_col(const DevColumn &c) {
        for (int i = 0; i < p.n_cols; ++i)
            if (p.cols[i].values == c.values && p.cols[i].valid == c.valid) return i;
        if (p.n_cols >= kValMaxCols) return -1;
        p.cols[p.n_cols] = ValCol{c.values, c.valid, (int32_t)c.type, 0};
        return p.n_cols++;
    }
    int add_const(uint64_t bits) {
        for (int i = 0; i < p.n_consts; ++i)
            if (p.consts[i] == bits) return i;
        if (p.n_consts >= kValMaxConsts) return -1;
        p.consts[p.n_consts] = bits;
        return p.n_consts++;
    }
    // pops: operands the operator takes off the stack; every operator pushes one result
    bool push(ValOpKind k, ValType type, int pops, int arg = 0, ValType to = ValType::NONE) {
        if (p.n_ops >= kValMaxOps || arg < 0) return false;
        const bool fits = result_fits_i32(k, type, pops, arg, to);   // (looks at the operands: before anything moves)
        if (pops == 2 && fuse_immediate(k, type)) {
            set_narrow(depth - 1, fits);
            return true;
        }
        p.ops[p.n_ops++] = ValOp{(uint8_t)k, (uint8_t)type, (uint8_t)to, (uint8_t)arg};
        depth += 1 - pops;
        if (depth > p.max_stack) p.max_stack = depth;
        if (depth >= 1 && depth <= kValMaxStack) set_narrow(depth - 1, fits);
        return depth >= 1 && depth <= kValMaxStack;
    }
    void set_narrow(int slot, bool v) { narrow = v ? narrow | (1u << slot) : narrow & ~(1u << slot); }
    bool is_narrow(int slot) const { return slot >= 0 && ((narrow >> slot) & 1u); }
    // whether the operator's result is known to lie in the Int32 range (operands: slots depth - pops .. depth - 1)
    bool result_fits_i32(ValOpKind k, ValType type, int pops, int arg, ValType to) const {
        switch (k) {
            case ValOpKind::Col: return p.cols[arg].type == (int32_t)ColType::I32;
            case ValOpKind::Const: return type != ValType::F64 && (int64_t)p.consts[arg] >= INT32_MIN && (int64_t)p.consts[arg] <= INT32_MAX && !(type == ValType::U64 && (int64_t)p.consts[arg] < 0);
            case ValOpKind::Null: return true;
            case ValOpKind::Add: case ValOpKind::Sub: case ValOpKind::Mul: case ValOpKind::Neg: case ValOpKind::Div: return type == ValType::I32;   // (an Int32 result is wrapped to its width)
            case ValOpKind::Mod: return type == ValType::I32 || (type != ValType::F64 && is_narrow(depth - 2));   // |x % y| <= |x|
            case ValOpKind::Cast: case ValOpKind::TryCast: return to == ValType::I32 || (type != ValType::F64 && to != ValType::F64 && is_narrow(depth - 1) && !(to == ValType::U64));
            case ValOpKind::Select: return is_narrow(depth - 1) && is_narrow(depth - 3);   // THEN and ELSE
            default: return true;   // comparisons and logic: 0 / 1
        }
    }
    // A binary operator whose RIGHT operand was just pushed as a literal takes it as an immediate: the Const push goes, the operator
    // reads consts[arg].  Division / remainder by a non-zero integer literal get the reciprocal of |divisor| next to it (Granlund /
    // Montgomery, round-up variant, 64-bit: q = mulhi(n, m) >> s, or the 65-bit multiplier form ((n - hi) >> 1) + hi) >> s).
    bool fuse_immediate(ValOpKind k, ValType type) {
        if (p.n_ops < 1 || p.ops[p.n_ops - 1].kind != (uint8_t)ValOpKind::Const) return false;
        const bool arith = k == ValOpKind::Add || k == ValOpKind::Sub || k == ValOpKind::Mul || k == ValOpKind::Div || k == ValOpKind::Mod;
        const bool cmp = k == ValOpKind::Eq || k == ValOpKind::Ne || k == ValOpKind::Lt || k == ValOpKind::Le || k == ValOpKind::Gt || k == ValOpKind::Ge;
        if (!arith && !cmp) return false;
        int arg = p.ops[p.n_ops - 1].arg;
        uint8_t imm_kind = kValImm;
        const uint64_t c = p.consts[arg];
        if ((k == ValOpKind::Div || k == ValOpKind::Mod) && type != ValType::F64) {
            if (c == 0) return false;   // (a zero divisor: the generic operator reports it -- for the valid rows only)
            uint64_t d = c;
            if (type != ValType::U64 && (int64_t)c < 0) d = 0 - c;   // |divisor| (2^63 for INT64_MIN)
            uint64_t magic = 0, shift = 63, add = 0;
            while (!((d >> shift) & 1u)) --shift;   // floor(log2 d)
            if (d & (d - 1)) {
                const unsigned __int128 two = (unsigned __int128)1 << (64 + shift);
                unsigned __int128 pm = two / d;
                const unsigned __int128 rem = two - pm * d;
                if ((unsigned __int128)d - rem >= ((unsigned __int128)1 << shift)) {   // one more bit of multiplier
                    pm = pm * 2 + (rem * 2 >= d ? 1 : 0);
                    add = 1;
                }
                magic = (uint64_t)(pm + 1);
            }
            if (p.n_consts + 4 > kValMaxConsts) return false;
            arg = p.n_consts;
            p.consts[p.n_consts++] = c;
            p.consts[p.n_consts++] = magic;
            p.consts[p.n_consts++] = shift | (add << 8);
            // a dividend known to fit Int32 (the slot below the literal's) and |divisor| < 2^32: ONE 32 x 32 -> high-32 multiply per row
            // (divmagic.hpp) instead of the four of the 64-bit form
            uint64_t r32 = 0;
            if (d < (uint64_t(1) << 32) && is_narrow(depth - 2)) {
                const UMod32 m = umod32_make((uint32_t)d);
                r32 = (uint64_t)m.magic | ((uint64_t)m.shift << 32) | ((uint64_t)m.add << 40);
                imm_kind = kValImm32;
            }
            p.consts[p.n_consts++] = r32;
        }
        p.ops[p.n_ops - 1] = ValOp{(uint8_t)k, (uint8_t)type, imm_kind, (uint8_t)arg};
        depth -= 1;   // (the literal's push is taken back; the operator replaces its left operand in place)
        return true;
    }
};

// out_type: the value column's type (I32 / I64 / U64 / F64).  out_valid (may be null when the caller knows the result holds no NULL:
// then a NULL result is an error) receives one byte per row.  Error codes come back as FLOCKGPU_ERR_INVALID with the reason
// (division by zero / INT_MIN / -1 / a value that does not fit its CAST).  One host wait.
int valprog_to_column(flockgpu_ctx *ctx, const char *name, const ValProgram &prog, int64_t rows, ColType out_type, void *out_values, uint8_t *out_valid);
// FilterExec: the rows where the BOOL result is TRUE (FALSE and NULL: dropped), in order -- flag words + wave counts from the evaluating
// kernel itself, then the scan / emit of pred_to_rows.  One host wait (row count + error word).
int valprog_to_rows(flockgpu_ctx *ctx, const char *name, const ValProgram &prog, int64_t rows, int32_t **out_rows, int64_t *n_out);

}  // namespace flockgpu
