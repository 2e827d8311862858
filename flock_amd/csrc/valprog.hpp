// Computed expressions of ProjectionExec / FilterExec in general (valprog.hip): what neither the column pass-through, q1's
// `literal * Int32 column` kernel nor the one-pass predicate program (pred.hpp) covers -- arithmetic over columns (+ - * / %, unary -),
// CAST / TRY_CAST between the numeric types, comparisons of computed values, CASE WHEN ... THEN ... ELSE ... END -- runs as ONE kernel
// per expression: the tree is flattened on the host into a postfix program (passed by value), every thread walks it for its rows
// with an operand stack in LDS (a column of (value, valid) slots per thread: no bank conflicts, no barrier), the result is a value
// column + validity bytes (projection) or a byte mask (filter: 1 where the predicate is TRUE).  An interpreter, not a code
// generator: ~25 instructions per operator and row, i.e. HBM-bound up to a handful of operators and issue-bound beyond -- the fast
// paths stay in front of it.
//
// Semantics restated from upstream DataFusion ~6 / arrow-rs 6 (the fork's expressions/*.rs are not in the
// reference tree: assumptions, not pinned by reference-held vectors -- tests/test_plan_round5b.py checks them against the oracle's
// twin, oracle/generic_ops.py: eval_physical_expr, and q1's reference-held projection stays on its own kernel):
//   * both operands of a binary operator have ONE type (the planner inserts the casts); an untyped literal takes the other side's;
//   * + - * and unary - on integers wrap at the type's width (arrow's unchecked kernels); Float64 is IEEE (no contraction: -ffp-contract=off);
//   * integer / and % truncate towards zero; a zero divisor in a VALID row is an error for the whole call (ArrowError::DivideByZero),
//     a NULL operand makes the row NULL before the divisor is looked at; INT_MIN / -1 wraps (arrow-rs would panic);
//   * CAST fails the call when a valid value does not fit the target (DataFusion casts with safe = false), TRY_CAST yields NULL;
//     Float64 -> integer truncates towards zero, NaN does not fit; integer -> Float64 rounds to nearest even;
//   * comparisons yield NULL when an operand is NULL; AND / OR / NOT are Kleene; IS [NOT] NULL never yields NULL;
//   * CASE evaluates every branch for every row (as the fork's CaseExpr does over the whole batch) and picks the first WHEN that is
//     TRUE, else ELSE, else NULL.
#pragma once
#include "relops.hpp"

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
    uint8_t kind, type, to, arg;   // ValOpKind; operand ValType; Cast target; Col / Const index
};
struct ValCol {
    const void *values;
    const uint8_t *valid;
    int32_t type;   // ColType
    int32_t pad;
};
struct ValProgram {
    ValCol cols[kValMaxCols];
    uint64_t consts[kValMaxConsts];
    ValOp ops[kValMaxOps];
    int32_t n_cols = 0, n_consts = 0, n_ops = 0, max_stack = 0;
};

// Host-side assembly (plan.hip walks the expression tree).  Every add returns false when the program is full.
struct ValBuilder {
    ValProgram p{};
    int depth = 0;
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
        p.ops[p.n_ops++] = ValOp{(uint8_t)k, (uint8_t)type, (uint8_t)to, (uint8_t)arg};
        depth += 1 - pops;
        if (depth > p.max_stack) p.max_stack = depth;
        return depth >= 1 && depth <= kValMaxStack;
    }
};

// out_type: the value column's type (I32 / I64 / U64 / F64).  out_valid (may be null when the caller knows the result holds no NULL:
// then a NULL result is an error) receives one byte per row.  Error codes come back as FLOCKGPU_ERR_INVALID with the reason
// (division by zero / a value that does not fit its CAST).  One host wait.
int valprog_to_column(flockgpu_ctx *ctx, const char *name, const ValProgram &prog, int64_t rows, ColType out_type, void *out_values, uint8_t *out_valid);
// mask[i] = 1 where the BOOL result is TRUE (FALSE and NULL: 0).  One host wait (the error word).
int valprog_to_mask(flockgpu_ctx *ctx, const char *name, const ValProgram &prog, int64_t rows, uint8_t *mask);

}  // namespace flockgpu
