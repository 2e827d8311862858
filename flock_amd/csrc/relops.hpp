// Generic relational operators over device columns, used by the plan interpreter (plan.hip) for every plan node that
// is not covered by one of the fused NEXMark pipelines: the STAGE plans either side of a `RepartitionExec Hash`
// (flock/src/distributed_plan/planner.rs:152-171, playground/.../nexmark/q{3,5,8}.dag; split rule
// flock/src/distributed_plan/stage.rs:269-367).  These run on the small, already filtered / aggregated relations of a
// stage, so they are written for generality (one lane per row, 64-bit normalised keys), not for the HBM roofline; the
// raw-row passes keep their fused kernels (q2 / q3 / q5 / q8 .hip).
//
// Semantics restated from upstream DataFusion ~6.x (not readable in the reference tree):
//   FilterExec keeps input order; HashAggregateExec groups compare by value (Utf8 bytewise); COUNT -> UInt64;
//   MAX ignores NULLs, an empty input gives NULL; inner HashJoinExec emits every matching pair, NULL keys never match.
#pragma once
#include <string>

#include "gather.hpp"
#include "sort.hpp"

namespace flockgpu {

enum class ColType : int32_t { I32 = 0, I64 = 1, U64 = 2, F64 = 3, UTF8 = 4 };
inline size_t col_width(ColType t) { return t == ColType::I32 ? 4 : 8; }

struct DevColumn {
    ColType type = ColType::I32;
    bool is_ts = false;     // Timestamp(Millisecond): Int64 storage, exported as "tsm:"
    bool nullable = false;  // schema flag only
    bool all_null = false;  // the single row of a global aggregate over no input (MAX -> NULL)
    // NULLs (round 4): one byte per row, 1 = valid; nullptr = every row valid (all NEXMark fields; the fused pipelines take only such
    // columns).  The slot of a NULL holds an unspecified value (a Utf8 NULL an empty or arbitrary range): every consumer looks here first.
    const uint8_t *valid = nullptr;
    const void *values = nullptr;      // fixed width values, or Utf8 bytes
    const int32_t *offsets = nullptr;  // Utf8 only, rows + 1 entries
    int64_t bytes = 0;                 // Utf8 only
};

enum class CmpOp : int32_t { EQ = 0, NE = 1, LT = 2, LE = 3, GT = 4, GE = 5 };
enum class AggKind : int32_t { NONE = 0, SUM = 1, MAX = 2 };

// ROW_NUMBER() of WindowAggExec over an input that arrives sorted by (PARTITION BY, ORDER BY): out[i] = 1 + i - (first row of the RUN of equal
// partition keys row i lies in); a run ends where any of the up to four key columns changes (NULL equals NULL; Float64 by its bits; Utf8 keys are
// refused).  No key column: the whole input is one run.  Launches only -- the caller's next wait covers it.
int row_number_runs(flockgpu_ctx *ctx, const char *name, const DevColumn *cols, int n_cols, int64_t rows, uint64_t *out);

// keys of an integer column as int64 (I32 sign-extended; I64 / U64 bit pattern): out[rows]
int widen_to_i64(flockgpu_ctx *ctx, const DevColumn &col, int64_t rows, int64_t *out);

// ---- predicates: pred.hpp (one pass per FilterExec).  A byte mask (1 = keep) -> its rows, for the operators' own bookkeeping:
// rows with mask != 0, in order.  *out_rows: ctx-owned (arena key `name`), n_out through ONE synchronisation.
int mask_to_rows(flockgpu_ctx *ctx, const char *name, const uint8_t *mask, int64_t rows, int32_t **out_rows, int64_t *n_out);

// ---- take (a column's validity bytes are taken along)
int take_column(flockgpu_ctx *ctx, const char *name, const DevColumn &src, const int32_t *rows, int64_t n, DevColumn *out);
int gather_u8(flockgpu_ctx *ctx, const uint8_t *src, const int32_t *rows, int64_t n, uint8_t *out);
// keys[i] = sentinel where valid[i] == 0: NULL group keys form ONE group (DataFusion groups NULLs together), NULL hash-partition keys one place
int replace_invalid_i64(flockgpu_ctx *ctx, int64_t *keys, const uint8_t *valid, int64_t rows, int64_t sentinel);
// out[i] = keys[i] != sentinel (the validity of a group-key column that went through replace_invalid_i64); out[i] = count[i] != 0 (AVG over no valid value)
int valid_from_i64(flockgpu_ctx *ctx, const int64_t *keys, int64_t n, int64_t sentinel, uint8_t *out);

// ---- GROUP BY one integer key (as int64): distinct keys + SUM / MAX of `values` (null: SUM counts rows) per key.
// Outputs are ctx-owned: keys[n_groups], agg[n_groups], first_row[n_groups] (smallest input row of the group).
struct GroupResult {
    int64_t n_groups = 0;
    int64_t *keys = nullptr;
    uint64_t *agg = nullptr;
    int32_t *first_row = nullptr;
};
int group_by_key64(flockgpu_ctx *ctx, const char *name, const int64_t *keys, const uint64_t *values, AggKind kind, int64_t rows,
                   GroupResult *out);
// The same GROUP BY with up to kMaxGroupAggs accumulators per group, all in one table (one pass over the rows).
// MAX_F64 / MIN_F64 order doubles through their order-preserving bit pattern (no NaN among the inputs: Arrow's min / max kernels
// skip NaN only against non-NaN values, a case the engine does not claim).
// Accumulators are 64-bit: COUNT (rows), SUM_INT (two's complement add of an integer column: Int32 sign-extended), MAX / MIN
// signed or unsigned, SUM_F64 (double add: exact -- hence order-free -- while every partial sum is an integer below 2^53, which
// is what AVG's partial sums of integer columns are).
enum class AggOp : int32_t { COUNT = 0, SUM_INT = 1, MAX_S = 2, MAX_U = 3, MIN_S = 4, MIN_U = 5, SUM_F64 = 6, MAX_F64 = 7, MIN_F64 = 8 };
constexpr int kMaxGroupAggs = 4;
struct AggSpec {
    AggOp op = AggOp::COUNT;
    const void *values = nullptr;  // null for COUNT
    ColType type = ColType::I64;   // storage type of `values`
    // validity of the argument (may be null): a NULL contributes to no accumulator -- COUNT(col) counts the valid rows, MIN / MAX / SUM
    // skip it (DataFusion's accumulators); a group none of whose values is valid comes out NULL (agg_valid)
    const uint8_t *valid = nullptr;
};
struct GroupResultN {
    int64_t n_groups = 0;
    int64_t *keys = nullptr;
    uint64_t *agg[kMaxGroupAggs] = {};  // 64-bit patterns: int64 / uint64 / double by AggOp
    uint8_t *agg_valid[kMaxGroupAggs] = {};  // per group: 1 when a valid value reached accumulator a; null when its spec carried no validity
    int32_t *first_row = nullptr;
    uint8_t *key_valid = nullptr;   // per group, when the call was given key validity: 0 for the NULL keys' group
};
// key_valid (may be null): rows whose key is NULL -- whatever their key bits -- form one group of their own
int group_by_key64_n(flockgpu_ctx *ctx, const char *name, const int64_t *keys, int64_t rows, const AggSpec *specs, int n_specs,
                     GroupResultN *out, const uint8_t *key_valid = nullptr);
// ---- dense integer keys: the perfect-hash variants of GROUP BY and JOIN.  Exact minimum / maximum of an integer column (signed order;
// unsigned for UInt64), through ONE synchronisation; `dense_range_ok`: the key range is affordable as a direct-address table for `rows` rows.
int column_minmax(flockgpu_ctx *ctx, const DevColumn &col, int64_t rows, int64_t *mn, int64_t *mx);
bool dense_range_ok(int64_t kmin, int64_t kmax, int64_t rows, bool uns);
// GROUP BY `key` (Int32 / Int64 / UInt64, no NULLs, every key inside [kmin, kmax]) with COUNT / SUM_INT / MIN / MAX accumulators over
// integer columns without NULLs: slot = key - kmin, tiles aggregated in LDS, groups come out in key order.  first_row / validity outputs
// are not produced (nothing on this path needs them).  A key outside [kmin, kmax] voids the call (FLOCKGPU_ERR_INVALID).
int group_by_dense(flockgpu_ctx *ctx, const char *name, const DevColumn &key, int64_t rows, int64_t kmin, int64_t kmax, const AggSpec *specs, int n_specs,
                   GroupResultN *out);
// ---- Utf8 keys (YSB joins and groups on UUID strings, flock/src/distributed_plan/planner.rs:298-346)
// out[i] = 64-bit hash of row i's bytes: equal strings -> equal keys (enough for a hash repartition; NOT an equality test)
int hash_utf8_i64(flockgpu_ctx *ctx, const DevColumn &col, int64_t rows, int64_t *out);
// Exact dictionary codes: codes[i] = row number of the first-inserted row of `build` with the same bytes (a hash table of row
// numbers with a full byte compare on every hit).  probe_codes[j] = the code of the equal `build` string, or -(j + 2) when
// `build` does not contain it (negative codes are pairwise different and match nothing).  Either output may be null.
int utf8_codes(flockgpu_ctx *ctx, const char *name, const DevColumn &build, int64_t n_build, int64_t *build_codes, const DevColumn *probe,
               int64_t n_probe, int64_t *probe_codes);
// (a, b) Int32 pairs <-> one 64-bit key: key = (int64(a) << 32) | uint32(b)
int pack_i32_pair(flockgpu_ctx *ctx, const int32_t *a, const int32_t *b, int64_t n, int64_t *out);
int unpack_i32_pair(flockgpu_ctx *ctx, const int64_t *keys, int64_t n, int32_t *a, int32_t *b);
// out[i] = (double)in[i]
int i64_to_f64(flockgpu_ctx *ctx, const int64_t *in, int64_t n, double *out);
// out[i] = sum[i] / (double)count[i]   (AVG's finish: one IEEE division, as DataFusion's AvgAccumulator::evaluate)
int avg_finish(flockgpu_ctx *ctx, const double *sum, const uint64_t *count, int64_t n, double *out);
// GROUP BY (int32 key, Utf8 value) without aggregates = DISTINCT over both columns: the first row of every distinct pair,
// ascending.
int distinct_i32_utf8(flockgpu_ctx *ctx, const char *name, const int32_t *key, const flockgpu_utf8 &text, int64_t rows,
                      int32_t **out_rows, int64_t *n_out);
// MAX of an integer column (signed, or unsigned for UInt64) as its 64-bit pattern, on the host; *any = 0 when there is no (valid) row.
int reduce_max(flockgpu_ctx *ctx, const DevColumn &col, int64_t rows, int64_t *out, int *any);

// ---- inner equi-join on one 64-bit key: every (left_row, right_row) pair with equal keys, ordered by right row
// (the probe side); the left rows of one right row come in the build chain's order, which is NOT defined (concurrent push-front
// inserts) -- callers compare multisets, as the reference does (test_util.rs:61-90).  Builds on the left (DataFusion's build side).
// More than 2^31 - 1 pairs: FLOCKGPU_ERR_UNSUPPORTED, decided from a 64-bit total before anything is emitted.
int join_key64(flockgpu_ctx *ctx, const char *name, const int64_t *left, int64_t n_left, const int64_t *right, int64_t n_right,
               int32_t **left_rows, int32_t **right_rows, int64_t *n_pairs);
// The table path of join_key64 with the keys read in their columns' own types (Int32 / Int64 / UInt64 of one signedness, no NULLs): 16-byte
// slots (key, claiming row, chain head), unique build keys probed as a filter in the flag-tile geometry.  Builds on the smaller side.
int join_hashed(flockgpu_ctx *ctx, const char *name, const DevColumn &left, int64_t n_left, const DevColumn &right, int64_t n_right,
                int32_t **left_rows, int32_t **right_rows, int64_t *n_pairs);
// join_key64 answers this pair of sizes with ONE workgroup and one launch (table, heads and chain links in LDS): no statistics, no dense path needed
bool join_is_tiny(int64_t n_left, int64_t n_right);

// The same join when the BUILD side's keys are dense (every key of `left` inside [kmin, kmax], range affordable: dense_range_ok): chain heads
// addressed by key - kmin, keys read in their columns' own types.  Pairs ordered by right row, as join_key64's.
int join_dense(flockgpu_ctx *ctx, const char *name, const DevColumn &left, int64_t n_left, int64_t kmin, int64_t kmax, const DevColumn &right, int64_t n_right,
               int32_t **left_rows, int32_t **right_rows, int64_t *n_pairs);

// ---- hash partition: rows grouped by destination (input order kept): dest = (fmix32(fold(key)) * n) >> 32, the
// mix of flockgpu_partition_by_key.  part_offsets: host, n_parts + 1.
int partition_rows_key64(flockgpu_ctx *ctx, const char *name, const int64_t *keys, int64_t rows, int32_t n_parts, int32_t **out_rows,
                         std::vector<int64_t> *part_offsets);

// ---- ORDER BY (SortExec; the reference's boundary goldens end in it, flock/src/runtime/context.rs:471,549, and the stage splitter
// cuts at it, flock/src/distributed_plan/stage.rs:337): the permutation of [0, rows) that orders the rows by `keys`, first key most
// significant, ties in input order (stable).  LSD over the keys: every key becomes an order-preserving unsigned 64-bit value in the
// current order (Int32 / Int64 / Timestamp: sign bit flipped; UInt64: as is; Float64: IEEE total order, NaN last; Utf8: bytewise
// lexicographic -- the length as the least significant sub-key, then 8-byte big-endian chunks from the last to the first;
// descending: complemented), its range is read back (one wait per sub-key; a constant sub-key costs no pass) and the stable radix
// passes of sort.hpp run over the bits the range needs.  *out_rows: ctx-owned, `rows` entries.  Columns hold no NULLs
// (`nulls_first` has nothing to place); an all-NULL column ties every row.
struct SortKey {
    DevColumn col;
    bool descending = false;
    bool nulls_first = false;   // where the rows whose col.valid is 0 go (SortOptions of the plan; DESC does not move them)
};
// sorted_i32 (may be null): set to the first key's column IN ITS SORTED ORDER when the sort made one on the way (one ascending Int32 key without
// NULLs: the radix sort carries the column itself), else to null -- the caller's take of that column is then a pointer.
int sort_rows(flockgpu_ctx *ctx, const char *name, const SortKey *keys, int n_keys, int64_t rows, int32_t **out_rows, const int32_t **sorted_i32 = nullptr);
// keys[order[i]] is non-decreasing in i (sort_rows' output): starts = the positions i where a new key begins (ascending; device), *n_keys of them.
// One host wait.
int key_run_starts(flockgpu_ctx *ctx, const char *name, const int64_t *keys, const int32_t *order, int64_t rows, int32_t **starts, int64_t *n_keys);

// u32 -> u64 (COUNT partial states are UInt64 in the reference's schemas)
int widen_u32_to_u64(flockgpu_ctx *ctx, const uint32_t *in, int64_t n, uint64_t *out);
// int64 -> int32 (group keys go back to the type of their column)
int narrow_i64_to_i32(flockgpu_ctx *ctx, const int64_t *in, int64_t n, int32_t *out);
// data[i] += delta for i in [0, n): Utf8 offsets of an appended batch rebased onto the column's byte cursor
int add_i32(flockgpu_ctx *ctx, int32_t *data, int64_t n, int32_t delta);

}  // namespace flockgpu
