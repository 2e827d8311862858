/* flockgpu_plan.h -- plan-level C ABI: the drop-in for one `actor::collect` call
 * (flock-function/src/aws/actor.rs:54-79), i.e. for
 *   ExecutionContext::feed_data_sources    flock/src/runtime/context.rs:257-325
 *   ExecutionContext::execute              flock/src/runtime/context.rs:172-191
 *   ExecutionContext::execute_partitioned  flock/src/runtime/context.rs:197-216  (chosen by is_shuffling, :328-337)
 *   ExecutionContext::clean_data_sources   flock/src/runtime/context.rs:227-254
 * over ONE physical plan of `CloudExecutionPlan.execution_plans` (flock/src/runtime/plan.rs:35-43).
 *
 * Hand-off formats are the reference's own:
 *   plan  = the serde_json text of `Arc<dyn ExecutionPlan>` (context.rs:477-480, stage.rs:271; tags
 *           "execution_plan": "projection_exec" | "filter_exec" | "hash_join_exec" | "hash_aggregate_exec" |
 *           "repartition_exec" | "coalesce_batches_exec" | "memory_exec" ..., "physical_expr": "column" |
 *           "binary_expr" | "literal" | "cast_expr" | "try_cast_expr"; fixtures under flock/src/tests/data/plan/)
 *   data  = Arrow RecordBatches through the Arrow C Data Interface (what arrow-rs exports as
 *           FFI_ArrowArray / FFI_ArrowSchema): one struct array per RecordBatch.
 *
 * What runs: the plan is parsed into an operator tree (scan, filter, projection, hash aggregate, inner hash join, hash
 * repartition; Int32 / Int64 / UInt64 / Float64 / Utf8 / Timestamp(ms) columns).  Sub-trees that are one of the NEXMark
 * pipelines -- q1, q2, q3, q5, q7, q8, q13 as whole-query plans, and the Partial COUNT of q5's stage 0 -- run as the fused
 * kernels of flockgpu.h; every other supported node runs on generic device operators, which is how the STAGE plans of
 * the distributed mode execute: the plans either side of a `RepartitionExec Hash` (flock/src/distributed_plan/planner.rs:
 * 152-171, playground/src/distributed_plan/nexmark/q{3,5,8}.dag, split rule flock/src/distributed_plan/stage.rs:269-367),
 * e.g. q3's filter -> Hash([seller]) stage and its join stage, q5's Partial / FinalPartitioned COUNT, MAX and join
 * stages, q8's DISTINCT stages.  A plan whose ROOT is a hash repartition is a shuffling stage (context.rs:328-337):
 * flockgpu_plan_execute_partitioned returns its P hash partitions, partition j going to ring member j
 * (flock-function/src/aws/actor.rs:425-543).  Transparent nodes (RepartitionExec RoundRobinBatch, CoalesceBatchesExec,
 * CoalescePartitions / MergeExec) change neither the row multiset nor the schema and are folded away.
 * SortExec (ORDER BY over Int32 / Int64 / UInt64 / Float64 / Utf8 / Timestamp columns, ASC / DESC, stable) and
 * GlobalLimitExec / LocalLimitExec run on the device too: the reference's own goldens at this boundary end in them
 * (flock/src/runtime/context.rs:471 `ORDER BY c3`, :549-550 `ORDER BY a ASC LIMIT 3`, flock/src/tests/data/plan/join.json) and its
 * stage splitter cuts at sort_exec (flock/src/distributed_plan/stage.rs:337).
 * Anything else -- window functions, other aggregates, other types, outer joins -- returns FLOCKGPU_ERR_UNSUPPORTED so the
 * host keeps its DataFusion path for that plan.  The root projection is honoured: output columns come back in the
 * plan's order under the plan's names.
 *
 * Memory: input batches are BORROWED from flockgpu_plan_feed until the next flockgpu_plan_execute* or flockgpu_plan_reset
 * returns (the reference's MemoryExec owns them for exactly that span), never released, never written.  Buffers in
 * pinned host memory (flockgpu_host_alloc / flockgpu_host_register) go to the DMA engine as they are; pageable buffers are
 * staged through the plan's pinned ring, chunk by chunk, while the previous chunk is in flight.  Feed never waits for
 * the device.  Output batches live in pinned host memory owned by the caller and are freed through their Arrow
 * `release` callback; one execute = one device synchronisation.
 */
#ifndef FLOCKGPU_PLAN_H
#define FLOCKGPU_PLAN_H

#include "flockgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
    const char *format;
    const char *name;
    const char *metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema **children;
    struct ArrowSchema *dictionary;
    void (*release)(struct ArrowSchema *);
    void *private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void **buffers;
    struct ArrowArray **children;
    struct ArrowArray *dictionary;
    void (*release)(struct ArrowArray *);
    void *private_data;
};
#endif

typedef struct flockgpu_plan flockgpu_plan;

/* Parses the plan JSON and builds the operator tree.  FLOCKGPU_ERR_PLAN: not JSON; FLOCKGPU_ERR_UNSUPPORTED: a node,
 * expression or type the engine does not execute (flockgpu_last_error names it).
 * Nodes (`execution_plan` tags): memory_exec, filter_exec, projection_exec, hash_aggregate_exec (Partial / Final / FinalPartitioned: count, max, min,
 * sum, avg), hash_join_exec (Inner), repartition_exec, coalesce_batches_exec / coalesce_partitions_exec / merge_exec (transparent), sort_exec,
 * global_limit_exec / local_limit_exec, window_agg_exec (ROW_NUMBER()).
 * Expressions (`physical_expr` tags): column, literal, cast_expr, try_cast_expr, binary_expr (Eq NotEq Lt LtEq Gt GtEq And Or Plus Minus
 * Multiply Divide Modulo), not_expr, is_null_expr, is_not_null_expr, negative_expr, in_list_expr, case_expr -- over Int32 / Int64 / UInt64 /
 * Float64 / Timestamp(Millisecond) values (Utf8: =, <>, IN, IS NULL against literals).  Both operands of a binary operator have one type,
 * as the reference's planner leaves them.  At execute, integer division / modulo by zero in a row whose operands are not NULL and a
 * CAST whose value does not fit its target are FLOCKGPU_ERR_INVALID for the call (the reference's execute fails with an ArrowError);
 * TRY_CAST yields NULL instead. */
int flockgpu_plan_create(flockgpu_ctx *ctx, const char *plan_json, size_t len, flockgpu_plan **out);
/* The same with options.  FLOCKGPU_PLAN_GENERIC_ONLY: no sub-tree is handed to a fused NEXMark pipeline, every node runs on the
 * generic device operators -- how the tests run every reference plan both ways and require identical rows. */
#define FLOCKGPU_PLAN_GENERIC_ONLY 1u
int flockgpu_plan_create_ex(flockgpu_ctx *ctx, const char *plan_json, size_t len, uint32_t flags, flockgpu_plan **out);
void flockgpu_plan_destroy(flockgpu_plan *plan);
/* Host-only: parses a plan without a device context.  *query receives the NEXMark query number (1, 2, 3, 4, 5, 7, 8, 9, 13;
 * 100 for the Yahoo Streaming Benchmark's query) when the whole plan is one fused pipeline, 0 for any other executable plan
 * (stage plans, generic operator trees).  q4 / q9 run fused on batches whose auction ids are dense and increasing and on the
 * generic operators otherwise (decided per execute). */
int flockgpu_plan_recognise(const char *plan_json, size_t len, int *query);
/* Host-only: the operator tree with derived schemas and, per node, what executes it (a fused pipeline or the generic
 * operators), as text.  Returns the status flockgpu_plan_create would; on UNSUPPORTED the text says why. */
int flockgpu_plan_explain(const char *plan_json, size_t len, char *out, size_t capacity);

int flockgpu_plan_query(const flockgpu_plan *plan);
const char *flockgpu_plan_description(const flockgpu_plan *plan); /* the text of flockgpu_plan_explain */
/* Leaves of the plan (MemoryExec), in the order feed expects them; the name is the relation whose columns the
 * leaf scans ("bid", "auction", "person", "side_input", or "" when the columns name none of them), found the way
 * feed_data_sources does: by column-name set (compare_schema, context.rs:402-416). */
int flockgpu_plan_num_inputs(const flockgpu_plan *plan);
const char *flockgpu_plan_input_name(const flockgpu_plan *plan, int input);
/* 1 when every column the plan READS from leaf `input` is present in `schema` (a struct schema). */
int flockgpu_plan_input_matches(const flockgpu_plan *plan, int input, const struct ArrowSchema *schema);
/* is_shuffling (context.rs:328-337): 1 when the plan's root is a hash repartition; its partition count (else 1). */
int flockgpu_plan_is_shuffling(const flockgpu_plan *plan);
int flockgpu_plan_output_partitions(const flockgpu_plan *plan);

/* feed_data_sources for one leaf: all batches of all partitions of the relation, flattened.  May be called more than
 * once per leaf (rows append); an unfed leaf is an empty relation (context.rs:305-314).  Only the columns the plan reads
 * are copied.  A rejected feed (missing column, wrong type) leaves the leaf unchanged.
 * NULLs (validity bitmaps, any `null_count`): a row whose NULL cannot change the result -- an inner-join key, a column that is
 * only compared under AND -- is left out at the feed; every other column that holds NULLs travels with one validity byte per
 * row and the device operators honour it the way DataFusion's do: a comparison with NULL
 * keeps no row (through OR, too), COUNT(col) counts the non-NULL values, MIN / MAX / SUM / AVG skip NULLs and are NULL over
 * nothing but NULLs, NULL values of a GROUP BY key column form one group, ORDER BY places NULLs by the plan's `nulls_first`,
 * NULLs in projected / joined-along columns come back as NULLs (validity bitmap + null_count on the exported arrays).  The fused
 * NEXMark pipelines read plain columns: an invocation whose leaf holds such NULLs runs on the generic operators.  Handed back
 * as FLOCKGPU_ERR_UNSUPPORTED at execute: NULLs in a two-column GROUP BY key, in DISTINCT
 * columns, in a computed join key; FLOCKGPU_ERR_UNSUPPORTED at feed: such NULLs on the q5 plan with an open pane ring (its ring keeps
 * Partial COUNT groups of plain columns; every other plan's ring keeps rows, validity included). */
int flockgpu_plan_feed(flockgpu_plan *plan, int input, const struct ArrowSchema *schema,
                       const struct ArrowArray *const *batches, int n_batches);

/* One upload for several plans hosted by one process: leaf `input` of `plan` reads the relation that `donor`'s leaf
 * `donor_input` was fed, in place (no copy; both plans must have been created on the same flockgpu_ctx, whose stream orders
 * the donor's transfers before this plan's kernels).  The reference feeds every function its own copy of the source
 * payload (flock-function/src/aws/actor.rs:54-79); two stage plans that scan the same relation -- q5's two subplans both
 * start at `bid` (playground/src/distributed_plan/nexmark/q5.dag) -- hosted by one GPU need the 4 B x bids on the device
 * once.  Valid until `donor` is fed again or reset: execute this plan first.  FLOCKGPU_ERR_UNSUPPORTED (nothing changed,
 * feed the plan its own copy) when the donor does not hold a column this plan reads, or dropped rows with NULLs. */
int flockgpu_plan_feed_shared(flockgpu_plan *plan, int input, const flockgpu_plan *donor, int donor_input);

/* execute(): runs the plan on everything fed so far as ONE window and exports one RecordBatch. */
int flockgpu_plan_execute(flockgpu_plan *plan, struct ArrowSchema *out_schema, struct ArrowArray *out_batch);
/* execute_partitioned(): `out_batches` has room for `capacity` record batches; *n_partitions of them are filled: the
 * plan's hash partitions when it is a shuffling stage (rows of equal key in one partition), else one batch.  The
 * batches share one pinned block; each is released on its own. */
int flockgpu_plan_execute_partitioned(flockgpu_plan *plan, struct ArrowSchema *out_schema, struct ArrowArray *out_batches,
                                      int capacity, int *n_partitions);

/* Stage-to-stage hand-over in HBM, for stage plans hosted by ONE process on one flockgpu_ctx (the reference's boundary between two
 * stages is a network hop carrying Arrow Flight data, flock-function/src/aws/actor.rs:425-543; between two stages on one GPU it is a
 * pointer).  execute_retain runs the plan like execute but leaves the result on the device instead of exporting it (*rows = its row
 * count; a root hash repartition is skipped: which partition a key lands in is unobservable when every partition goes to the same
 * consumer).  feed_from makes leaf `input` of `plan` read that result in place -- columns matched by name and type like
 * flockgpu_plan_feed does; FLOCKGPU_ERR_UNSUPPORTED, nothing changed, when the result lacks a column the plan reads, so a host can
 * try its producers in turn.  The result stays valid until the producer is fed, executed or reset again: execute the consumer first.
 * No host wait in either call. */
int flockgpu_plan_execute_retain(flockgpu_plan *plan, int64_t *rows);
int flockgpu_plan_feed_from(flockgpu_plan *plan, int input, const flockgpu_plan *producer);

/* clean_data_sources(): drops the inputs, keeps device arenas and hash-table sizing for the next invocation.  On a plan with
 * an open pane ring (below) it ends the window instead: the oldest pane of a full ring is retired, the others stay. */
int flockgpu_plan_reset(flockgpu_plan *plan);

/* ---- asynchronous execute: the reference runs every plan of a function on its own tokio task and joins them
 * (`tokio::spawn(collect(plan))`, flock/src/runtime/context.rs:172-191).  execute_async hands the whole execute (kernels, its host
 * waits, the export into pinned memory) to the worker thread of the plan's flockgpu_ctx and returns at once; flockgpu_plan_wait blocks
 * until it has finished, returns ITS status and fills the outputs exactly as flockgpu_plan_execute[_partitioned] would.  One call in
 * flight per ctx; plans on different ctxs (different streams) overlap -- the host gap of one call is covered by the other's kernels.
 * Between the two calls the plan and its ctx must not be touched. */
int flockgpu_plan_execute_async(flockgpu_plan *plan, int partitioned);
int flockgpu_plan_wait(flockgpu_plan *plan, struct ArrowSchema *out_schema, struct ArrowArray *out_batches, int capacity, int *n_partitions);

/* ---- hopping windows on the streaming path: a device-side pane ring.
 * The reference's hopping launcher re-sends every window whole (flock-function/src/aws/window/hopping.rs:52-74; the local twin
 * flock/src/datasource/nexmark/queries/q5.rs:78-132): with Hopping(size, hop) every event crosses the wire -- here: PCIe -- size / hop
 * times and is aggregated as often.  A plan with an open ring keeps the last `panes_per_window` panes (pane = hop seconds of events)
 * ON THE DEVICE across executes; the host feeds only the NEW pane:
 *     flockgpu_plan_ring_open(plan, size / hop);
 *     per pane p:  flockgpu_plan_feed_pane(plan, input, p, schema, batches, n)   (once per input and as often as there are batches;
 *                                                                                 an empty pane is fed with n_batches = 0)
 *                  flockgpu_plan_execute*(...)     -> the window of the panes the ring holds, [p - panes_per_window + 1, p]
 *                  flockgpu_plan_reset(plan)       -> retires pane p - panes_per_window + 1
 * Every event is uploaded once.  What is retained per pane depends on the plan: for q5 (COUNT(*) GROUP BY auction -> MAX -> join)
 * it is the pane's Partial aggregate state -- its (auction, count) groups, the state HashAggregateExec(Partial) emits -- so every bid
 * is also COUNTED once and a window is the FinalPartitioned merge of its panes' groups; for every other plan it is the pane's rows
 * (the columns the plan reads), and the window is executed over the retained rows.
 * Pane ids are consecutive: the first feed_pane names the first pane, after that only the newest pane (more batches) or its
 * successor (after a reset if the ring is full) may be fed; anything else -- an older pane, a skipped pane -- is
 * FLOCKGPU_ERR_INVALID and leaves the ring as it was.  flockgpu_plan_feed on a plan with an open ring is FLOCKGPU_ERR_INVALID.
 * ring_close drops every pane and returns the plan to whole-window feeding. */
int flockgpu_plan_ring_open(flockgpu_plan *plan, int panes_per_window);
int flockgpu_plan_ring_close(flockgpu_plan *plan);
/* panes currently held: ids [*first_pane, *first_pane + *n_panes); n_panes = 0 before the first feed_pane */
int flockgpu_plan_ring_state(const flockgpu_plan *plan, int64_t *first_pane, int *n_panes, int *panes_per_window);
int flockgpu_plan_feed_pane(flockgpu_plan *plan, int input, int64_t pane_id, const struct ArrowSchema *schema,
                            const struct ArrowArray *const *batches, int n_batches);
/* Overlap of the NEXT pane's upload with the current window's execute: brings the batches of pane `pane_id` (the ring's next pane) of
 * `input` into side buffers -- on a stream of the plan's own; pageable memory is staged by threads of the plan's own -- and returns at
 * once; the ring does not change.  The pane is then fed with flockgpu_plan_feed_pane(plan, input, pane_id, NULL, NULL, 0), which
 * appends the prefetched rows device to device.  One pane of one input at a time; fixed-width and (since round 5) Utf8 columns -- offsets
 * and bytes travel raw, the offsets are rebased onto the column's byte cursor when the pane is appended -- without NULLs
 * (FLOCKGPU_ERR_UNSUPPORTED otherwise: feed that pane the ordinary way).  The batches stay borrowed until the flockgpu_plan_reset that
 * follows the pane's feed; flockgpu_plan_ring_close / flockgpu_plan_destroy drop a prefetch that was never fed. */
int flockgpu_plan_prefetch_pane(flockgpu_plan *plan, int input, int64_t pane_id, const struct ArrowSchema *schema,
                                const struct ArrowArray *const *batches, int n_batches);

/* ---- hash placement of a shuffling stage.  flockgpu_plan_execute_partitioned places a row in partition
 *     (murmur3_fmix32(fold32(key)) * n) >> 32
 * (flockgpu.h: flockgpu_partition_by_key), NOT where the reference's DataFusion fork would put it
 * (`create_hashes(.., ahash::RandomState::with_seeds(0, 0, 0, 0)) % n`, restated at playground/src/distributed_plan/shuffle_writer.rs:106-128;
 * the ahash version is unpinned).  Which partition a key lands in is unobservable as long as EVERY producer
 * of a stage places rows the same way; a stage with mixed producers -- some function instances on libflockgpu, some on DataFusion
 * during a roll-out, a GPU-less fallback instance -- would split one key over two consumers (partition j -> ring member j,
 * flock-function/src/aws/actor.rs:425-543) and silently lose join / aggregate rows.  Two guards:
 *   * every batch of a shuffling stage's output carries the Arrow schema metadata  "flockgpu.partition_scheme" = this string
 *     (Arrow Flight keeps schema metadata, so it travels with the payload);
 *   * flockgpu_plan_feed refuses (FLOCKGPU_ERR_INVALID, nothing changed) a batch whose tag differs from what the plan's other
 *     co-partitioned inputs -- the leaves below a HashJoinExec mode=Partitioned or a FinalPartitioned aggregate -- were fed since
 *     the last reset: tagged with another scheme, or untagged next to tagged ones.
 * A host that schedules stages should also compare flockgpu_plan_partition_scheme() across the producers it is about to start. */
const char *flockgpu_plan_partition_scheme(void);
int flockgpu_plan_check_partition_scheme(const char *scheme);   /* FLOCKGPU_OK | FLOCKGPU_ERR_UNSUPPORTED */

/* Pinned host memory for the host's Arrow buffers ("Arrow buffers pinned and hipMemcpyAsync'd"): allocate buffers
 * with flockgpu_host_alloc, or register existing ones for as long as they are fed. */
int flockgpu_host_alloc(size_t bytes, void **out);
int flockgpu_host_free(void *ptr);
int flockgpu_host_register(void *ptr, size_t bytes);
int flockgpu_host_unregister(void *ptr);

#ifdef __cplusplus
}
#endif
#endif /* FLOCKGPU_PLAN_H */
