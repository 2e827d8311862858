/* flockgpu_plan.h -- plan-level C ABI: the drop-in for one `actor::collect` call
 * (flock-function/src/aws/actor.rs:54-79), i.e. for
 *   ExecutionContext::feed_data_sources   flock/src/runtime/context.rs:257-325
 *   ExecutionContext::execute             flock/src/runtime/context.rs:172-191
 *   ExecutionContext::clean_data_sources  flock/src/runtime/context.rs:227-254
 * over ONE physical plan of `CloudExecutionPlan.execution_plans` (flock/src/runtime/plan.rs:35-43).
 *
 * Hand-off formats are the reference's own:
 *   plan  = the serde_json text of `Arc<dyn ExecutionPlan>` (context.rs:477-480, stage.rs:271; tags
 *           "execution_plan": "projection_exec" | "filter_exec" | "hash_join_exec" | "hash_aggregate_exec" |
 *           "repartition_exec" | "coalesce_batches_exec" | "memory_exec" ..., "physical_expr": "column" |
 *           "binary_expr" | "literal" | "cast_expr" | "try_cast_expr"; fixtures under flock/src/tests/data/plan/)
 *   data  = Arrow RecordBatches through the Arrow C Data Interface (what arrow-rs exports as
 *           FFI_ArrowArray / FFI_ArrowSchema): one struct array per RecordBatch.
 * The engine recognises the plan shapes of NEXMark q1, q2, q3, q5 and q8, of the "next"
 * queries q7 and q13 (section 8(f): q13's side input is fed as the plan's second relation), and
 * returns FLOCKGPU_ERR_UNSUPPORTED for anything else, so the host can keep its DataFusion path for those.
 * Transparent nodes (RepartitionExec, CoalesceBatchesExec, CoalescePartitions/MergeExec, renaming
 * ProjectionExec) and the Partial/Final split of HashAggregateExec have no effect on the row multiset and are
 * folded away.
 *
 * Input batches are BORROWED for the duration of flockgpu_plan_feed (never released, never written); the
 * output batch is owned by the caller and freed through its Arrow `release` callback.  Host buffers cross
 * PCIe here; device-resident callers use the flockgpu_q*_ entry points of flockgpu.h instead.
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

/* Parses the plan JSON and matches it against the supported shapes. */
int flockgpu_plan_create(flockgpu_ctx *ctx, const char *plan_json, size_t len, flockgpu_plan **out);
void flockgpu_plan_destroy(flockgpu_plan *plan);
/* Host-only: parses + matches a plan without a device context.  *query receives 1/2/3/5/8; returns
 * FLOCKGPU_OK, FLOCKGPU_ERR_PLAN (bad JSON) or FLOCKGPU_ERR_UNSUPPORTED (not one of the recognised shapes). */
int flockgpu_plan_recognise(const char *plan_json, size_t len, int *query);

/* NEXMark query number the plan was recognised as (1, 2, 3, 5, 7, 8, 13). */
int flockgpu_plan_query(const flockgpu_plan *plan);
/* Leaves of the plan (MemoryExec), in the order feed expects them; the name is the relation whose columns the
 * leaf scans ("bid", "auction", "person"), found the way feed_data_sources does: by column-name set
 * (compare_schema, context.rs:402-416). */
int flockgpu_plan_num_inputs(const flockgpu_plan *plan);
const char *flockgpu_plan_input_name(const flockgpu_plan *plan, int input);
/* 1 when every column name the leaf `input` needs is present in `schema` (a struct schema). */
int flockgpu_plan_input_matches(const flockgpu_plan *plan, int input, const struct ArrowSchema *schema);

/* feed_data_sources for one leaf: all batches of all partitions of the relation, flattened.  May be called once
 * per leaf; an unfed leaf is an empty relation (context.rs:305-314). */
int flockgpu_plan_feed(flockgpu_plan *plan, int input, const struct ArrowSchema *schema,
                       const struct ArrowArray *const *batches, int n_batches);

/* execute(): runs the fused pipeline on everything fed so far as ONE window and exports one RecordBatch. */
int flockgpu_plan_execute(flockgpu_plan *plan, struct ArrowSchema *out_schema, struct ArrowArray *out_batch);

/* clean_data_sources(): drops the inputs, keeps device arenas and hash-table sizing for the next invocation. */
int flockgpu_plan_reset(flockgpu_plan *plan);

#ifdef __cplusplus
}
#endif
#endif /* FLOCKGPU_PLAN_H */
