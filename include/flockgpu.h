/* flockgpu.h -- C ABI of libflockgpu, the MI355X (gfx950) execution kernel for Flock's
 * per-batch DataFusion hot path (filter, projection, hash-join, hash-aggregate over NEXMark
 * RecordBatches).  Plain C, plain pointers and sizes: this is exactly what a Rust
 * `extern "C"` block in flock/src/runtime would bind (INTEGRATION.md shows the stub).
 *
 * Every entry point returns an int status (FLOCKGPU_OK = 0); nothing throws or aborts across
 * the ABI; `flockgpu_last_error(ctx)` returns the message of the last failure on that ctx.
 * (The reference stringifies DataFusion errors into FlockError::Execution and then
 * unwraps them, flock/src/runtime/context.rs:181,189 -- i.e. an operator error kills the
 * Lambda; here it is a status code the host can turn into FlockError::Execution.)
 *
 * Threading: a ctx (stream + device arena) is used by one thread at a time; different ctxs
 * may be used concurrently from arbitrary threads (the reference runs each plan on its own
 * tokio task, context.rs:178).  No global mutable state.
 *
 * Memory: all column pointers in the `*_cols` views are DEVICE pointers (HBM) laid out
 * exactly like the corresponding Arrow buffers (values; int32 offsets + bytes for Utf8;
 * all NEXMark fields are non-nullable, event.rs:130-149,220-245,336-352).  They are
 * BORROWED for the duration of the call and never written (the reference asserts inputs
 * survive execution, datasource/nexmark/queries/q5.rs:127-131).  Result buffers are owned
 * by the ctx arena and stay valid until the next call of the same query on that ctx or
 * flockgpu_ctx_destroy.
 */
#ifndef FLOCKGPU_H
#define FLOCKGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLOCKGPU_ABI_VERSION 1

enum {
    FLOCKGPU_OK = 0,
    FLOCKGPU_ERR_INVALID = 1,     /* bad argument                                            */
    FLOCKGPU_ERR_HIP = 2,         /* a HIP runtime call failed (message has the hipError)     */
    FLOCKGPU_ERR_OOM = 3,         /* device arena allocation failed                           */
    FLOCKGPU_ERR_UNSUPPORTED = 4, /* plan shape / size outside what the kernels implement     */
    FLOCKGPU_ERR_CAPACITY = 5,    /* hash table overflow after the retry budget               */
    FLOCKGPU_ERR_PLAN = 6,        /* plan JSON could not be parsed / matched                  */
    FLOCKGPU_ERR_PEER = 7         /* exchange: another rank of the communicator failed (flockgpu_comm.h); this rank's inputs are fine */
};

typedef struct flockgpu_ctx flockgpu_ctx;

/* ---- context ------------------------------------------------------------------------------- */
/* `hip_stream` may be NULL (the ctx creates its own stream, ordered against the legacy default
 * stream like any blocking HIP stream) or an existing hipStream_t the caller launches on. */
int flockgpu_ctx_create(int device, void *hip_stream, flockgpu_ctx **out);
void flockgpu_ctx_destroy(flockgpu_ctx *ctx);
const char *flockgpu_last_error(const flockgpu_ctx *ctx);
int flockgpu_ctx_synchronize(flockgpu_ctx *ctx);
int flockgpu_abi_version(void);

/* ---- plain memory helpers (hosts without their own HIP binding) --------------------------------- */
enum { FLOCKGPU_H2D = 1, FLOCKGPU_D2H = 2, FLOCKGPU_D2D = 3 };
int flockgpu_malloc(flockgpu_ctx *ctx, size_t bytes, void **out_device_ptr);
int flockgpu_free(flockgpu_ctx *ctx, void *device_ptr);
/* Copies on the ctx stream and waits for completion. */
int flockgpu_memcpy(flockgpu_ctx *ctx, void *dst, const void *src, size_t bytes, int kind);
/* Testing aid: device memory that ENDS where mapped address space ends (`bytes` rounded up to 16, placed at the tail of its own
 * physical allocation; the address granule behind it is reserved and left unmapped).  A kernel that touches memory past the end of a
 * column placed there faults -- with ordinary allocations it reads whatever lies next to it and nobody notices.  Freed with
 * flockgpu_free_guarded (or with the ctx): the physical memory goes back, the address range stays reserved for the life of the
 * process, so that no guarded address is ever handed out twice. */
int flockgpu_malloc_guarded(flockgpu_ctx *ctx, size_t bytes, void **out_device_ptr);
int flockgpu_free_guarded(flockgpu_ctx *ctx, void *device_ptr);

/* Per-kernel HIP-event timing (bench.py's roofline leg).  When enabled, every kernel launch of
 * the ctx is bracketed by hipEvents on the ctx stream; totals are read back per kernel name. */
int flockgpu_profile_enable(flockgpu_ctx *ctx, int on);
/* Restricts the bracketing to launches of one kernel (NULL: every kernel again), or of any of several: "a|b" -- the kernels a call
 * chooses between for one step (q3's probe runs `q3_probe_flag_small_kernel` while all its tiles are resident at once, else
 * `q3_probe_flag_kernel`).  Two event records per launch are markers on the stream: bracketing all ~20 launches of a small-batch
 * query costs as much as its kernels. */
int flockgpu_profile_only(flockgpu_ctx *ctx, const char *kernel_name);
int flockgpu_profile_reset(flockgpu_ctx *ctx);
/* Fills up to `cap` entries; returns the number of distinct kernels seen in *n. */
typedef struct {
    char name[48];
    uint64_t launches;
    double total_ms;
} flockgpu_kernel_stat;
int flockgpu_profile_read(flockgpu_ctx *ctx, flockgpu_kernel_stat *out, int cap, int *n);
/* The per-launch durations behind one kernel's total, in launch order (the first 4096 since the last reset): *n = how many
   exist, out_ms receives min(*n, cap) of them.  A HIP-event pair also spans the host's gap between recording the start event and
   enqueueing the kernel when the stream is idle, so for 10-microsecond kernels one preempted launch can double a 20-launch mean;
   the spread (bench.py reports min / median / max beside the mean) shows it. */
int flockgpu_profile_samples(flockgpu_ctx *ctx, const char *kernel_name, float *out_ms, int cap, int *n);

/* ---- column views (device pointers, Arrow buffer layout) -------------------------------------- */
typedef struct { /* Bid::schema, event.rs:336-352 */
    const int32_t *auction, *bidder, *price;
    const int64_t *b_date_time;
    int64_t rows;
} flockgpu_bid_cols;

typedef struct { /* Auction::schema projection used by q3/q8 (q3_plan.fmt:4, q8_plan.fmt:10) */
    const int32_t *a_id, *seller, *category;
    int64_t rows;
} flockgpu_auction_cols;

typedef struct { /* Arrow Utf8: offsets has rows+1 entries */
    const int32_t *offsets;
    const uint8_t *data;
} flockgpu_utf8;

typedef struct { /* Person::schema projection used by q3/q8 (q3_plan.fmt:6, q8_plan.fmt:6) */
    const int32_t *p_id;
    flockgpu_utf8 name, city, state;
    int64_t rows;
} flockgpu_person_cols;

/* A window schedule over one relation: `n_panes + 1` row offsets (HOST pointer) delimit
 * consecutive, disjoint panes of rows; window w covers panes [win_pane_lo[w], win_pane_hi[w]).
 *   ElementWise  (window/elementwise.rs:46)   : pane = 1-s epoch, window = 1 pane
 *   Tumbling(s)  (window/tumbling.rs:55-57)   : pane = s epochs,  window = 1 pane
 *   Hopping(w,h) (window/hopping.rs:54-57)    : pane = gcd(w,h) epochs, window = w/gcd panes, stride h/gcd
 * One call executes every window of the schedule = the reference's per-window loop of
 * `actor::collect` (flock-function/src/aws/actor.rs:54-79) batched into a few launches. */
typedef struct {
    const int64_t *pane_row_offsets; /* host, n_panes + 1, non-decreasing */
    int32_t n_panes;
    const int32_t *win_pane_lo; /* host, n_windows */
    const int32_t *win_pane_hi; /* host, n_windows */
    int32_t n_windows;
} flockgpu_windows;

/* ---- q1: ProjectionExec [auction, bidder, 0.908 * CAST(price AS Float64) AS price, b_date_time]
 * (planner.rs:90, q1_plan.fmt:1).  Pass-through columns are zero-copy in the reference (Arc
 * clones); the kernel materialises only the computed Float64 column.  `factor` = the literal. */
int flockgpu_q1_project(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, double factor,
                        double *out_price /* device, bid->rows */);

/* ---- q2: FilterExec CAST(auction AS Int64) % modulus = 0 -> CoalesceBatches -> Projection
 * [auction, price] (planner.rs:120-124, q2_plan.fmt:1-3).  Stable: input row order is kept.
 * out_*: device buffers in the ctx arena; win_out_offsets: HOST array of n_windows + 1 row
 * offsets into out_* (arena-owned).  Requires single-pane windows (ElementWise). */
typedef struct {
    const int32_t *auction, *price;   /* device */
    const int64_t *win_out_offsets;   /* host, n_windows + 1 */
    int64_t rows;
} flockgpu_q2_result;
int flockgpu_q2_filter(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                       int64_t modulus, flockgpu_q2_result *out);

/* ---- q3: Filter(category = lit) on auction, Filter(state = l0 OR state = l1 OR ...) on person,
 * HashJoinExec(Inner, seller = p_id), Projection [name, city, state, a_id]
 * (planner.rs:152-171, q3_plan.fmt:1-6).  `auction_win` / `person_win` describe the same
 * n_windows windows over the two relations.  Output rows are grouped by window, ordered by
 * auction row inside a window (row order across partitions is not a contract in the reference:
 * results compare as sorted multisets, test_util.rs:61-90). */
typedef struct {
    flockgpu_utf8 name, city, state;  /* device */
    const int32_t *a_id;              /* device */
    const int32_t *auction_row, *person_row; /* device: matching input row pairs */
    const int64_t *win_out_offsets;   /* host, n_windows + 1 */
    int64_t rows;
    int64_t name_bytes, city_bytes, state_bytes;
} flockgpu_q3_result;
int flockgpu_q3_join(flockgpu_ctx *ctx, const flockgpu_auction_cols *auction, const flockgpu_windows *auction_win,
                     const flockgpu_person_cols *person, const flockgpu_windows *person_win,
                     int64_t category_lit, const char *const *state_lits, int n_state_lits,
                     flockgpu_q3_result *out);

/* ---- q5: COUNT(*) GROUP BY auction; MAX(num); rows with num = maxn, ties kept
 * (q5.sql, q5_plan.fmt:1-13, q5.dag).  Hopping windows share panes: every bid is read once.
 * Output: per window, (auction Int32, num UInt64) rows sorted by auction.
 * The call returns with its results complete and may leave one clean-up kernel queued on the ctx stream (it zeroes the counters the
 * call used, so that the next call does not have to): work the caller queues on the same stream simply runs behind it. */
typedef struct {
    const int32_t *auction;           /* device */
    const uint64_t *num;              /* device */
    const int64_t *win_out_offsets;   /* host, n_windows + 1 */
    const uint64_t *win_max;          /* host, n_windows: MAX(num) per window (0 = empty window) */
    const uint64_t *win_groups;       /* host, n_windows: number of distinct auctions per window */
    int64_t rows;
} flockgpu_q5_result;
int flockgpu_q5_hot_items(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                          flockgpu_q5_result *out);

/* q5 as the two stages of q5.dag around the hash repartition (the key-partitioned exchange of a multi-GPU run):
 *   partial  : HashAggregateExec mode=Partial gby=[auction] COUNT(1), PER PANE -- the groups of pane p are rows
 *              [pane_out_offsets[p], pane_out_offsets[p+1]) of (auction, count); windows are not used, only the panes
 *   weighted : the FinalPartitioned side + MAX + join of flockgpu_q5_hot_items over rows that each carry a count (the
 *              partial groups this partition received); the sum of `count` over a window must stay below 2^32.
 * partial on N row-stripes, repartition of the groups by auction, weighted on what arrives == hot_items on all rows. */
typedef struct {
    const int32_t *auction;            /* device, ctx-owned */
    const uint32_t *count;             /* device, ctx-owned */
    const int64_t *pane_out_offsets;   /* host, n_panes + 1 */
    int64_t rows;
} flockgpu_q5_partial_result;
int flockgpu_q5_partial_counts(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                               flockgpu_q5_partial_result *out);
int flockgpu_q5_hot_items_weighted(flockgpu_ctx *ctx, const int32_t *auction, const uint32_t *count, int64_t rows,
                                   const flockgpu_windows *win, flockgpu_q5_result *out);

/* ---- q7 (SURVEY.md section 8(f) "next" query): bid JOIN (SELECT MAX(price) AS maxprice FROM bid) ON price = maxprice,
 * Projection [auction, price, bidder, b_date_time] (benchmarks/src/nexmark/query/q7.sql, q7_plan.fmt), per
 * Tumbling(10 s) window (benchmarks/src/nexmark/main.rs:119).  Every row that reaches the window's maximum is returned
 * (ties kept), in input order; an empty window (MAX = NULL) returns nothing.  win_max: HOST array, the maximum per
 * window (INT32_MIN for an empty window). */
typedef struct {
    const int32_t *auction, *price, *bidder; /* device */
    const int64_t *b_date_time;              /* device */
    const int64_t *win_out_offsets;          /* host, n_windows + 1 */
    const int64_t *win_max;                  /* host, n_windows */
    int64_t rows;
} flockgpu_q7_result;
int flockgpu_q7_highest_bid(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                            flockgpu_q7_result *out);

/* ---- q4 / q9 (SURVEY.md section 8(f) "next" queries), per ElementWise window (benchmarks/src/nexmark/main.rs:117):
 *   Q  = SELECT a_id [, category], MAX(price) AS final FROM auction INNER JOIN bid ON a_id = auction
 *        WHERE b_date_time BETWEEN a_date_time AND expires GROUP BY a_id [, category]
 *   q9 = bid JOIN Q ON auction = id AND price = final -> [auction, bidder, price, b_date_time]  (q9.sql, q9_plan.fmt)
 *   q4 = SELECT category, AVG(final) FROM Q GROUP BY category -> [category Int32, AVG Float64]   (q4.sql)
 * Auction ids must be strictly increasing over a dense range inside every window (the generator's order);
 * otherwise FLOCKGPU_ERR_UNSUPPORTED and the host keeps its own engine for that input.  q9 rows keep input order;
 * q4 rows are ordered by category inside a window. */
typedef struct { /* Auction::schema projection of q4 / q9 (q9_plan.fmt: [0, 5, 6]; q4 adds category) */
    const int32_t *a_id, *category; /* category may be NULL for q9 */
    const int64_t *a_date_time, *expires;
    int64_t rows;
} flockgpu_auction_time_cols;
typedef struct {
    const int32_t *auction, *price, *bidder; /* device */
    const int64_t *b_date_time;              /* device */
    const int64_t *win_out_offsets;          /* host, n_windows + 1 */
    int64_t rows;
} flockgpu_q9_result;
typedef struct {
    const int32_t *category;        /* device */
    const double *avg_final;        /* device */
    const int64_t *win_out_offsets; /* host, n_windows + 1 */
    int64_t rows;
} flockgpu_q4_result;
int flockgpu_q9_winning_bids(flockgpu_ctx *ctx, const flockgpu_auction_time_cols *auction, const flockgpu_windows *auction_win,
                             const flockgpu_bid_cols *bid, const flockgpu_windows *bid_win, flockgpu_q9_result *out);
int flockgpu_q4_avg_final_by_category(flockgpu_ctx *ctx, const flockgpu_auction_time_cols *auction,
                                      const flockgpu_windows *auction_win, const flockgpu_bid_cols *bid,
                                      const flockgpu_windows *bid_win, flockgpu_q4_result *out);

/* ---- q13 (SURVEY.md section 8(f) "next" query): bid JOIN side_input ON auction = key ->
 * [auction, bidder, price, b_date_time, value] (benchmarks/src/nexmark/query/q13.sql, q13_plan.fmt; side_input schema
 * flock/src/datasource/nexmark/event.rs:375-388), per ElementWise window.  The side input (device pointers, any keys,
 * duplicates allowed) is the same for every window.  Rows are ordered by bid row; bid_row / side_row name the joined
 * input rows. */
typedef struct {
    const int32_t *auction, *bidder, *price; /* device */
    const int64_t *b_date_time;              /* device */
    const int32_t *value;                    /* device */
    const int32_t *bid_row, *side_row;       /* device */
    const int64_t *win_out_offsets;          /* host, n_windows + 1 */
    int64_t rows;
} flockgpu_q13_result;
int flockgpu_q13_side_join(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                           const int32_t *side_key, const int32_t *side_value, int64_t side_rows, flockgpu_q13_result *out);

/* ---- q8: DISTINCT (p_id, name) JOIN DISTINCT seller ON p_id = seller -> [p_id, name]
 * (q8.sql, q8_plan.fmt:1-10, q8.dag).  Output grouped by window, ordered by person row. */
typedef struct {
    const int32_t *p_id;              /* device */
    flockgpu_utf8 name;               /* device */
    const int32_t *person_row;        /* device */
    const int64_t *win_out_offsets;   /* host, n_windows + 1 */
    int64_t rows;
    int64_t name_bytes;
} flockgpu_q8_result;
int flockgpu_q8_join(flockgpu_ctx *ctx, const flockgpu_person_cols *person, const flockgpu_windows *person_win,
                     const flockgpu_auction_cols *auction, const flockgpu_windows *auction_win,
                     flockgpu_q8_result *out);

/* ---- asynchronous calls.  The reference runs every plan of a function on its own tokio task and joins them
 * (flock/src/runtime/context.rs:172-191); a synchronous call here leaves the GPU idle while the host prepares the next one and
 * wakes up from its wait (~0.09 ms of q5's 0.94 ms step, a third of q3's 0.10 ms at 1e8 events).  Every ctx owns one worker thread
 * (started on first use).  A `*_async` entry point copies its argument structs, hands the call to the worker and returns;
 * flockgpu_ctx_wait blocks until the call has finished and returns ITS status, with `out` filled exactly as by the synchronous
 * call.  One call in flight per ctx -- a second submit before the wait is FLOCKGPU_ERR_INVALID; the arrays the arguments point to
 * (device columns, window schedules, literals, `out`) are borrowed until the wait returns, and the ctx must not be used in
 * between.  Calls on DIFFERENT ctxs (different streams) are in flight together. */
int flockgpu_q3_join_async(flockgpu_ctx *ctx, const flockgpu_auction_cols *auction, const flockgpu_windows *auction_win,
                           const flockgpu_person_cols *person, const flockgpu_windows *person_win,
                           int64_t category_lit, const char *const *state_lits, int n_state_lits,
                           flockgpu_q3_result *out);
int flockgpu_q5_hot_items_async(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win,
                                flockgpu_q5_result *out);
int flockgpu_q8_join_async(flockgpu_ctx *ctx, const flockgpu_person_cols *person, const flockgpu_windows *person_win,
                           const flockgpu_auction_cols *auction, const flockgpu_windows *auction_win,
                           flockgpu_q8_result *out);
int flockgpu_ctx_wait(flockgpu_ctx *ctx);

/* ---- key-partitioned exchange (the RepartitionExec Hash([key], n) step of the distributed plans,
 * flock/src/distributed_plan/planner.rs:152-171; row routing restated from
 * playground/src/distributed_plan/shuffle_writer.rs:106-148).  The reference hashes with ahash seeds (0,0,0,0);
 * which partition a key lands on is unobservable in query results, so a fixed integer mix is used:
 *   part(key) = (murmur3_fmix32((uint32_t)key) * n_parts) >> 32.
 * Result: the row numbers of every row that lies inside a window, grouped by (partition, window) -- partition
 * major -- with input order kept inside a group, so that `take` over them yields send buffers that are contiguous
 * per destination (one all-to-all chunk) and per window inside a destination.  part_win_offsets is a HOST array of
 * n_parts * n_windows + 1 offsets into `row` (arena-owned, valid until the next call on this ctx). */
typedef struct {
    const int32_t *row;              /* device */
    const int64_t *part_win_offsets; /* host */
    int64_t rows;
} flockgpu_partition_result;
int flockgpu_partition_by_key(flockgpu_ctx *ctx, const int32_t *keys /* device, 16-byte aligned */, int64_t rows,
                              const flockgpu_windows *win, int32_t n_parts, flockgpu_partition_result *out);

/* `take` (arrow::compute::take as used by shuffle_writer.rs:131-141 and by HashJoinExec): out[i] = src[rows[i]].
 * All pointers are device pointers; `out` of the fixed-width variants is caller-allocated (n entries).
 * take_utf8 writes into ctx-arena buffers keyed by `slot` (0..15: one slot per column that must stay alive). */
int flockgpu_take_i32(flockgpu_ctx *ctx, const int32_t *src, const int32_t *rows, int64_t n, int32_t *out);
int flockgpu_take_i64(flockgpu_ctx *ctx, const int64_t *src, const int32_t *rows, int64_t n, int64_t *out);
int flockgpu_take_utf8(flockgpu_ctx *ctx, const flockgpu_utf8 *src, const int32_t *rows, int64_t n, int32_t slot,
                       flockgpu_utf8 *out, int64_t *out_bytes);
/* In-place inclusive prefix sum (rebuilds Arrow Utf8 offsets from received value lengths). */
int flockgpu_inclusive_scan_i32(flockgpu_ctx *ctx, int32_t *data /* device */, int64_t n);

/* ---- q11, user sessions: Window::Session(timeout) keyed by bidder
 * (benchmarks/src/nexmark/main.rs:120,346-349), the launcher's session walk over a run of epochs
 * (flock-function/src/aws/window/session.rs:64-178,205-262) and, per closed session,
 *   SELECT bidder, COUNT(*) AS bid_count, MIN(b_date_time) AS start_time, MAX(b_date_time) AS end_time
 *   FROM bid GROUP BY bidder                                        (benchmarks/src/nexmark/query/q11.sql)
 * Epoch t = rows [epoch_row_offsets[t], epoch_row_offsets[t+1]) of `bid` (only bidder and b_date_time are read;
 * b_date_time must be >= 0).  A bidder's bids of one epoch join its open session unless the first of them lies more
 * than `timeout_s` whole seconds after the session's last bid; after every epoch t the sessions whose last bid is more
 * than `timeout_s` whole seconds older than  base_time_ms/1000 + t  are closed.  Output: the sessions closed in epoch t
 * are rows [epoch_out_offsets[t], epoch_out_offsets[t+1]), ordered by bidder; two sessions of one bidder closed in
 * the same epoch are one row (the query groups by bidder); sessions still open after the last epoch are not reported. */
typedef struct {
    int32_t *bidder;                  /* device, ctx-owned */
    uint64_t *bid_count;              /* COUNT(*) -> UInt64 */
    int64_t *start_time, *end_time;   /* Timestamp(ms) */
    const int64_t *epoch_out_offsets; /* host, n_epochs + 1 */
    int64_t rows;
    int64_t sessions_total;           /* every session found, reported or not */
} flockgpu_q11_result;
int flockgpu_q11_user_sessions(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const int64_t *epoch_row_offsets /* host */,
                               int32_t n_epochs, int32_t timeout_s, int64_t base_time_ms, flockgpu_q11_result *out);

/* Rows grouped by key, arrival order kept inside a key -- the `repartition(.., HashDiff([key], distinct))` of the session
 * and global windows (flock-function/src/aws/window/session.rs:242-250): out_keys[i] = keys[out_rows[i]], ascending,
 * stable.  `keys` is a 16-byte aligned device column; the outputs are ctx-owned device arrays of `rows` entries. */
int flockgpu_group_rows_by_key(flockgpu_ctx *ctx, const int32_t *keys, int64_t rows, int32_t **out_keys, int32_t **out_rows);

/* ---- JSON lines -> columns (the decode step of `select_event_to_batches`,
 * flock/src/datasource/nexmark/nexmark.rs:180-205): the reference keeps an epoch's events as newline-delimited
 * serde_json objects (generator.rs:79-93) and decodes them with arrow's json::Reader against the relation's schema
 * (`event_bytes_to_batch`, flock/src/transmute.rs:255-266).  Every line is one object; members are matched to `fields`
 * by name in any order, other members are skipped; row = line.  Int32 / Int64 fields take JSON integer literals (a
 * Timestamp(ms) column is Int64 here), Utf8 fields take strings (all escapes, \uXXXX surrogate pairs included).
 * Errors name the first offending line: a missing field, malformed JSON or an integer outside the type ->
 * FLOCKGPU_ERR_INVALID; a number with a fraction / exponent, a blank line, an escape inside a KEY ->
 * FLOCKGPU_ERR_UNSUPPORTED.  `json`: device text, 16-byte aligned, below 2^31 bytes per call. */
enum { FLOCKGPU_JSON_INT32 = 0, FLOCKGPU_JSON_INT64 = 1, FLOCKGPU_JSON_UTF8 = 2 };
typedef struct {
    const char *name; /* host, at most 31 bytes */
    int32_t type;     /* FLOCKGPU_JSON_* */
} flockgpu_json_field;
typedef struct {
    void *values;       /* device, ctx-owned: int32_t / int64_t per row (NULL for a Utf8 field) */
    flockgpu_utf8 utf8; /* device, ctx-owned: offsets (rows + 1) and bytes of a Utf8 field */
    int64_t utf8_bytes;
} flockgpu_json_column;
int flockgpu_json_lines_decode(flockgpu_ctx *ctx, const uint8_t *json /* device */, int64_t n_bytes, const flockgpu_json_field *fields,
                               int32_t n_fields, flockgpu_json_column *out /* n_fields */, int64_t *rows);

/* ---- Arrow IPC body assembly: the body of the Arrow Flight data the reference builds for
 * every output batch (`flight_data_from_arrow_batch`, flock/src/transmute.rs:155-170,190-205; `Payload`,
 * flock/src/runtime/payload.rs:118-192) = the batch's buffers in field order, each padded with zeros to 8 bytes.
 * `buffers[i]` are device buffers (bytes = 0: an absent validity bitmap, contributes nothing); `out` is a device buffer
 * of `out_capacity` bytes (NULL: only the size is returned).  The header next to it is written on the host
 * (flock_amd/payload.py); compression stays on the CPU as in the reference (flock/src/encoding.rs:57-100). */
typedef struct {
    const void *data; /* device */
    int64_t bytes;
} flockgpu_ipc_buffer;
int flockgpu_ipc_pack_body(flockgpu_ctx *ctx, const flockgpu_ipc_buffer *buffers, int32_t n_buffers, uint8_t *out /* device */,
                           int64_t out_capacity, int64_t *out_bytes);

/* ---- Yahoo Streaming Benchmark: per Tumbling(10 s) window (benchmarks/src/ysb/main.rs:91)
 *   SELECT campaign_id, COUNT(*) FROM ad_event INNER JOIN campaign ON ad_id = c_ad_id WHERE event_type = lit
 *   GROUP BY campaign_id         (benchmarks/src/ysb/ysb.sql; flock/src/distributed_plan/planner.rs:298-346)
 * Columns are the projections of AdEvent / Campaign the query scans (flock/src/datasource/ysb/event.rs:24-87), Arrow Utf8.
 * Keys and the literal may be up to 40 bytes (UUIDs are 36); longer values -> FLOCKGPU_ERR_UNSUPPORTED.  Output rows:
 * (campaign_id Utf8, COUNT UInt64) of every campaign with at least one matching event, grouped by window. */
typedef struct {
    flockgpu_utf8 ad_id, event_type;
    int64_t rows;
} flockgpu_ysb_event_cols;
typedef struct {
    flockgpu_utf8 c_ad_id, campaign_id;
    int64_t rows;
} flockgpu_ysb_campaign_cols;
typedef struct {
    flockgpu_utf8 campaign_id;      /* device */
    const uint64_t *count;          /* device */
    const int64_t *win_out_offsets; /* host, n_windows + 1 */
    int64_t rows;
    int64_t campaign_bytes;
} flockgpu_ysb_result;
int flockgpu_ysb_campaign_counts(flockgpu_ctx *ctx, const flockgpu_ysb_event_cols *events, const flockgpu_windows *win,
                                 const flockgpu_ysb_campaign_cols *campaigns, const char *event_type_lit,
                                 flockgpu_ysb_result *out);
/* Device-side YSB source (flock/src/datasource/ysb/generator.rs:38-101 restated; every value is a pure function of
 * (seed, index): ysb_gen.hip).  Offsets arrays have rows + 1 entries; ad-id / campaign-id byte buffers hold 36 bytes per
 * row, the event_type buffer up to 8 bytes per event. */
int flockgpu_ysb_gen_campaigns(flockgpu_ctx *ctx, uint64_t seed, int64_t n_campaigns, int64_t ads, int32_t *c_ad_id_off,
                               uint8_t *c_ad_id_bytes, int32_t *campaign_off, uint8_t *campaign_bytes);
int flockgpu_ysb_gen_events(flockgpu_ctx *ctx, uint64_t seed, uint64_t first_event, int64_t n_events, int64_t n_ads,
                            int32_t *ad_id_off, uint8_t *ad_id_bytes, int32_t *event_type_off, uint8_t *event_type_bytes);

/* ---- device-side NEXMark source (flock/src/datasource/nexmark/{event,config,generator}.rs restated,
 * deviations D1-D4 documented in DESIGN.md).  Generates the columns the five plans scan for event
 * numbers [n0, n1) of a stream straight into HBM; any output pointer may be NULL to skip it. */
typedef struct {
    uint64_t seed, first_event_id, eps, base_time;
} flockgpu_nexmark_stream;
int flockgpu_nexmark_counts(const flockgpu_nexmark_stream *s, uint64_t n0, uint64_t n1,
                            uint64_t *n_person, uint64_t *n_auction, uint64_t *n_bid);
int flockgpu_nexmark_gen_bids(flockgpu_ctx *ctx, const flockgpu_nexmark_stream *s, uint64_t n0, uint64_t n1,
                              int32_t *auction, int32_t *bidder, int32_t *price, int64_t *b_date_time);
int flockgpu_nexmark_gen_auctions(flockgpu_ctx *ctx, const flockgpu_nexmark_stream *s, uint64_t n0, uint64_t n1,
                                  int32_t *a_id, int32_t *seller, int32_t *category);
int flockgpu_nexmark_gen_auction_times(flockgpu_ctx *ctx, const flockgpu_nexmark_stream *s, uint64_t n0, uint64_t n1,
                                       int64_t *a_date_time, int64_t *expires);
/* Persons: offsets arrays have rows + 1 entries; byte buffers must hold rows*14 / rows*13 / rows*2. */
int flockgpu_nexmark_gen_persons(flockgpu_ctx *ctx, const flockgpu_nexmark_stream *s, uint64_t n0, uint64_t n1,
                                 int32_t *p_id, int32_t *name_off, uint8_t *name_bytes, int32_t *city_off,
                                 uint8_t *city_bytes, int32_t *state_off, uint8_t *state_bytes);

#ifdef __cplusplus
}
#endif
#endif /* FLOCKGPU_H */
