// Shared device-side primitives of the operator kernels (defined in gather.hip):
//   * tile scan (count -> scan -> emit, scan.hpp)
//   * flag words -> row list
//   * exact per-window key range + sortedness of an int32 column
//   * take() for fixed-width and Utf8 columns (the `take` calls DataFusion's HashJoinExec issues to materialise
//     join output).  Utf8: lengths -> tile scan -> offsets + byte copy in one kernel.
#pragma once
#include <string>

#include "scan.hpp"

namespace flockgpu {

// out_rows[tile_base[tile] + i] = global row number of the i-th flagged row of the tile (row order).
// flag_words / counts as written by store_flags_and_counts (scan.hpp) over kFlagTile-row tiles of `st`.
int emit_flagged_rows(flockgpu_ctx *ctx, const SegTiles &st, const uint32_t *flag_words, const uint32_t *counts,
                      const uint64_t *tile_base, int32_t *out_rows);
// One segment of up to kSelfScanMaxTiles tiles: no tile scan in front -- every workgroup sums the lower tiles' counts itself, the last one stores
// {0, selected rows} into the PINNED h_off (read it after the stream synchronises).
int emit_flagged_rows_self(flockgpu_ctx *ctx, const SegTiles &st, const uint32_t *flag_words, const uint32_t *counts, int32_t *out_rows, int64_t *h_off);

// Copies (auction, price, bidder, b_date_time) of every flagged bid row to out_*[tile_base[tile] + i], row order kept
// (the Projection [auction, price, bidder, b_date_time] of q7 / q9 over the rows that survive the join).
int emit_flagged_bids(flockgpu_ctx *ctx, const SegTiles &st, const uint32_t *flag_words, const uint32_t *counts,
                      const uint64_t *tile_base, const flockgpu_bid_cols &bid, int32_t *o_auction, int32_t *o_price,
                      int32_t *o_bidder, int64_t *o_time);

// Exact per-segment minimum / maximum of `col` and whether the segment is strictly increasing (sorted and
// duplicate-free).  d_min / d_max / d_sorted: device arrays of st.n_seg entries, initialised by the call
// (INT32_MAX / INT32_MIN / 1); empty segments keep those values.  `st` must use kFlagTile-row tiles.
int segment_key_stats(flockgpu_ctx *ctx, const int32_t *col, int64_t n_rows, const SegTiles &st, int32_t *d_min,
                      int32_t *d_max, int32_t *d_sorted);

int gather_i32(flockgpu_ctx *ctx, const int32_t *src, const int32_t *rows, int64_t n, int32_t *out);
int gather_i64(flockgpu_ctx *ctx, const int64_t *src, const int32_t *rows, int64_t n, int64_t *out);
// Up to kGatherMulti fixed-width columns (4 or 8 bytes) taken at ONE row list in one launch: the row list is read once (four rows per lane, one
// 16-byte load), every column's four values are in flight together and leave as 16-byte stores.  A take of a bid's four columns was four
// launches that each re-read the list and kept one 4-byte load per lane in flight.  out[c] must be 16-byte aligned (arena buffers are).
constexpr int kGatherMulti = 8;
struct GatherCols {
    const void *src[kGatherMulti];
    void *out[kGatherMulti];
    int32_t width[kGatherMulti];   // 4 | 8
    int32_t n = 0;
};
int gather_fixed_multi(flockgpu_ctx *ctx, const GatherCols &cols, const int32_t *rows, int64_t n);
// The same take for a row list in NO order that names most of the relation's `in_rows` rows (ORDER BY): two to four columns of 16 bytes in all are
// interleaved into records first (scratch under `name`), so that the take reads one 16-byte record per row; anything else falls through to the plain take.
int gather_fixed_packed(flockgpu_ctx *ctx, const char *name, const GatherCols &cols, int64_t in_rows, const int32_t *rows, int64_t n);

// Gathers `n` Utf8 values in two phases so that several columns share ONE host synchronisation:
//   begin  : lengths -> tile scan; queues the D2H copy of the total byte count (the host needs it to size the
//            byte buffer; Arrow Utf8 offsets are int32, so it must also be checked against 2^31)
//   -- the caller synchronises the stream once --
//   finish : offsets + bytes.  out_off (n + 1 entries) and out_bytes live in the ctx arena under `name`.
struct Utf8Gather {
    std::string name;
    flockgpu_utf8 src{};
    const int32_t *rows = nullptr;
    int64_t n = 0, tiles = 0;
    uint32_t *counts = nullptr;
    uint64_t *tile_base = nullptr;
    uint64_t *h_total = nullptr;  // pinned
};
// d_n (device, may be null): the true length of `rows` when the host does not know it yet -- `n` is then an upper
// bound (buffers and grid are sized for it) and, once the host has read the true length, gather_utf8_narrow() sets it
// before finish.  The tiles of the bound beyond the true length count zero bytes.
int gather_utf8_begin(flockgpu_ctx *ctx, const char *name, const flockgpu_utf8 &src, const int32_t *rows, int64_t n,
                      Utf8Gather *g, const uint64_t *d_n = nullptr);
void gather_utf8_narrow(Utf8Gather *g, int64_t n);
int gather_utf8_finish(flockgpu_ctx *ctx, const Utf8Gather &g, flockgpu_utf8 *out, int64_t *n_bytes);
// begin + synchronise + finish for a single column.
int gather_utf8(flockgpu_ctx *ctx, const char *name, const flockgpu_utf8 &src, const int32_t *rows, int64_t n,
                flockgpu_utf8 *out, int64_t *n_bytes);

// Up to four Utf8 columns gathered with ONE row list (q3's name / city / state of the joined persons; the Utf8 columns of a relation
// in the exchange): one length pass that reads the row numbers once, ONE tile scan over all columns' tiles, one emit launch.
// Same two-phase protocol as the single-column gather (begin -- one synchronisation, shared with whatever else the caller waits
// for -- finish); outputs live in the ctx arena under `name`.
struct Utf8MultiGather {
    std::string name;
    int k = 0;
    flockgpu_utf8 src[4]{};
    const int32_t *rows = nullptr;
    int64_t n = 0, tiles = 0, tiles_stride = 0;
    uint32_t *counts = nullptr;
    uint64_t *tile_base = nullptr;
    uint64_t *h_col_base = nullptr;  // pinned, k + 1: where each column's bytes start in the scan over all columns
};
int gather_utf8_multi_begin(flockgpu_ctx *ctx, const char *name, const flockgpu_utf8 *srcs, int k, const int32_t *rows, int64_t n,
                            Utf8MultiGather *g, const uint64_t *d_n = nullptr);
// The wait between begin and finish when nothing else has to finish with it: polls the byte totals in pinned memory (common.hpp: wait_pinned).
int gather_utf8_multi_wait(flockgpu_ctx *ctx, const Utf8MultiGather &g);
// Between begin and finish: run_bytes[c][d] = bytes of column c over the send-order rows [run_start[d], run_start[d + 1]) (device arrays; run_start
// ascending, its last entry <= the rows of the take), from the take's scanned tile bases + the boundary tiles' lengths.  Queued, no wait.
int gather_utf8_multi_run_bytes(flockgpu_ctx *ctx, const Utf8MultiGather &g, const int64_t *d_run_start, int n_runs, unsigned long long *const *d_run_bytes);
void gather_utf8_multi_narrow(Utf8MultiGather *g, int64_t n);
int gather_utf8_multi_finish(flockgpu_ctx *ctx, const Utf8MultiGather &g, flockgpu_utf8 *outs, int64_t *n_bytes, const int64_t *known_bytes = nullptr);

// The same take with NO scan launch and NO host wait between lengths and bytes (a small-batch call is launch- and wait-bound:
// q3 at 1e8 events): the emit workgroups sum the lower tiles' byte counts themselves, the byte buffers are sized by `cap_bytes`
// (the caller's estimate: the previous call's totals plus slack), totals and an "estimate too small" flag land in pinned memory.
// Queue it, synchronise ONCE with whatever else the call waits for, then read h_tot[c] / *h_over: on overflow (or more than n_bound
// rows) redo the take with gather_utf8_multi_begin / finish.  out[c] are valid when *h_over == 0.
struct Utf8FastGather {
    std::string name;
    int k = 0;
    int64_t tiles = 0;
    flockgpu_utf8 out[4]{};
    const uint64_t *h_tot = nullptr;   // pinned, k byte totals
    uint32_t *h_over = nullptr;        // pinned
};
int gather_utf8_multi_fast(flockgpu_ctx *ctx, const char *name, const flockgpu_utf8 *srcs, int k, const int32_t *rows, int64_t n_bound,
                           const uint64_t *d_n, const int64_t *cap_bytes, Utf8FastGather *g);
// ---- small-scalar plumbing of the operators that end in a host wait (round 5).  One execute of a generic plan made 7 hipMemsetAsync
// and 11 small device-to-host hipMemcpyAsync calls (rocprofv3 --hip-trace: ~10 us of host time each, a 5.5 us copy kernel per
// memcpy on the stream) around 23 kernels; these two replace them:
// fill_words: up to four regions of 32-bit words, each with its own value, in ONE launch (error flags, totals, counters, chain heads);
// publish_words: up to four runs of <= 64 words copied into pinned host memory by one 64-thread kernel (read them after the
// stream synchronises).  (A tile scan's segment offsets need neither: launch_tile_scan takes a pinned pointer for them.)
struct FillList {
    void *p[4]{};
    uint32_t v[4]{};
    uint64_t words[4]{};
    int n = 0;
    FillList &add(void *ptr, uint32_t value, uint64_t n_words) {
        if (n < 4 && n_words) { p[n] = ptr; v[n] = value; words[n] = n_words; ++n; }
        return *this;
    }
};
int fill_words(flockgpu_ctx *ctx, const FillList &f);
struct PublishList {
    void *h[4]{};
    const void *d[4]{};
    int words[4]{};
    int n = 0;
    PublishList &add(void *h_pinned, const void *dev, int n_words) {
        if (n < 4 && n_words > 0) { h[n] = h_pinned; d[n] = dev; words[n] = n_words; ++n; }
        return *this;
    }
};
int publish_words(flockgpu_ctx *ctx, const PublishList &l);

// Utf8 gather whose total byte count the caller already knows (a permutation of a column of known size): no host wait.
int gather_utf8_finish_known(flockgpu_ctx *ctx, Utf8Gather &g, int64_t total_bytes, flockgpu_utf8 *out);

// flockgpu_partition_by_key without its host wait (shuffle.hip): rows grouped by (destination, window) on the device, the
// n_parts * n_win + 1 group offsets on the device and queued into pinned memory (valid after the next stream synchronisation).
// `payload`: up to four 4-byte columns whose values the emit pass writes in send order next to the row numbers (it has the
// tile's rows at hand: the separate `take` per column -- 12 B of traffic per row and column -- is not needed for them).
struct PartPayload {
    int32_t n = 0;
    const int32_t *src[4] = {};
    int32_t *dst[4] = {};  // rows of every window entries each, caller-owned
    bool skip_rows = false;  // the caller needs no row numbers (every column it moves is a payload column)
};
int partition_by_key_async(flockgpu_ctx *ctx, const int32_t *keys, int64_t rows, const flockgpu_windows *win, int32_t n_parts,
                           const int32_t **d_rows, const int64_t **d_group_off, const int64_t **h_group_off, int64_t *n_out,
                           const PartPayload *payload = nullptr, const char *cache_name = nullptr);

// q3's stage 0 for the in-library exchange (q3.hip; planner.rs:152-171: FilterExec category = lit / state = a OR b OR ... BEFORE the
// hash repartition): the rows each filter keeps, in input order, with the windows' new row offsets.  Synchronises once for both.
struct Q3Stage0 {
    const int32_t *auction_rows = nullptr, *person_rows = nullptr;   // device
    std::vector<int64_t> auction_off, person_off;                    // host, n_windows + 1
    int64_t n_auctions = 0, n_persons = 0;
};
int q3_stage0_filters(flockgpu_ctx *ctx, const flockgpu_auction_cols *auction, const flockgpu_windows *auction_win,
                      const flockgpu_person_cols *person, const flockgpu_windows *person_win, int64_t category_lit,
                      const char *const *state_lits, int n_state_lits, Q3Stage0 *out);

// q8.dag's Partial DISTINCT for the in-library exchange (shuffle.hip): the first occurrence of every key inside each 8192-row tile of a
// window (a tile = one input partition of HashAggregateExec(Partial) with no aggregates), as a compact key column in input order with the
// windows' new row offsets.  Keys may repeat across tiles (and INT32_MIN, the table's empty mark, is never deduplicated): the
// FinalPartitioned DISTINCT on the receiving side removes what is left.  Synchronises once.
int tile_distinct_i32(flockgpu_ctx *ctx, const char *name, const int32_t *keys, int64_t rows, const flockgpu_windows *win,
                      const int32_t **out_keys, std::vector<int64_t> *out_win_off, int64_t *n_out);

// q5.dag's Partial stage for the in-library exchange (q5.hip): COUNT GROUP BY auction per 8192-row TILE of a pane -- a tile plays
// the part of one input partition of HashAggregateExec(Partial), whose RoundRobin-fed partitions each emit their own groups in the
// reference, too -- written as (auction, count) pairs into the pane's region of ONE pass over the bids: no global counters, no
// clear, no compaction passes.  Region p starts at offsets[2p] and holds offsets[2p + 1] - offsets[2p] pairs; [offsets[2p + 1],
// offsets[2p + 2]) is unused space (a pane's region is as large as the pane: pairs <= rows).  Synchronises once.
struct Q5TilePartial {
    const int32_t *auction = nullptr;   // device, `capacity` slots
    const uint32_t *count = nullptr;
    int64_t capacity = 0;
    std::vector<int64_t> offsets;       // host, 2 * n_panes + 1
    int64_t pairs = 0;
};
int q5_partial_by_tile(flockgpu_ctx *ctx, const flockgpu_bid_cols *bid, const flockgpu_windows *win, Q5TilePartial *out);

// In-place inclusive scan of n int32 values; `name` keys the scan-state arena buffers.
int inclusive_scan_i32(flockgpu_ctx *ctx, const char *name, int32_t *data, int64_t n);

}  // namespace flockgpu
