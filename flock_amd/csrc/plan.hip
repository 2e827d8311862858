// Plan-level C ABI (include/flockgpu_plan.h): serde_json physical plan in, Arrow C Data Interface batches in / out.
//   feed_data_sources  flock/src/runtime/context.rs:257-325   -> flockgpu_plan_feed (pinned or staged H2D, no host wait)
//   execute            flock/src/runtime/context.rs:172-191   -> flockgpu_plan_execute
//   execute_partitioned context.rs:197-216 (chosen by is_shuffling, context.rs:328-337; flock-function/src/aws/actor.rs:60-66)
//                                                             -> flockgpu_plan_execute_partitioned
//   clean_data_sources context.rs:227-254                     -> flockgpu_plan_reset
// The plan is parsed into the operator tree of plan_ir.hpp.  Sub-trees that are one of the NEXMark pipelines run as the
// fused kernels of flockgpu.h (q2 filter, q3 / q8 / q13 joins, q5 / q7 aggregates, the Partial COUNT of q5.dag);
// every other node -- the STAGE plans either side of a `RepartitionExec Hash` -- runs on the generic operators of
// relops.hpp.  Host-side code only; all compute goes through those two layers.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <map>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>

#include "../../include/flockgpu_plan.h"
#include "plan_ir.hpp"
#include "pred.hpp"
#include "valprog.hpp"

using namespace flockgpu;
using namespace flockgpu::ir;

namespace {
#ifdef FLOCKGPU_EXPERIMENTAL
// (A/B builds: wall time per entry point, printed by flockgpu_plan_ring_close when FLOCKGPU_PLAN_TIMES is set)
struct ApiClock {
    const char *what;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    static std::map<std::string, std::pair<double, long>> &acc() {
        static std::map<std::string, std::pair<double, long>> m;
        return m;
    }
    explicit ApiClock(const char *w) : what(w) {}
    ~ApiClock() {
        auto &e = acc()[what];
        e.first += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        e.second += 1;
    }
    static void dump() {
        if (!flockgpu::exp_env("FLOCKGPU_PLAN_TIMES")) return;
        for (auto &kv : acc()) fprintf(stderr, "[plan times] %-28s %8ld calls %10.1f us/call\n", kv.first.c_str(), kv.second.second, kv.second.first / std::max(1L, kv.second.second));
        acc().clear();
    }
};
#define API_CLOCK(name) ApiClock api_clock_(name)
#define API_CLOCK2(var, name) ApiClock var(name)
#else
#define API_CLOCK(name) do {} while (0)
#define API_CLOCK2(var, name) do {} while (0)
#endif


// ------------------------------------------------------------------ pinned host memory, pooled
// Output batches live in pinned host memory (one block per execute); blocks return to a process-wide pool when the
// consumer calls the Arrow release callback (possibly from another thread, possibly after the ctx is gone).
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<size_t, void *>> free_blocks;
    size_t held = 0;
    static constexpr size_t kMaxHeld = size_t(2) << 30;
    void *get(size_t bytes, size_t *cap) {
        {
            std::lock_guard<std::mutex> g(mu);
            size_t best = free_blocks.size();
            for (size_t i = 0; i < free_blocks.size(); ++i)
                if (free_blocks[i].first >= bytes && free_blocks[i].first <= bytes * 2 + (1 << 20) &&
                    (best == free_blocks.size() || free_blocks[i].first < free_blocks[best].first))
                    best = i;
            if (best != free_blocks.size()) {
                auto b = free_blocks[best];
                free_blocks.erase(free_blocks.begin() + (long)best);
                held -= b.first;
                *cap = b.first;
                return b.second;
            }
        }
        size_t want = std::max<size_t>(bytes + bytes / 8, 4096);
        want = (want + 4095) & ~size_t(4095);
        void *p = nullptr;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
        *cap = want;
        return p;
    }
    void put(void *p, size_t cap) {
        {
            std::lock_guard<std::mutex> g(mu);
            if (held + cap <= kMaxHeld) {
                free_blocks.emplace_back(cap, p);
                held += cap;
                return;
            }
        }
        (void)hipHostFree(p);
    }
};
PinnedPool &pool() {
    static PinnedPool *p = new PinnedPool();  // never destroyed: release callbacks may run during process exit
    return *p;
}
struct HostBlock {  // shared by the partitions of one execute
    void *ptr = nullptr;
    size_t cap = 0;
    ~HostBlock() {
        if (ptr) pool().put(ptr, cap);
    }
};

// ------------------------------------------------------------------ tables
struct TCol {
    DevColumn c;
    bool present = false;
    // every value of this column is a value of the column at `subset_of` (it was taken from it by a filter, a join or as a GROUP BY's keys): when
    // that column is a leaf's, its cached minimum / maximum BOUND this one's -- enough to size the dense paths without a pass and a wait of its own
    const void *subset_of = nullptr;
};
struct Table {
    std::vector<TCol> cols;
    int64_t rows = 0;
};

struct DevBuf {  // a leaf column on the device, grown by feeds
    void *values = nullptr;  // fixed-width values or Utf8 bytes
    int32_t *offsets = nullptr;
    int64_t bytes = 0;
    // NULLs that may reach an output or an aggregate (round 4): one validity byte per row, materialised from the first batch that
    // holds such a NULL on; rows [0, valid_rows) are covered (the rest -- batches without NULLs -- is filled with 1 when the leaf is scanned)
    uint8_t *valid = nullptr;
    int64_t valid_rows = 0;
};
struct LeafData {
    std::vector<DevBuf> cols;
    int64_t rows = 0;
    int64_t dropped = 0;     // rows a feed left out because of NULLs
    bool borrowed = false;   // cols alias another plan's leaf (flockgpu_plan_feed_shared): nothing here is this leaf's to grow
    // pane ring (flockgpu_plan_ring_open): rows -- and per Utf8 column, bytes -- of every pane whose rows this leaf still holds, oldest first
    std::vector<int64_t> pane_rows;
    std::vector<std::vector<int64_t>> pane_bytes;   // [pane][column]
};

enum Fused { kNone = 0, kQ2, kQ3, kQ5, kQ7, kQ8, kQ13, kPartialCount, kQ9, kQ4, kYsb };
struct FusedInfo {
    Fused kind = kNone;
    int leaf_a = -1, leaf_b = -1;     // the scans below
    std::vector<int> a_cols, b_cols;  // leaf columns handed to the fused entry point, in its argument order (-1: unused)
    std::vector<int> out_map;         // node output column -> fused output slot (-1: not produced)
    int64_t lit = 0;                  // q2 modulus / q3 category
    std::vector<std::string> strs;    // q3 state literals
};

constexpr int kStageLanes = 8;                     // at most this many host threads fill pinned chunks side by side (four do; FLOCKGPU_STAGE_LANES in experimental builds)
constexpr int kStageChunks = kStageLanes * 2;      // two chunks per lane: one is filled while the other is in flight
constexpr size_t kStageChunk = size_t(4) << 20;

}  // namespace

struct flockgpu_plan {
    flockgpu_ctx *ctx = nullptr;
    ir::Plan ir;
    std::vector<LeafData> leaves;
    std::vector<FusedInfo> fused;  // by node id
    int query = 0;                 // NEXMark query number when the whole plan is one fused pipeline, else 0
    std::string description;
    bool generic_only = false;
    // pageable feeds go through a ring of pinned chunks: the host copies into chunk k + 1 while chunk k is in flight
    void *stage[kStageChunks] = {};
    hipEvent_t stage_done[kStageChunks] = {};
    int stage_next[kStageLanes] = {};
    struct CopyJob { void *dst; const void *src; size_t bytes; };
    std::vector<CopyJob> jobs;   // the pageable copies of the feed in progress
    std::vector<CopyJob> runs;   // transfers of the feed in progress, merged while they stay contiguous (h2d)
    int64_t fed_bytes = 0;
    Table retained;              // flockgpu_plan_execute_retain: the result, left on the device for the plans that consume it
    bool has_retained = false;
    // ---- device-side pane ring (flockgpu_plan.h): the last ring_ppw panes stay in HBM across executes
    int ring_ppw = 0;            // panes per window; 0: no ring, whole-window feeding
    int64_t ring_first = 0;      // id of the oldest pane held
    int ring_n = 0;              // panes held
    // q5 (COUNT GROUP BY auction -> MAX -> join): a pane is retained as its Partial aggregate state, the (auction, count) groups, in
    // `ring.q5a` / `ring.q5c` (pane after pane); the leaf holds only the newest pane's rows, until the next pane begins
    bool ring_q5 = false;
    std::vector<int64_t> ring_groups;   // groups per held pane (the newest pane's entry is valid when ring_newest_done)
    bool ring_newest_done = false;
    // flockgpu_plan_prefetch_pane: the NEXT pane's bytes on their way into side buffers (a stream of their own, host staging on threads of
    // their own) while the current window executes; flockgpu_plan_feed_pane(.., NULL, NULL, 0) for that pane appends them device to device
    struct Prefetch {
        bool active = false;
        int input = -1;
        int64_t pane = 0, rows = 0;
        std::vector<void *> dev;        // per leaf column: the side buffer (null: not prefetched); a Utf8 column's: its bytes
        std::vector<void *> dev_off;    // Utf8 columns: the pane's `rows` raw END offsets, batch by batch (rebased when the pane is appended)
        std::vector<int64_t> bytes;     // Utf8 columns: bytes in the side buffer
        struct Rebase { int col; int64_t row0, n; int64_t delta; };   // rows [row0, row0 + n) of column col: + delta turns a raw offset into a pane-relative one
        std::vector<Rebase> rebase;
        std::vector<std::thread> workers;
        std::vector<int> rc;
    } pre;
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done = nullptr, append_done = nullptr;   // the pane's copies are queued / its side buffers have been read
    bool copy_done_set = false, append_done_set = false;
    int32_t *ring_auction = nullptr;
    uint32_t *ring_count = nullptr;
    // ---- flockgpu_plan_execute_async: the finished call's outputs, handed over by flockgpu_plan_wait
    bool async_pending = false;
    ArrowSchema async_schema{};
    std::vector<ArrowArray> async_batches;
    int async_n = 0;
    // ---- exact (min, max) of integer LEAF columns, computed when an operator first asks (the dense GROUP BY / join paths size their
    // direct-address tables from it) and kept while the leaf's contents stay what they were: every feed / reset / ring step drops them.
    // The reference's arch harness feeds once and executes ten times (flock-function/src/aws/arch/source.rs:25-65); a streaming host
    // pays one 4-byte-per-row pass per fed window.
    struct ColStat { int64_t rows, mn, mx; };
    std::map<const void *, ColStat> col_stats;
    // ---- common sub-plans: the signature of every Aggregate sub-tree that occurs more than once in the plan (q5's SQL names its
    // COUNT(*) GROUP BY auction subquery in both join inputs, q5_plan.fmt:6,13) and the scan leaves underneath it, by node id; an execute
    // runs such a sub-tree once per (signature, leaves its scans resolve to) and hands the table to its twins (Exec::memo)
    std::vector<std::string> twin_sig;               // empty: the node has no twin
    std::vector<std::vector<int>> twin_leaves;
    // ---- hash-placement guard: the scheme tag of the co-partitioned inputs fed since the last reset (-1: none fed yet)
    int scheme_state = -1;       // 0: untagged batches, 1: tagged with scheme_tag
    std::string scheme_tag;
};

namespace {

// ------------------------------------------------------------------ fused-pipeline recognition (structure only)
struct Peeled {
    const Node *n = nullptr;
    std::vector<int> map;  // output column of the peeled-from node -> column of n (-1: computed)
};
// Follows projections that only select / rename columns and hash repartitions (no effect on the row multiset).
Peeled peel(const Node *n) {
    Peeled p;
    p.n = n;
    p.map.resize(n->schema.size());
    for (size_t i = 0; i < p.map.size(); ++i) p.map[i] = (int)i;
    for (;;) {
        if (p.n->kind == NKind::Repartition) {
            p.n = p.n->in[0].get();
            continue;
        }
        if (p.n->kind == NKind::Project) {
            bool pure = true;
            for (auto &e : p.n->proj) pure = pure && e.first->kind == EKind::Col;
            if (!pure) return p;
            for (auto &m : p.map)
                if (m >= 0) m = p.n->proj[(size_t)m].first->col;
            p.n = p.n->in[0].get();
            continue;
        }
        return p;
    }
}
const Expr *uncast(const Expr *e) {
    while (e && e->kind == EKind::Cast && (e->cast_to == ColType::I64 || e->cast_to == ColType::F64)) e = e->l.get();
    return e;
}
bool is_bin(const Expr *e, const char *op) { return e && e->kind == EKind::Bin && e->s == op; }

// Logical aggregate: Final / FinalPartitioned over (peeled) Partial with the same shape, or a single stage.
// Returns the node below the lowest aggregate stage and the group / argument columns in ITS schema.
struct LogicalAgg {
    const Node *below = nullptr;
    std::vector<int> group;
    std::vector<std::string> fns;
    std::vector<int> args;
};
bool logical_agg(const Node *n, LogicalAgg *out) {
    if (n->kind != NKind::Aggregate) return false;
    const Node *low = n;
    if (n->mode != "Partial") {
        Peeled p = peel(n->in[0].get());
        if (p.n->kind == NKind::Aggregate && p.n->mode == "Partial" && p.n->group.size() == n->group.size() && p.n->aggs.size() == n->aggs.size()) {
            // the final stage must read the partial stage's columns in place
            for (size_t g = 0; g < n->group.size(); ++g)
                if (p.map[(size_t)n->group[g]] != (int)g) return false;
            int state_at = (int)n->group.size();
            for (size_t a = 0; a < n->aggs.size(); ++a) {
                if (p.map[(size_t)n->aggs[a].arg] != state_at || p.n->aggs[a].fn != n->aggs[a].fn) return false;
                if (n->aggs[a].arg2 >= 0 && p.map[(size_t)n->aggs[a].arg2] != state_at + 1) return false;
                state_at += agg_state_cols(n->aggs[a].fn);
            }
            low = p.n;
        } else {
            return false;  // a lone final stage (a stage plan): generic
        }
    }
    out->below = low->in[0].get();
    out->group = low->group;
    for (auto &a : low->aggs) {
        out->fns.push_back(a.fn);
        out->args.push_back(a.arg);
    }
    return true;
}
const Node *as_scan(const Node *n, std::vector<int> *map) {
    Peeled p = peel(n);
    if (p.n->kind != NKind::Scan) return nullptr;
    if (map) *map = p.map;
    return p.n;
}
bool req_subset(const Node *n, const std::vector<int> &allowed) {
    for (size_t i = 0; i < n->required.size(); ++i)
        if (n->required[i] && std::find(allowed.begin(), allowed.end(), (int)i) == allowed.end()) return false;
    return true;
}

// Q of q4.sql / q9.sql: MAX(price) per auction [, category] over auction JOIN bid ON a_id = auction WHERE b_date_time BETWEEN
// a_date_time AND expires (planner.rs:218-256 stages 0-2; BETWEEN arrives as `x >= lo AND x <= hi`).
struct WinningBids {
    const Node *auction_scan = nullptr, *bid_scan = nullptr;
    int a_id = -1, a_dt = -1, a_exp = -1, a_cat = -1;  // columns of the auction leaf
    int b_auction = -1, b_price = -1, b_dt = -1;       // columns of the bid leaf
};
bool match_winning_bids(const Node *agg, bool with_category, WinningBids *w) {
    LogicalAgg la;
    if (!logical_agg(agg, &la) || la.fns != std::vector<std::string>{"max"} || la.group.size() != (with_category ? 2u : 1u)) return false;
    Peeled pf = peel(la.below);
    if (pf.n->kind != NKind::Filter) return false;
    Peeled pj = peel(pf.n->in[0].get());
    if (pj.n->kind != NKind::Join || pj.n->on_l2 >= 0) return false;
    const Node *J = pj.n;
    const int nl = (int)J->in[0]->schema.size();
    auto to_join = [&](int below_col) {  // column of la.below -> column of the join's output
        const int f = pf.map[(size_t)below_col];
        return f < 0 ? -1 : pj.map[(size_t)f];
    };
    // b_date_time >= a_date_time AND b_date_time <= expires (either order of the conjuncts)
    const Expr *p = pf.n->pred.get();
    if (!is_bin(p, "And")) return false;
    int x = -1, lo = -1, hi = -1;
    for (const Expr *c : {p->l.get(), p->r.get()}) {
        if (!c || c->kind != EKind::Bin) return false;
        const Expr *a = uncast(c->l.get()), *b = uncast(c->r.get());
        if (a->kind != EKind::Col || b->kind != EKind::Col) return false;
        int cx, bound;
        bool is_lo;
        if (c->s == "GtEq") { cx = a->col; bound = b->col; is_lo = true; }
        else if (c->s == "LtEq") { cx = a->col; bound = b->col; is_lo = false; }
        else return false;
        if (x >= 0 && x != cx) return false;
        x = cx;
        (is_lo ? lo : hi) = bound;
    }
    if (x < 0 || lo < 0 || hi < 0) return false;
    x = pj.map[(size_t)x]; lo = pj.map[(size_t)lo]; hi = pj.map[(size_t)hi];
    const int key = to_join(la.group[0]), cat = with_category ? to_join(la.group[1]) : -1, price = to_join(la.args[0]);
    if (x < nl || lo < 0 || lo >= nl || hi < 0 || hi >= nl || price < nl || key != J->on_l || (with_category && (cat < 0 || cat >= nl))) return false;
    std::vector<int> lmap, rmap;
    w->auction_scan = as_scan(J->in[0].get(), &lmap);
    w->bid_scan = as_scan(J->in[1].get(), &rmap);
    if (!w->auction_scan || !w->bid_scan) return false;
    w->a_id = lmap[(size_t)J->on_l];
    w->a_dt = lmap[(size_t)lo];
    w->a_exp = lmap[(size_t)hi];
    w->a_cat = with_category ? lmap[(size_t)cat] : -1;
    w->b_auction = rmap[(size_t)J->on_r];
    w->b_price = rmap[(size_t)(price - nl)];
    w->b_dt = rmap[(size_t)(x - nl)];
    auto is = [](const Node *scan, int c, ColType t, bool ts) { return c >= 0 && scan->schema[(size_t)c].type == t && scan->schema[(size_t)c].is_ts == ts; };
    return is(w->auction_scan, w->a_id, ColType::I32, false) && is(w->auction_scan, w->a_dt, ColType::I64, true) &&
           is(w->auction_scan, w->a_exp, ColType::I64, true) && (!with_category || is(w->auction_scan, w->a_cat, ColType::I32, false)) &&
           is(w->bid_scan, w->b_auction, ColType::I32, false) && is(w->bid_scan, w->b_price, ColType::I32, false) && is(w->bid_scan, w->b_dt, ColType::I64, true);
}

void recognise_fused(flockgpu_plan *pl, const Node *n) {
    for (auto &c : n->in) recognise_fused(pl, c.get());
    FusedInfo &fi = pl->fused[(size_t)n->id];
    if (pl->generic_only) return;
    auto leaf_schema = [&](const Node *scan) -> const std::vector<Field> & { return pl->ir.leaves[(size_t)scan->leaf].schema; };
    if (n->kind == NKind::Filter) {
        // ---- q2: [cast] col % m = 0 over a scan; the engine emits (that column, one more Int32 column)
        std::vector<int> map;
        const Node *scan = as_scan(n->in[0].get(), &map);
        const Expr *p = n->pred.get();
        if (scan && is_bin(p, "Eq") && uncast(p->r.get())->kind == EKind::LitI && uncast(p->r.get())->i == 0) {
            const Expr *m = uncast(p->l.get());
            if (is_bin(m, "Modulo") && uncast(m->l.get())->kind == EKind::Col && uncast(m->r.get())->kind == EKind::LitI) {
                const int kc = uncast(m->l.get())->col;
                const int64_t mod = uncast(m->r.get())->i;
                int other = -1, extra = 0;
                for (size_t i = 0; i < n->required.size(); ++i)
                    if (n->required[i] && (int)i != kc) { other = (int)i; ++extra; }
                const bool ok_types = n->schema[(size_t)kc].type == ColType::I32 && (other < 0 || n->schema[(size_t)other].type == ColType::I32);
                if (mod != 0 && extra <= 1 && ok_types && map[(size_t)kc] >= 0 && (other < 0 || map[(size_t)other] >= 0)) {
                    fi.kind = kQ2;
                    fi.leaf_a = scan->leaf;
                    fi.a_cols = {map[(size_t)kc], other < 0 ? map[(size_t)kc] : map[(size_t)other]};
                    fi.out_map.assign(n->schema.size(), -1);
                    fi.out_map[(size_t)kc] = 0;
                    if (other >= 0) fi.out_map[(size_t)other] = 1;
                    fi.lit = mod;
                }
            }
        }
        return;
    }
    if (n->kind == NKind::Aggregate) {
        // ---- Partial COUNT GROUP BY one Int32 column of a scan (stage 0 of q5.dag)
        std::vector<int> map;
        const Node *scan = as_scan(n->in[0].get(), &map);
        if (scan && n->mode == "Partial" && n->group.size() == 1 && n->aggs.size() == 1 && n->aggs[0].fn == "count" &&
            n->in[0]->schema[(size_t)n->group[0]].type == ColType::I32 && map[(size_t)n->group[0]] >= 0) {
            fi.kind = kPartialCount;
            fi.leaf_a = scan->leaf;
            fi.a_cols = {map[(size_t)n->group[0]]};
            fi.out_map = {0, 1};
            return;
        }
        LogicalAgg top;
        if (n->mode == "Partial" || !logical_agg(n, &top) || top.group.size() != 1 || top.fns.size() != 1) return;
        // ---- q4: AVG(final) GROUP BY category over Q (q4.sql; planner.rs:218-256)
        if (top.fns[0] == "avg") {
            Peeled pq = peel(top.below);
            WinningBids w;
            if (match_winning_bids(pq.n, true, &w) && pq.map[(size_t)top.group[0]] == 1 && pq.map[(size_t)top.args[0]] == 2) {
                fi.kind = kQ4;
                fi.leaf_a = w.auction_scan->leaf;
                fi.leaf_b = w.bid_scan->leaf;
                fi.a_cols = {w.a_id, w.a_cat, w.a_dt, w.a_exp};
                fi.b_cols = {w.b_auction, w.b_price, w.b_dt};
                fi.out_map = {0, 1};
            }
            return;
        }
        // ---- YSB: COUNT(*) GROUP BY campaign_id over Filter(event_type = lit)(ad_event) JOIN campaign ON ad_id = c_ad_id
        // (ysb.sql; planner.rs:298-346)
        if (top.fns[0] == "count" && top.below->schema[(size_t)top.group[0]].type == ColType::UTF8) {
            Peeled pj = peel(top.below);
            if (pj.n->kind != NKind::Join || pj.n->on_l2 >= 0) return;
            const Node *J = pj.n;
            const int nl = (int)J->in[0]->schema.size();
            Peeled pl_ = peel(J->in[0].get());
            std::vector<int> lmap, rmap;
            const Node *rs = as_scan(J->in[1].get(), &rmap);
            if (pl_.n->kind != NKind::Filter || !rs) return;
            const Node *ls = as_scan(pl_.n->in[0].get(), &lmap);
            const Expr *p = pl_.n->pred.get();
            if (!ls || !is_bin(p, "Eq") || p->l->kind != EKind::Col || p->r->kind != EKind::LitS || p->r->s.size() > 40) return;
            const int grp = pj.map[(size_t)top.group[0]];
            const int ev_key = pl_.map[(size_t)J->on_l];
            if (grp < nl || ev_key < 0) return;
            const int ad = lmap[(size_t)ev_key], ty = lmap[(size_t)p->l->col], cad = rmap[(size_t)J->on_r], camp = rmap[(size_t)(grp - nl)];
            auto text = [](const Node *scan, int c) { return c >= 0 && scan->schema[(size_t)c].type == ColType::UTF8; };
            if (!text(ls, ad) || !text(ls, ty) || !text(rs, cad) || !text(rs, camp)) return;
            fi.kind = kYsb;
            fi.leaf_a = ls->leaf;
            fi.leaf_b = rs->leaf;
            fi.a_cols = {ad, ty};
            fi.b_cols = {cad, camp};
            fi.strs = {p->r->s};
            fi.out_map = {0, 1};
        }
        return;
    }
    if (n->kind != NKind::Join) return;
    const Node *L = n->in[0].get(), *R = n->in[1].get();
    const size_t nl = L->schema.size();
    const Field &lk = L->schema[(size_t)n->on_l], &rk = R->schema[(size_t)n->on_r];
    if (n->on_l2 >= 0) {
        // ---- q9: bid JOIN Q ON auction = id AND price = final (q9.sql, q9_plan.fmt)
        std::vector<int> lmap;
        const Node *ls = as_scan(L, &lmap);
        Peeled pr = peel(R);
        WinningBids w;
        if (!ls || !match_winning_bids(pr.n, false, &w)) return;
        const auto &bs = pl->ir.leaves[(size_t)ls->leaf].schema;
        static const char *names[4] = {"auction", "bidder", "price", "b_date_time"};
        bool ok = bs.size() == 4 && lmap.size() == 4 && n->required.size() == 6;
        for (int i = 0; ok && i < 4; ++i)
            ok = bs[(size_t)i].name == names[i] && lmap[(size_t)i] == i && n->required[(size_t)i] && bs[(size_t)i].type == (i == 3 ? ColType::I64 : ColType::I32);
        // the pairs in either order: (auction, id = the group key) and (price, final = the maximum)
        int k_auction = n->on_l, k_id = pr.map[(size_t)n->on_r], k_price = n->on_l2, k_final = pr.map[(size_t)n->on_r2];
        if (k_auction == 2) { std::swap(k_auction, k_price); std::swap(k_id, k_final); }
        ok = ok && k_auction == 0 && k_price == 2 && k_id == 0 && k_final == 1;
        // the inner bid scan must be the same relation (the SQL scans `bid` twice)
        const auto &inner = pl->ir.leaves[(size_t)w.bid_scan->leaf].schema;
        ok = ok && inner[(size_t)w.b_auction].name == "auction" && inner[(size_t)w.b_price].name == "price" && inner[(size_t)w.b_dt].name == "b_date_time";
        if (ok) {
            fi.kind = kQ9;
            fi.leaf_a = w.auction_scan->leaf;
            fi.leaf_b = ls->leaf;
            fi.a_cols = {w.a_id, -1, w.a_dt, w.a_exp};
            fi.out_map = {0, 1, 2, 3, 0, 2};  // id equals auction, final equals price on every output row
        }
        return;
    }
    // ---- q3: Filter(int col = lit)(scan) JOIN Filter(utf8 col = a OR ...)(scan) on Int32 keys
    {
        Peeled pl_ = peel(L), pr = peel(R);
        if (pl_.n->kind == NKind::Filter && pr.n->kind == NKind::Filter && lk.type == ColType::I32 && rk.type == ColType::I32) {
            std::vector<int> lmap, rmap;
            const Node *ls = as_scan(pl_.n->in[0].get(), &lmap), *rs = as_scan(pr.n->in[0].get(), &rmap);
            const Expr *lp = pl_.n->pred.get();
            bool ok = ls && rs && is_bin(lp, "Eq") && uncast(lp->l.get())->kind == EKind::Col && uncast(lp->r.get())->kind == EKind::LitI;
            int cat_col = -1, state_col = -1;
            std::vector<std::string> states;
            if (ok) {
                cat_col = uncast(lp->l.get())->col;
                ok = pl_.n->schema[(size_t)cat_col].type == ColType::I32;
                std::vector<const Expr *> stack{pr.n->pred.get()};
                while (!stack.empty() && ok) {
                    const Expr *e = stack.back();
                    stack.pop_back();
                    if (is_bin(e, "Or")) { stack.push_back(e->r.get()); stack.push_back(e->l.get()); }
                    else if (is_bin(e, "Eq") && e->l->kind == EKind::Col && e->r->kind == EKind::LitS && (state_col < 0 || state_col == e->l->col) &&
                             pr.n->schema[(size_t)e->l->col].type == ColType::UTF8) { state_col = e->l->col; states.push_back(e->r->s); }
                    else ok = false;
                }
                ok = ok && !states.empty() && states.size() <= 8;
            }
            if (ok) {
                // required outputs: at most one Int32 column of the left side (-> the a_id slot), Utf8 columns of the right
                // side: the filter column (-> state slot) and up to two more (-> name, city slots)
                int a_id = -1;
                std::vector<int> texts;
                for (size_t i = 0; i < n->required.size() && ok; ++i) {
                    if (!n->required[i]) continue;
                    if (i < nl) {
                        const int c = pl_.map[i];
                        if (c < 0 || pl_.n->schema[(size_t)c].type != ColType::I32 || (a_id >= 0 && a_id != c)) ok = false;
                        else a_id = c;
                    } else {
                        const int c = pr.map[i - nl];
                        if (c < 0 || pr.n->schema[(size_t)c].type != ColType::UTF8) ok = false;
                        else if (c != state_col && std::find(texts.begin(), texts.end(), c) == texts.end()) texts.push_back(c);
                    }
                }
                ok = ok && texts.size() <= 2;
                const int lkey = pl_.map[(size_t)n->on_l], rkey = pr.map[(size_t)n->on_r];
                ok = ok && lkey >= 0 && rkey >= 0;
                if (ok) {
                    if (a_id < 0) a_id = lkey;
                    while (texts.size() < 2) texts.push_back(state_col);
                    auto to_leaf = [](const std::vector<int> &m, int c) { return m[(size_t)c]; };
                    fi.kind = kQ3;
                    fi.leaf_a = ls->leaf;
                    fi.leaf_b = rs->leaf;
                    fi.a_cols = {to_leaf(lmap, a_id), to_leaf(lmap, lkey), to_leaf(lmap, cat_col)};            // a_id, seller, category
                    fi.b_cols = {to_leaf(rmap, rkey), to_leaf(rmap, texts[0]), to_leaf(rmap, texts[1]), to_leaf(rmap, state_col)};  // p_id, name, city, state
                    ok = std::all_of(fi.a_cols.begin(), fi.a_cols.end(), [](int c) { return c >= 0; }) &&
                         std::all_of(fi.b_cols.begin(), fi.b_cols.end(), [](int c) { return c >= 0; });
                    fi.out_map.assign(n->schema.size(), -1);
                    for (size_t i = 0; i < n->required.size(); ++i) {
                        if (!n->required[i]) continue;
                        if (i < nl) fi.out_map[i] = 3;  // a_id slot
                        else {
                            const int c = pr.map[i - nl];
                            fi.out_map[i] = c == state_col ? 2 : (c == texts[0] ? 0 : 1);
                        }
                    }
                    fi.lit = uncast(lp->r.get())->i;
                    fi.strs = states;
                    if (!ok) fi = FusedInfo{};
                    else return;
                }
            }
        }
    }
    // ---- q8: DISTINCT (Int32, Utf8)(scan) JOIN DISTINCT (Int32)(scan)
    {
        Peeled pl_ = peel(L), pr = peel(R);
        LogicalAgg la, ra;
        if (logical_agg(pl_.n, &la) && logical_agg(pr.n, &ra) && la.fns.empty() && ra.fns.empty() && la.group.size() == 2 && ra.group.size() == 1) {
            std::vector<int> lmap, rmap;
            const Node *ls = as_scan(la.below, &lmap), *rs = as_scan(ra.below, &rmap);
            const int lkey = pl_.map[(size_t)n->on_l], rkey = pr.map[(size_t)n->on_r];
            if (ls && rs && lkey == 0 && rkey == 0 && la.below->schema[(size_t)la.group[0]].type == ColType::I32 &&
                la.below->schema[(size_t)la.group[1]].type == ColType::UTF8 && ra.below->schema[(size_t)ra.group[0]].type == ColType::I32 &&
                lmap[(size_t)la.group[0]] >= 0 && lmap[(size_t)la.group[1]] >= 0 && rmap[(size_t)ra.group[0]] >= 0) {
                bool ok = true;
                fi.out_map.assign(n->schema.size(), -1);
                for (size_t i = 0; i < n->required.size() && ok; ++i) {
                    if (!n->required[i]) continue;
                    const int c = i < nl ? pl_.map[i] : pr.map[i - nl];
                    if (c < 0) ok = false;
                    else fi.out_map[i] = i < nl ? c : 0;  // the seller column equals p_id on every output row
                }
                if (ok) {
                    fi.kind = kQ8;
                    fi.leaf_a = ls->leaf;
                    fi.leaf_b = rs->leaf;
                    fi.a_cols = {lmap[(size_t)la.group[0]], lmap[(size_t)la.group[1]]};
                    fi.b_cols = {rmap[(size_t)ra.group[0]]};
                    return;
                }
                fi = FusedInfo{};
            }
        }
    }
    // ---- q5: (COUNT GROUP BY k)(scan) JOIN (MAX over the same counts) ON count = max
    {
        Peeled pl_ = peel(L), pr = peel(R);
        LogicalAgg cnt, mx, cnt2;
        if (logical_agg(pl_.n, &cnt) && cnt.group.size() == 1 && cnt.fns == std::vector<std::string>{"count"} && logical_agg(pr.n, &mx) &&
            mx.group.empty() && mx.fns == std::vector<std::string>{"max"}) {
            Peeled below = peel(mx.below);
            std::vector<int> lmap, rmap;
            const Node *ls = as_scan(cnt.below, &lmap);
            if (ls && logical_agg(below.n, &cnt2) && cnt2.group.size() == 1 && cnt2.fns == std::vector<std::string>{"count"} &&
                below.map[(size_t)mx.args[0]] == 1 /* MAX over the count column */) {
                const Node *rs = as_scan(cnt2.below, &rmap);
                const int lkey = pl_.map[(size_t)n->on_l];
                if (rs && lkey == 1 && pr.map[(size_t)n->on_r] == 0 && cnt.below->schema[(size_t)cnt.group[0]].type == ColType::I32 &&
                    lmap[(size_t)cnt.group[0]] >= 0 && rmap[(size_t)cnt2.group[0]] >= 0 &&
                    leaf_schema(ls)[(size_t)lmap[(size_t)cnt.group[0]]].name == leaf_schema(rs)[(size_t)rmap[(size_t)cnt2.group[0]]].name) {
                    bool ok = true;
                    fi.out_map.assign(n->schema.size(), -1);
                    for (size_t i = 0; i < n->required.size() && ok; ++i) {
                        if (!n->required[i]) continue;
                        const int c = i < nl ? pl_.map[i] : pr.map[i - nl];
                        if (c < 0) ok = false;
                        else fi.out_map[i] = i < nl ? c : 1;  // maxn equals num on every output row
                    }
                    if (ok) {
                        fi.kind = kQ5;
                        fi.leaf_a = ls->leaf;
                        fi.leaf_b = rs->leaf;  // the SQL scans the relation twice; whichever leaf was fed holds it
                        fi.a_cols = {lmap[(size_t)cnt.group[0]]};
                        fi.b_cols = {rmap[(size_t)cnt2.group[0]]};
                        return;
                    }
                    fi = FusedInfo{};
                }
            }
        }
    }
    // ---- q7: bid JOIN (MAX(price) over bid) ON price = maxprice; q13: bid JOIN side_input ON auction = key
    {
        std::vector<int> lmap, rmap;
        const Node *ls = as_scan(L, &lmap);
        auto names_are = [&](const Node *scan, std::initializer_list<const char *> want) {
            const auto &s = leaf_schema(scan);
            if (s.size() != want.size()) return false;
            size_t i = 0;
            for (const char *w : want)
                if (s[i++].name != w) return false;
            return true;
        };
        const bool bid4 = ls && names_are(ls, {"auction", "bidder", "price", "b_date_time"}) && leaf_schema(ls)[0].type == ColType::I32 &&
                          leaf_schema(ls)[1].type == ColType::I32 && leaf_schema(ls)[2].type == ColType::I32 && leaf_schema(ls)[3].type == ColType::I64;
        bool lmap_id = bid4;
        for (size_t i = 0; lmap_id && i < lmap.size(); ++i) lmap_id = lmap[i] == (int)i;
        if (lmap_id && lmap.size() == 4) {
            Peeled pr = peel(R);
            LogicalAgg mx;
            const bool all_bid = n->required[0] && n->required[1] && n->required[2] && n->required[3];  // the entry points read all four
            if (all_bid && n->on_l == 2 && logical_agg(pr.n, &mx) && mx.group.empty() && mx.fns == std::vector<std::string>{"max"} && pr.map[(size_t)n->on_r] == 0) {
                const Node *rs = as_scan(mx.below, &rmap);
                if (rs && rmap[(size_t)mx.args[0]] >= 0 && leaf_schema(rs)[(size_t)rmap[(size_t)mx.args[0]]].name == "price" &&
                    leaf_schema(rs)[(size_t)rmap[(size_t)mx.args[0]]].type == ColType::I32) {
                    fi.kind = kQ7;
                    fi.leaf_a = ls->leaf;
                    fi.leaf_b = rs->leaf;
                    fi.out_map = {0, 1, 2, 3, 2};  // maxprice equals price on every output row
                    return;
                }
            }
            const Node *rs = as_scan(R, &rmap);
            if (all_bid && n->required.size() == 6 && n->required[5] && n->on_l == 0 && rs && names_are(rs, {"key", "value"}) && n->on_r == 0 && rmap == std::vector<int>{0, 1} &&
                leaf_schema(rs)[0].type == ColType::I32 && leaf_schema(rs)[1].type == ColType::I32) {
                fi.kind = kQ13;
                fi.leaf_a = ls->leaf;
                fi.leaf_b = rs->leaf;
                fi.out_map = {0, 1, 2, 3, 0, 4};  // key equals auction
                return;
            }
        }
    }
}

const char *fused_name(Fused f) {
    switch (f) {
        case kQ2: return "fused q2 filter (q1q2.hip)";
        case kQ3: return "fused q3 filter + hash join (q3.hip)";
        case kQ5: return "fused q5 count / max / select (q5.hip)";
        case kQ7: return "fused q7 max + select (q7.hip)";
        case kQ8: return "fused q8 distinct + join (q8.hip)";
        case kQ13: return "fused q13 side-input join (q13.hip)";
        case kPartialCount: return "fused Partial COUNT (q5.hip)";
        case kQ9: return "fused q9 winning bids (q4q9.hip; generic when the auction ids of a batch are not dense)";
        case kQ4: return "fused q4 average winning bid by category (q4q9.hip; generic when the auction ids of a batch are not dense)";
        case kYsb: return "fused YSB filter + join + count (ysb.hip)";
        default: return "generic (relops.hip)";
    }
}
void describe(const flockgpu_plan *pl, const Node *n, int depth, std::ostringstream &os) {
    static const char *kinds[] = {"Scan", "Filter", "Project", "Aggregate", "Join", "Repartition", "Sort", "Limit", "Window"};
    os << std::string((size_t)depth * 2, ' ') << kinds[(int)n->kind];
    if (n->kind == NKind::Aggregate) os << "(" << n->mode << ")";
    if (n->kind == NKind::Repartition) os << (n->hash_diff ? "(HashDiff, " : "(Hash, ") << n->n_parts << ")";
    if (n->kind == NKind::Scan) os << "(" << pl->ir.leaves[(size_t)n->leaf].relation << ")";
    if (n->kind == NKind::Limit) os << "(" << n->limit << ")";
    if (n->kind == NKind::Window)
        for (size_t w = 0; w < n->win_part.size(); ++w) {
            os << (w ? ", " : "(") << "ROW_NUMBER PARTITION BY";
            for (int c : n->win_part[w]) os << " " << n->in[0]->schema[(size_t)c].name;
            if (w + 1 == n->win_part.size()) os << ")";
        }
    if (n->kind == NKind::Sort) {
        os << "(";
        for (size_t i = 0; i < n->sort_cols.size(); ++i)
            os << (i ? ", " : "") << n->schema[(size_t)n->sort_cols[i].col].name << (n->sort_cols[i].descending ? " DESC" : " ASC");
        os << ")";
    }
    os << " [";
    for (size_t i = 0; i < n->schema.size(); ++i) os << (i ? ", " : "") << n->schema[i].name << ":" << type_name(n->schema[i]);
    os << "]";
    const Fused f = pl->fused[(size_t)n->id].kind;
    if (n->kind != NKind::Scan && n->kind != NKind::Project && n->kind != NKind::Repartition) os << "  <- " << fused_name(f);
    os << "\n";
    if (f != kNone) return;  // the fused pipeline swallows the sub-tree
    for (auto &c : n->in) describe(pl, c.get(), depth + 1, os);
}

// The whole plan is one NEXMark pipeline when, below pure projections, its root is a fused node (or q1's projection).
int classify(const flockgpu_plan *pl) {
    const Node *r = pl->ir.root.get();
    if (r->kind == NKind::Project) {
        int computed = 0;
        for (auto &e : r->proj) computed += e.first->kind != EKind::Col;
        if (computed == 1 && peel(r->in[0].get()).n->kind == NKind::Scan) return 1;
    }
    const Node *n = peel(r).n;
    switch (pl->fused[(size_t)n->id].kind) {
        case kQ2: return 2;
        case kQ3: return 3;
        case kQ5: return 5;
        case kQ7: return 7;
        case kQ8: return 8;
        case kQ13: return 13;
        case kQ9: return 9;
        case kQ4: return 4;
        case kYsb: return 100;  // not a NEXMark number: the Yahoo Streaming Benchmark's one query
        default: return 0;
    }
}

// ---- structural signatures (flockgpu_plan::twin_sig): everything about a sub-tree that decides the table it produces, leaves by their
// scanned columns (which leaf DATA a scan reads is decided per execute: Exec::resolve_leaf)
void expr_sig(const Expr *e, std::string *o) {
    if (!e) { *o += "~"; return; }
    uint64_t fbits = 0;   // (the literal's bits: std::to_string keeps six decimals, and two literals that differ beyond them are two tables)
    std::memcpy(&fbits, &e->f, sizeof fbits);
    *o += "(" + std::to_string((int)e->kind) + "," + std::to_string(e->col) + "," + std::to_string(e->i) + "," + std::to_string(fbits) + "," + e->s + "," +
          std::to_string((int)e->cast_to) + (e->negated ? "n" : "") + (e->big_unsigned ? "u" : "") + (e->try_cast ? "t" : "") + (e->cast_ts ? "s" : "") + "," + e->lit_kind;
    expr_sig(e->l.get(), o);
    expr_sig(e->r.get(), o);
    for (auto &x : e->list) expr_sig(x.get(), o);
    *o += ")";
}
void node_sig(const flockgpu_plan *pl, const Node *n, bool top, std::string *o, std::vector<int> *leaves) {
    *o += "[" + std::to_string((int)n->kind) + ":";
    for (size_t i = 0; i < n->schema.size(); ++i)
        *o += n->schema[i].name + "/" + std::to_string((int)n->schema[i].type) + (n->schema[i].is_ts ? "t" : "") + (n->schema[i].nullable ? "?" : "") +
              (!top && i < n->required.size() && n->required[i] ? "!" : "") + ",";   // (an Aggregate computes every output column: its own `required` does not matter)
    *o += "|" + n->mode + "|";
    for (int c : n->group) *o += std::to_string(c) + ",";
    for (auto &a : n->aggs) *o += a.fn + "." + std::to_string(a.arg) + "." + std::to_string(a.arg2) + "." + std::to_string((int)a.type) + ",";
    *o += "|" + std::to_string(n->on_l) + "," + std::to_string(n->on_r) + "," + std::to_string(n->on_l2) + "," + std::to_string(n->on_r2) + (n->join_partitioned ? "p" : "") + "|";
    for (int c : n->hash_cols) *o += std::to_string(c) + ",";
    *o += std::to_string(n->n_parts) + (n->hash_diff ? "d" : "") + "|";
    for (auto &k : n->sort_cols) *o += std::to_string(k.col) + (k.descending ? "d" : "a") + (k.nulls_first ? "f" : "l") + ",";
    *o += std::to_string(n->limit) + "|";
    for (auto &part : n->win_part) {   // (Window: the PARTITION BY columns of each ROW_NUMBER())
        for (int c : part) *o += std::to_string(c) + ",";
        *o += ";";
    }
    *o += "|";
    expr_sig(n->pred.get(), o);
    for (auto &pr : n->proj) { expr_sig(pr.first.get(), o); *o += pr.second + ","; }
    if (n->kind == NKind::Scan) {
        leaves->push_back(n->leaf);
        *o += pl->ir.leaves[(size_t)n->leaf].relation;
    }
    for (auto &c : n->in) node_sig(pl, c.get(), false, o, leaves);
    *o += "]";
}
void find_twins(flockgpu_plan *pl) {
    pl->twin_sig.assign((size_t)pl->ir.n_nodes, std::string());
    pl->twin_leaves.assign((size_t)pl->ir.n_nodes, std::vector<int>());
    std::map<std::string, std::vector<const Node *>> by_sig;
    std::vector<std::vector<int>> leaves((size_t)pl->ir.n_nodes);
    std::vector<const Node *> walk{pl->ir.root.get()};
    while (!walk.empty()) {
        const Node *n = walk.back();
        walk.pop_back();
        for (auto &c : n->in) walk.push_back(c.get());
        if (n->kind != NKind::Aggregate) continue;
        std::string sig;
        node_sig(pl, n, true, &sig, &leaves[(size_t)n->id]);
        by_sig[sig].push_back(n);
    }
    for (auto &kv : by_sig)
        if (kv.second.size() > 1)
            for (const Node *n : kv.second) {
                pl->twin_sig[(size_t)n->id] = kv.first;
                pl->twin_leaves[(size_t)n->id] = leaves[(size_t)n->id];
            }
}

int parse_and_build(flockgpu_ctx *ctx, const char *plan_json, size_t len, flockgpu_plan *pl, uint32_t flags = 0) {
    JParser jp{plan_json, plan_json + len, {}};
    JPtr root;
    if (!jp.parse(root) || root->kind != JValue::Obj)
        return fail(ctx, FLOCKGPU_ERR_PLAN, "plan: JSON error: %s", jp.err.empty() ? "not an object" : jp.err.c_str());
    if (!build_plan(root.get(), &pl->ir)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan: %s", pl->ir.why.c_str());
    pl->generic_only = (flags & FLOCKGPU_PLAN_GENERIC_ONLY) != 0;
    pl->fused.assign((size_t)pl->ir.n_nodes, FusedInfo{});
    recognise_fused(pl, pl->ir.root.get());
    pl->query = classify(pl);
    find_twins(pl);
    std::ostringstream os;
    describe(pl, pl->ir.root.get(), 0, os);
    pl->description = os.str();
    pl->leaves.resize(pl->ir.leaves.size());
    for (size_t i = 0; i < pl->leaves.size(); ++i) pl->leaves[i].cols.resize(pl->ir.leaves[i].schema.size());
    return FLOCKGPU_OK;
}

// ------------------------------------------------------------------ feeding
int find_child(const ArrowSchema *schema, const std::string &name) {
    for (int64_t i = 0; i < schema->n_children; ++i)
        if (schema->children[i] && schema->children[i]->name && name == schema->children[i]->name) return (int)i;
    return -1;
}
bool format_ok(const Field &f, const char *got) {
    if (!got) return false;
    if (f.is_ts) return !strncmp(got, "tsm:", 4) || !strcmp(got, "l");
    switch (f.type) {
        case ColType::I32: return !strcmp(got, "i");
        case ColType::I64: return !strcmp(got, "l") || !strncmp(got, "tsm:", 4);
        case ColType::U64: return !strcmp(got, "L");
        case ColType::F64: return !strcmp(got, "g");
        default: return !strcmp(got, "u");
    }
}
const char *format_of(const DevColumn &c) {
    if (c.is_ts) return "tsm:";
    switch (c.type) {
        case ColType::I32: return "i";
        case ColType::I64: return "l";
        case ColType::U64: return "L";
        case ColType::F64: return "g";
        default: return "u";
    }
}

std::string leaf_key(const flockgpu_plan *pl, int leaf, int col, const char *what) {
    char buf[96];
    snprintf(buf, sizeof buf, "plan%p.l%d.c%d.%s", (const void *)pl, leaf, col, what);
    return buf;
}
std::string node_key(const flockgpu_plan *pl, const Node *n, const char *what, int i = 0) {
    char buf[96];
    snprintf(buf, sizeof buf, "plan%p.n%d.%s%d", (const void *)pl, n->id, what, i);
    return buf;
}

// Device buffer that keeps its contents when it grows (append-only feeding).
int grow(flockgpu_ctx *ctx, const std::string &key, size_t keep_bytes, size_t want_bytes, void **ptr) {
    DeviceBuf &b = ctx->arena[key];
    if (b.cap >= want_bytes && b.ptr) { *ptr = b.ptr; return FLOCKGPU_OK; }
    size_t cap = std::max<size_t>(want_bytes + want_bytes / 2, 1024);
    cap = (cap + 255) & ~size_t(255);
    if (guard_arena()) cap = want_bytes;   // (experimental builds: exactly what was asked for, at the end of mapped memory)
    void *np = nullptr;
    hipError_t e = dev_alloc(ctx, &np, cap);
    if (e != hipSuccess) return fail(ctx, FLOCKGPU_ERR_OOM, "plan feed: hipMalloc(%zu): %s", cap, hipGetErrorString(e));
    if (b.ptr) {
        if (keep_bytes) {
            e = hipMemcpyAsync(np, b.ptr, keep_bytes, hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) { dev_free(ctx, np); return fail(ctx, FLOCKGPU_ERR_HIP, "plan feed: grow copy: %s", hipGetErrorString(e)); }
        }
        (void)hipStreamSynchronize(ctx->stream);
        dev_free(ctx, b.ptr);
    }
    b.ptr = np;
    b.cap = cap;
    *ptr = np;
    return FLOCKGPU_OK;
}

bool host_is_pinned(const void *p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory: the query fails and leaves a sticky error behind
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Host -> device on the plan's stream without a host wait.  A transfer that continues the previous one of its column (source and
// destination both contiguous: the batches of a relation are often slices of one allocation -- the reference's own
// `event_bytes_to_batch` output is -- and 52 copies of 0.7 MB cost 15 us of set-up each) is merged into it; the runs go out in
// flush_jobs(): pinned (registered) memory straight to the DMA engine, pageable memory through the plan's pinned ring.
int h2d(flockgpu_plan *pl, void *dst, const void *src, size_t bytes) {
    if (!bytes) return FLOCKGPU_OK;
    pl->fed_bytes += (int64_t)bytes;
    for (auto &r : pl->runs)
        if (static_cast<uint8_t *>(r.dst) + r.bytes == dst && static_cast<const uint8_t *>(r.src) + r.bytes == src) {
            r.bytes += bytes;
            return FLOCKGPU_OK;
        }
    pl->runs.push_back(flockgpu_plan::CopyJob{dst, src, bytes});
    return FLOCKGPU_OK;
}
int issue_runs(flockgpu_plan *pl) {
    flockgpu_ctx *ctx = pl->ctx;
    for (auto &r : pl->runs) {
        // (both ends: a merged run may have grown out of a registered range into pageable memory)
        if (host_is_pinned(r.src) && host_is_pinned(static_cast<const uint8_t *>(r.src) + r.bytes - 1)) {
            FG_HIP(ctx, hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyHostToDevice, ctx->stream));
        } else {
            pl->jobs.push_back(r);
        }
    }
    pl->runs.clear();
    return FLOCKGPU_OK;
}

// One staging lane: its share of the pieces, each copied into one of the lane's two pinned chunks (the host fills one while the
// other is in flight) and sent with hipMemcpyAsync on the plan's stream.  (hipMemcpyAsync from pageable memory would block the
// caller for the whole transfer and moves ~10 GB/s; one host thread's memcpy is no faster -- several lanes side by side are.)
int stage_lane(flockgpu_plan *pl, int lane, const std::vector<flockgpu_plan::CopyJob> &pieces, int n_lanes) {
    flockgpu_ctx *ctx = pl->ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) return FLOCKGPU_ERR_HIP;
    for (size_t i = (size_t)lane; i < pieces.size(); i += (size_t)n_lanes) {
        const int k = lane * 2 + pl->stage_next[lane];
        pl->stage_next[lane] ^= 1;
        if (!pl->stage[k]) {   // a lane's two chunks exist from the lane's first use on (all 16 up front pinned 64 MiB per plan)
            if (hipHostMalloc(&pl->stage[k], kStageChunk, hipHostMallocDefault) != hipSuccess) return FLOCKGPU_ERR_OOM;
            if (hipEventCreateWithFlags(&pl->stage_done[k], hipEventDisableTiming) != hipSuccess) return FLOCKGPU_ERR_HIP;
        } else if (hipEventSynchronize(pl->stage_done[k]) != hipSuccess) {
            return FLOCKGPU_ERR_HIP;
        }
        std::memcpy(pl->stage[k], pieces[i].src, pieces[i].bytes);
        if (hipMemcpyAsync(pieces[i].dst, pl->stage[k], pieces[i].bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return FLOCKGPU_ERR_HIP;
        if (hipEventRecord(pl->stage_done[k], ctx->stream) != hipSuccess) return FLOCKGPU_ERR_HIP;
    }
    return FLOCKGPU_OK;
}

int flush_jobs(flockgpu_plan *pl) {
    flockgpu_ctx *ctx = pl->ctx;
    FG_TRY(issue_runs(pl));
    if (pl->jobs.empty()) return FLOCKGPU_OK;
    std::vector<flockgpu_plan::CopyJob> pieces;
    size_t total = 0;
    for (auto &j : pl->jobs)
        for (size_t done = 0; done < j.bytes; done += kStageChunk) {
            const size_t n = std::min(kStageChunk, j.bytes - done);
            pieces.push_back(flockgpu_plan::CopyJob{static_cast<uint8_t *>(j.dst) + done, static_cast<const uint8_t *>(j.src) + done, n});
            total += n;
        }
    pl->jobs.clear();
    // small feeds stay on the calling thread; from a few MB on the lanes pay for their start-up
    // (2 / 4 / 6 / 8 lanes: 1.12 / 1.09-1.13 / 1.09 / 1.12 ms per 36.8 MB window, round 4 -- the transfer, not the host copy, is what is left)
    static const int want_lanes = exp_env("FLOCKGPU_STAGE_LANES") ? std::max(1, std::min(kStageLanes, atoi(exp_env("FLOCKGPU_STAGE_LANES")))) : 4;
    const int n_lanes = total < (size_t(2) << 20) ? 1 : (int)std::min<size_t>((size_t)want_lanes, std::max<size_t>(1, std::thread::hardware_concurrency()));
    if (n_lanes == 1) {
        if (stage_lane(pl, 0, pieces, 1) != FLOCKGPU_OK) return fail(ctx, FLOCKGPU_ERR_HIP, "plan feed: staged host-to-device copy failed");
        return FLOCKGPU_OK;
    }
    std::vector<std::thread> workers;
    std::vector<int> rc((size_t)n_lanes, FLOCKGPU_OK);
    for (int l = 1; l < n_lanes; ++l) workers.emplace_back([&, l] { rc[(size_t)l] = stage_lane(pl, l, pieces, n_lanes); });
    rc[0] = stage_lane(pl, 0, pieces, n_lanes);
    for (auto &w : workers) w.join();
    for (int r : rc)
        if (r != FLOCKGPU_OK) return fail(ctx, FLOCKGPU_ERR_HIP, "plan feed: staged host-to-device copy failed");
    return FLOCKGPU_OK;
}

bool validity_has_nulls(const ArrowArray *a, int64_t offset, int64_t n) {
    if (a->null_count == 0 || a->n_buffers < 1 || !a->buffers[0]) return false;
    if (a->null_count > 0) return true;
    const uint8_t *bits = static_cast<const uint8_t *>(a->buffers[0]);  // null_count == -1: not computed, look at the bitmap
    for (int64_t i = offset; i < offset + n; ++i)
        if (!((bits[i >> 3] >> (i & 7)) & 1)) return true;
    return false;
}

}  // namespace
extern "C" {
static int ring_q5_partial(flockgpu_plan *plan);   // (pane ring, below)
static void prefetch_drop(flockgpu_plan *plan);
}
namespace {

// ------------------------------------------------------------------ execution
struct Exec {
    flockgpu_plan *pl;
    flockgpu_ctx *ctx;
    std::map<std::string, Table> memo;   // this execute's tables of the sub-trees that have twins (flockgpu_plan::twin_sig)

    // A relation the SQL scans twice (q5 and q7 read `bid` in both join inputs, q5_plan.fmt:6,13) has two MemoryExec leaves
    // but arrives as ONE source, which feed_data_sources hands to the first matching leaf (context.rs:273-300) -- the
    // second would stay empty.  An unfed leaf therefore reads the fed leaf that scans the same columns.
    int resolve_leaf(int leaf) const {
        if (pl->leaves[(size_t)leaf].rows > 0) return leaf;
        const Leaf &me = pl->ir.leaves[(size_t)leaf];
        for (size_t o = 0; o < pl->leaves.size(); ++o) {
            if ((int)o == leaf || pl->leaves[o].rows == 0) continue;
            const Leaf &other = pl->ir.leaves[o];
            bool ok = true;
            for (size_t c = 0; c < me.schema.size() && ok; ++c) {
                if (!me.needed[c]) continue;
                ok = false;
                for (size_t d = 0; d < other.schema.size(); ++d)
                    if (other.schema[d].name == me.schema[c].name && other.schema[d].type == me.schema[c].type && other.needed[d]) ok = true;
            }
            if (ok) return (int)o;
        }
        return leaf;
    }
    int leaf_column(int leaf, int src_leaf, int col) const {  // column `col` of `leaf`, as an index into `src_leaf`
        if (leaf == src_leaf) return col;
        const Leaf &me = pl->ir.leaves[(size_t)leaf], &other = pl->ir.leaves[(size_t)src_leaf];
        for (size_t d = 0; d < other.schema.size(); ++d)
            if (other.schema[d].name == me.schema[(size_t)col].name) return (int)d;
        return -1;
    }

    int scan_table(const Node *n, Table *t) {
        const int src = resolve_leaf(n->leaf);
        const LeafData &ld = pl->leaves[(size_t)src];
        const Leaf &lf = pl->ir.leaves[(size_t)n->leaf];
        t->rows = ld.rows;
        t->cols.resize(n->schema.size());
        for (size_t i = 0; i < n->schema.size(); ++i) {
            TCol &tc = t->cols[i];
            tc.c.type = lf.schema[i].type;
            tc.c.is_ts = lf.schema[i].is_ts;
            tc.c.nullable = lf.schema[i].nullable;
            if (!lf.needed[i]) continue;
            tc.present = true;
            const DevBuf &b = ld.cols[(size_t)leaf_column(n->leaf, src, (int)i)];
            if (tc.c.type == ColType::UTF8 && !b.offsets) {  // never fed: an empty relation still has a one-entry offsets array
                void *p = nullptr;
                FG_TRY(arena_get(ctx, leaf_key(pl, n->leaf, (int)i, "empty").c_str(), 64, &p));
                FG_HIP(ctx, hipMemsetAsync(p, 0, 64, ctx->stream));
                tc.c.offsets = static_cast<int32_t *>(p);
                tc.c.values = static_cast<uint8_t *>(p) + 16;
            } else if (!b.values) {
                void *p = nullptr;
                FG_TRY(arena_get(ctx, leaf_key(pl, n->leaf, (int)i, "empty").c_str(), 64, &p));
                tc.c.values = p;
            } else {
                tc.c.values = b.values;
                tc.c.offsets = b.offsets;
                tc.c.bytes = b.bytes;
                if (b.valid && b.valid_rows > 0) {   // batches fed after the last one with NULLs: all valid
                    DevBuf &wb = pl->leaves[(size_t)src].cols[(size_t)leaf_column(n->leaf, src, (int)i)];
                    if (wb.valid_rows < ld.rows) {
                        FG_HIP(ctx, hipMemsetAsync(wb.valid + wb.valid_rows, 1, (size_t)(ld.rows - wb.valid_rows), ctx->stream));
                        wb.valid_rows = ld.rows;
                    }
                    tc.c.valid = b.valid;
                }
            }
        }
        return FLOCKGPU_OK;
    }
    // a fused pipeline reads plain NEXMark columns: a leaf column that carries validity sends its sub-tree to the generic operators
    bool leaf_has_validity(int leaf) const {
        if (leaf < 0) return false;
        for (auto &c : pl->leaves[(size_t)resolve_leaf(leaf)].cols)
            if (c.valid && c.valid_rows > 0) return true;
        return false;
    }

    // leaf column as a typed device pointer (fused entry points)
    template <typename T>
    const T *leaf_col(int leaf, int col) {
        if (col < 0) return nullptr;
        return static_cast<const T *>(pl->leaves[(size_t)leaf].cols[(size_t)col].values);
    }
    int leaf_utf8(int leaf, int col, flockgpu_utf8 *out) {
        const DevBuf &b = pl->leaves[(size_t)leaf].cols[(size_t)col];
        if (b.offsets) {
            *out = flockgpu_utf8{b.offsets, static_cast<const uint8_t *>(b.values)};
            return FLOCKGPU_OK;
        }
        void *p = nullptr;
        FG_TRY(arena_get(ctx, leaf_key(pl, leaf, col, "empty").c_str(), 64, &p));
        FG_HIP(ctx, hipMemsetAsync(p, 0, 64, ctx->stream));
        *out = flockgpu_utf8{static_cast<int32_t *>(p), static_cast<uint8_t *>(p) + 16};
        return FLOCKGPU_OK;
    }
    static flockgpu_windows whole(int64_t rows, int64_t (&off)[2], int32_t (&lo)[1], int32_t (&hi)[1]) {
        off[0] = 0; off[1] = rows; lo[0] = 0; hi[0] = 1;
        return flockgpu_windows{off, 1, lo, hi, 1};
    }
    static TCol dev_col(ColType t, const void *values, const int32_t *offsets = nullptr, int64_t bytes = 0, bool ts = false) {
        TCol c;
        c.present = true;
        c.c.type = t;
        c.c.is_ts = ts;
        c.c.values = values;
        c.c.offsets = offsets;
        c.c.bytes = bytes;
        return c;
    }
    // slots produced by a fused entry point -> the node's output columns
    void place(const Node *n, const FusedInfo &fi, const std::vector<TCol> &slots, int64_t rows, Table *t) {
        t->rows = rows;
        t->cols.assign(n->schema.size(), TCol{});
        for (size_t i = 0; i < n->schema.size(); ++i) {
            t->cols[i].c.type = n->schema[i].type;
            t->cols[i].c.is_ts = n->schema[i].is_ts;
            if (fi.out_map[i] < 0 || !n->required[i]) continue;
            t->cols[i] = slots[(size_t)fi.out_map[i]];
            t->cols[i].c.is_ts = n->schema[i].is_ts;
            t->cols[i].c.nullable = n->schema[i].nullable;
        }
    }

    int run_fused(const Node *n, const FusedInfo &fi, Table *t) {
        int64_t off_a[2], off_b[2];
        int32_t lo_a[1], hi_a[1], lo_b[1], hi_b[1];
        const LeafData &A = pl->leaves[(size_t)fi.leaf_a];
        switch (fi.kind) {
            case kQ2: {
                flockgpu_bid_cols bc{leaf_col<int32_t>(fi.leaf_a, fi.a_cols[0]), nullptr, leaf_col<int32_t>(fi.leaf_a, fi.a_cols[1]), nullptr, A.rows};
                flockgpu_windows w = whole(A.rows, off_a, lo_a, hi_a);
                flockgpu_q2_result r{};
                FG_TRY(flockgpu_q2_filter(ctx, &bc, &w, fi.lit, &r));
                place(n, fi, {dev_col(ColType::I32, r.auction), dev_col(ColType::I32, r.price)}, r.rows, t);
                return FLOCKGPU_OK;
            }
            case kPartialCount: {
                flockgpu_bid_cols bc{leaf_col<int32_t>(fi.leaf_a, fi.a_cols[0]), nullptr, nullptr, nullptr, A.rows};
                flockgpu_windows w = whole(A.rows, off_a, lo_a, hi_a);
                flockgpu_q5_partial_result r{};
                FG_TRY(flockgpu_q5_partial_counts(ctx, &bc, &w, &r));
                // node-owned copies: a second Partial COUNT of the same plan reuses the entry point's ctx-level buffers
                uint64_t *wide = nullptr;
                int32_t *keys = nullptr;
                FG_TRY(arena_get_t(ctx, node_key(pl, n, "cnt").c_str(), (size_t)r.rows + 2, &wide));
                FG_TRY(arena_get_t(ctx, node_key(pl, n, "key").c_str(), (size_t)r.rows + 4, &keys));
                FG_TRY(widen_u32_to_u64(ctx, r.count, r.rows, wide));
                if (r.rows) FG_HIP(ctx, hipMemcpyAsync(keys, r.auction, sizeof(int32_t) * (size_t)r.rows, hipMemcpyDeviceToDevice, ctx->stream));
                place(n, fi, {dev_col(ColType::I32, keys), dev_col(ColType::U64, wide)}, r.rows, t);
                return FLOCKGPU_OK;
            }
            case kQ3: {
                const LeafData &B = pl->leaves[(size_t)fi.leaf_b];
                flockgpu_auction_cols ac{leaf_col<int32_t>(fi.leaf_a, fi.a_cols[0]), leaf_col<int32_t>(fi.leaf_a, fi.a_cols[1]),
                                         leaf_col<int32_t>(fi.leaf_a, fi.a_cols[2]), A.rows};
                flockgpu_person_cols pc{};
                pc.p_id = leaf_col<int32_t>(fi.leaf_b, fi.b_cols[0]);
                FG_TRY(leaf_utf8(fi.leaf_b, fi.b_cols[1], &pc.name));
                FG_TRY(leaf_utf8(fi.leaf_b, fi.b_cols[2], &pc.city));
                FG_TRY(leaf_utf8(fi.leaf_b, fi.b_cols[3], &pc.state));
                pc.rows = B.rows;
                flockgpu_windows aw = whole(A.rows, off_a, lo_a, hi_a), pw = whole(B.rows, off_b, lo_b, hi_b);
                std::vector<const char *> lits;
                for (auto &s : fi.strs) lits.push_back(s.c_str());
                flockgpu_q3_result r{};
                FG_TRY(flockgpu_q3_join(ctx, &ac, &aw, &pc, &pw, fi.lit, lits.data(), (int)lits.size(), &r));
                place(n, fi,
                      {dev_col(ColType::UTF8, r.name.data, r.name.offsets, r.name_bytes), dev_col(ColType::UTF8, r.city.data, r.city.offsets, r.city_bytes),
                       dev_col(ColType::UTF8, r.state.data, r.state.offsets, r.state_bytes), dev_col(ColType::I32, r.a_id)},
                      r.rows, t);
                return FLOCKGPU_OK;
            }
            case kQ5: {
                if (pl->ring_ppw && pl->ring_q5) {
                    // pane ring: every pane was counted once, when it was the newest (its Partial groups stay on the device); the window is
                    // the FinalPartitioned merge of the held panes' groups -> MAX -> join, i.e. q5.dag's second half over (auction, count) rows
                    FG_TRY(ring_q5_partial(pl));
                    int64_t total = 0;
                    for (int64_t g : pl->ring_groups) total += g;
                    flockgpu_windows w = whole(total, off_a, lo_a, hi_a);
                    flockgpu_q5_result r{};
                    API_CLOCK("ring.q5_weighted");
                    FG_TRY(flockgpu_q5_hot_items_weighted(ctx, pl->ring_auction, pl->ring_count, total, &w, &r));
                    place(n, fi, {dev_col(ColType::I32, r.auction), dev_col(ColType::U64, r.num)}, r.rows, t);
                    return FLOCKGPU_OK;
                }
                // both leaves scan the same relation; whichever was fed holds it (feed_data_sources gives it to the first match)
                const bool first = A.rows > 0 || pl->leaves[(size_t)fi.leaf_b].rows == 0;
                const int leaf = first ? fi.leaf_a : fi.leaf_b, col = first ? fi.a_cols[0] : fi.b_cols[0];
                const int64_t rows = pl->leaves[(size_t)leaf].rows;
                flockgpu_bid_cols bc{leaf_col<int32_t>(leaf, col), nullptr, nullptr, nullptr, rows};
                flockgpu_windows w = whole(rows, off_a, lo_a, hi_a);
                flockgpu_q5_result r{};
                API_CLOCK("q5_hot_items");
                FG_TRY(flockgpu_q5_hot_items(ctx, &bc, &w, &r));
                place(n, fi, {dev_col(ColType::I32, r.auction), dev_col(ColType::U64, r.num)}, r.rows, t);
                return FLOCKGPU_OK;
            }
            case kQ7: {
                const bool first = A.rows > 0 || pl->leaves[(size_t)fi.leaf_b].rows == 0;
                const int leaf = first ? fi.leaf_a : fi.leaf_b;
                const Leaf &lf = pl->ir.leaves[(size_t)leaf];
                auto by_name = [&](const char *nm) {
                    for (size_t i = 0; i < lf.schema.size(); ++i)
                        if (lf.schema[i].name == nm) return (int)i;
                    return -1;
                };
                const int64_t rows = pl->leaves[(size_t)leaf].rows;
                flockgpu_bid_cols bc{leaf_col<int32_t>(leaf, by_name("auction")), leaf_col<int32_t>(leaf, by_name("bidder")),
                                     leaf_col<int32_t>(leaf, by_name("price")), leaf_col<int64_t>(leaf, by_name("b_date_time")), rows};
                if (rows > 0 && (!bc.auction || !bc.bidder || !bc.price || !bc.b_date_time))
                    return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: the fed bid relation lacks a column of [auction, bidder, price, b_date_time]");
                flockgpu_windows w = whole(rows, off_a, lo_a, hi_a);
                flockgpu_q7_result r{};
                FG_TRY(flockgpu_q7_highest_bid(ctx, &bc, &w, &r));
                place(n, fi, {dev_col(ColType::I32, r.auction), dev_col(ColType::I32, r.bidder), dev_col(ColType::I32, r.price),
                              dev_col(ColType::I64, r.b_date_time, nullptr, 0, true)}, r.rows, t);
                return FLOCKGPU_OK;
            }
            case kQ13: {
                const LeafData &B = pl->leaves[(size_t)fi.leaf_b];
                flockgpu_bid_cols bc{leaf_col<int32_t>(fi.leaf_a, 0), leaf_col<int32_t>(fi.leaf_a, 1), leaf_col<int32_t>(fi.leaf_a, 2),
                                     leaf_col<int64_t>(fi.leaf_a, 3), A.rows};
                flockgpu_windows w = whole(A.rows, off_a, lo_a, hi_a);
                flockgpu_q13_result r{};
                FG_TRY(flockgpu_q13_side_join(ctx, &bc, &w, leaf_col<int32_t>(fi.leaf_b, 0), leaf_col<int32_t>(fi.leaf_b, 1), B.rows, &r));
                place(n, fi, {dev_col(ColType::I32, r.auction), dev_col(ColType::I32, r.bidder), dev_col(ColType::I32, r.price),
                              dev_col(ColType::I64, r.b_date_time, nullptr, 0, true), dev_col(ColType::I32, r.value)}, r.rows, t);
                return FLOCKGPU_OK;
            }
            case kQ8: {
                const LeafData &B = pl->leaves[(size_t)fi.leaf_b];
                flockgpu_person_cols pc{};
                pc.p_id = leaf_col<int32_t>(fi.leaf_a, fi.a_cols[0]);
                FG_TRY(leaf_utf8(fi.leaf_a, fi.a_cols[1], &pc.name));
                pc.rows = A.rows;
                flockgpu_auction_cols ac{nullptr, leaf_col<int32_t>(fi.leaf_b, fi.b_cols[0]), nullptr, B.rows};
                flockgpu_windows pw = whole(A.rows, off_a, lo_a, hi_a), aw = whole(B.rows, off_b, lo_b, hi_b);
                flockgpu_q8_result r{};
                FG_TRY(flockgpu_q8_join(ctx, &pc, &pw, &ac, &aw, &r));
                place(n, fi, {dev_col(ColType::I32, r.p_id), dev_col(ColType::UTF8, r.name.data, r.name.offsets, r.name_bytes)}, r.rows, t);
                return FLOCKGPU_OK;
            }
            case kQ9:
            case kQ4: {
                const bool q9 = fi.kind == kQ9;
                const LeafData &B = pl->leaves[(size_t)fi.leaf_b];
                if (B.rows == 0 && A.rows > 0 && q9)  // the bids went to the inner scan's leaf, which holds three of the four columns
                    return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: q9's bids were fed to the inner scan");
                flockgpu_auction_time_cols ac{leaf_col<int32_t>(fi.leaf_a, fi.a_cols[0]), leaf_col<int32_t>(fi.leaf_a, fi.a_cols[1]),
                                              leaf_col<int64_t>(fi.leaf_a, fi.a_cols[2]), leaf_col<int64_t>(fi.leaf_a, fi.a_cols[3]), A.rows};
                flockgpu_bid_cols bc{};
                if (q9) bc = flockgpu_bid_cols{leaf_col<int32_t>(fi.leaf_b, 0), leaf_col<int32_t>(fi.leaf_b, 1), leaf_col<int32_t>(fi.leaf_b, 2), leaf_col<int64_t>(fi.leaf_b, 3), B.rows};
                else bc = flockgpu_bid_cols{leaf_col<int32_t>(fi.leaf_b, fi.b_cols[0]), nullptr, leaf_col<int32_t>(fi.leaf_b, fi.b_cols[1]), leaf_col<int64_t>(fi.leaf_b, fi.b_cols[2]), B.rows};
                flockgpu_windows aw = whole(A.rows, off_a, lo_a, hi_a), bw = whole(B.rows, off_b, lo_b, hi_b);
                if (q9) {
                    flockgpu_q9_result r{};
                    FG_TRY(flockgpu_q9_winning_bids(ctx, &ac, &aw, &bc, &bw, &r));
                    place(n, fi, {dev_col(ColType::I32, r.auction), dev_col(ColType::I32, r.bidder), dev_col(ColType::I32, r.price),
                                  dev_col(ColType::I64, r.b_date_time, nullptr, 0, true)}, r.rows, t);
                } else {
                    flockgpu_q4_result r{};
                    FG_TRY(flockgpu_q4_avg_final_by_category(ctx, &ac, &aw, &bc, &bw, &r));
                    place(n, fi, {dev_col(ColType::I32, r.category), dev_col(ColType::F64, r.avg_final)}, r.rows, t);
                }
                return FLOCKGPU_OK;
            }
            case kYsb: {
                const LeafData &B = pl->leaves[(size_t)fi.leaf_b];
                flockgpu_ysb_event_cols ev{};
                FG_TRY(leaf_utf8(fi.leaf_a, fi.a_cols[0], &ev.ad_id));
                FG_TRY(leaf_utf8(fi.leaf_a, fi.a_cols[1], &ev.event_type));
                ev.rows = A.rows;
                flockgpu_ysb_campaign_cols cc{};
                FG_TRY(leaf_utf8(fi.leaf_b, fi.b_cols[0], &cc.c_ad_id));
                FG_TRY(leaf_utf8(fi.leaf_b, fi.b_cols[1], &cc.campaign_id));
                cc.rows = B.rows;
                flockgpu_windows w = whole(A.rows, off_a, lo_a, hi_a);
                flockgpu_ysb_result r{};
                FG_TRY(flockgpu_ysb_campaign_counts(ctx, &ev, &w, &cc, fi.strs[0].c_str(), &r));
                place(n, fi, {dev_col(ColType::UTF8, r.campaign_id.data, r.campaign_id.offsets, r.campaign_bytes), dev_col(ColType::U64, r.count)}, r.rows, t);
                return FLOCKGPU_OK;
            }
            default:
                return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: unknown fused pipeline");
        }
    }

    // ---- predicates
    static bool cmp_of(const std::string &op, CmpOp *out, bool flip) {
        static const struct { const char *n; CmpOp a, flipped; } t[] = {
            {"Eq", CmpOp::EQ, CmpOp::EQ}, {"NotEq", CmpOp::NE, CmpOp::NE}, {"Lt", CmpOp::LT, CmpOp::GT},
            {"LtEq", CmpOp::LE, CmpOp::GE}, {"Gt", CmpOp::GT, CmpOp::LT}, {"GtEq", CmpOp::GE, CmpOp::LE}};
        for (auto &e : t)
            if (op == e.n) { *out = flip ? e.flipped : e.a; return true; }
        return false;
    }
    // FilterExec's predicate -> the postfix program of pred.hpp (ONE kernel evaluates it per flag tile).  Leaves: a column against a
    // literal or another column (integers, Float64, Utf8 `=` / `<>`), `column % literal` against a literal, IS [NOT] NULL, a boolean
    // literal; IN lists become OR chains of `=` leaves; NOT / AND / OR combine in three-valued logic on the device.  AND / OR push
    // their deeper operand first (they commute), so the operand stack stays at log2(leaves) + 1.
    struct Operand {
        enum K { COL, INT, FLT, STR, MOD, NUL, BAD } k = BAD;
        const TCol *col = nullptr;
        int64_t i = 0;
        double f = 0;
        std::string s;
        int64_t modulus = 0;
        bool big = false;   // INT: a UInt64 literal above INT64_MAX (`i` is its bit pattern)
    };
    // `e` without the casts that change no value (and no NULL): to the operand's own type, Int32 -> Int64 / Float64, an integer literal to any
    // numeric type.  A cast that may truncate, overflow or -- TRY_CAST -- turn a value into NULL stays: the one-pass predicate program has
    // no leaf for it and the general evaluator (valprog.hpp) takes the predicate.
    const Expr *uncast_exact(const Expr *e, const Table &in) const {
        while (e && e->kind == EKind::Cast) {
            const Expr *x = e->l.get();
            const int from = val_type_of(x, in);
            const bool exact = from == (int)e->cast_to || (from == 0 && (e->cast_to == ColType::I64 || e->cast_to == ColType::F64)) || x->kind == EKind::LitI ||
                               (x->kind == EKind::LitF && e->cast_to == ColType::F64) || x->kind == EKind::LitNull;
            if (!exact) break;
            e = x;
        }
        return e;
    }
    Operand operand(const Expr *e, const Table &in) const {
        Operand o;
        bool neg = false;
        e = uncast_exact(e, in);
        while (e->kind == EKind::Neg) {   // -literal: folded here (a negated column is not taken)
            neg = !neg;
            e = uncast_exact(e->l.get(), in);
        }
        auto column = [&](const Expr *x) -> const TCol * {
            const TCol &c = in.cols[(size_t)x->col];
            return c.present || c.c.all_null ? &c : nullptr;
        };
        switch (e->kind) {
            case EKind::LitI:
                if (e->big_unsigned && neg) return o;   // -(a UInt64 beyond 2^63) is below every Int64: not a value any column type holds
                o.k = Operand::INT;
                o.i = neg ? (int64_t)(0 - (uint64_t)e->i) : e->i;
                o.big = e->big_unsigned;
                return o;
            case EKind::LitF: o.k = Operand::FLT; o.f = neg ? -e->f : e->f; return o;
            case EKind::LitS: if (!neg) { o.k = Operand::STR; o.s = e->s; } return o;
            case EKind::LitNull: o.k = Operand::NUL; return o;
            case EKind::Col:
                if (!neg && (o.col = column(e))) o.k = o.col->c.all_null ? Operand::NUL : Operand::COL;
                return o;
            case EKind::Bin:
                if (!neg && e->s == "Modulo") {
                    const Expr *c = uncast_exact(e->l.get(), in), *m = uncast_exact(e->r.get(), in);
                    if (c->kind == EKind::Col && m->kind == EKind::LitI && (o.col = column(c))) {
                        o.k = o.col->c.all_null ? Operand::NUL : Operand::MOD;
                        o.modulus = m->i;
                    }
                }
                return o;
            default: return o;
        }
    }
    static int pred_need(const Expr *e) {   // operand-stack slots the sub-tree needs when its deeper side goes first
        if (e->kind == EKind::Not) return pred_need(e->l.get());
        if (e->kind == EKind::Bin && (e->s == "And" || e->s == "Or")) {
            const int a = pred_need(e->l.get()), b = pred_need(e->r.get());
            return a == b ? a + 1 : std::max(a, b);
        }
        return e->kind == EKind::InList ? 2 : 1;
    }
    int pred_full() { return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: predicate beyond %d comparisons / %d columns / %d literal bytes", kPredMaxLeaves, kPredMaxCols, kPredLitPool); }
    int pred_const(PredBuilder &b, int v) {   // 0 FALSE, 1 TRUE, 2 NULL
        PredLeafDesc l{};
        l.kind = (uint8_t)PredLeafKind::Const;
        l.lit = v;
        return b.add_leaf(l) ? FLOCKGPU_OK : pred_full();
    }
    // `x op y` with the literal (if any) on the right
    int pred_compare(const std::string &opname, const Expr *le, const Expr *re, const Table &in, PredBuilder &b) {
        Operand l = operand(le, in), r = operand(re, in);
        bool flip = false;
        if (l.k == Operand::INT || l.k == Operand::FLT || l.k == Operand::STR) { std::swap(l, r); flip = true; }
        CmpOp op;
        if (!cmp_of(opname, &op, flip)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: operator '%s' in a predicate", opname.c_str());
        if (l.k == Operand::BAD || r.k == Operand::BAD) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: predicate shape is not supported");
        if (l.k == Operand::NUL || r.k == Operand::NUL) return pred_const(b, 2);   // a comparison with NULL is NULL
        PredLeafDesc d{};
        d.cmp = (uint8_t)op;
        auto is_int = [](const TCol *c) { return c->c.type == ColType::I32 || c->c.type == ColType::I64 || c->c.type == ColType::U64; };
        if (l.k == Operand::COL && r.k == Operand::COL) {
            const bool fa = l.col->c.type == ColType::F64, fb = r.col->c.type == ColType::F64;
            if (fa != fb || !(fa || (is_int(l.col) && is_int(r.col))) || ((l.col->c.type == ColType::U64) != (r.col->c.type == ColType::U64)))
                return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: column comparison needs two integer columns of one signedness, or two Float64 columns");
            d.kind = (uint8_t)(fa ? PredLeafKind::CmpF64Col : PredLeafKind::CmpIntCol);
            d.uns = l.col->c.type == ColType::U64;
            const int ia = b.add_col(l.col->c), ib = b.add_col(r.col->c);
            if (ia < 0 || ib < 0) return pred_full();
            d.a = (uint8_t)ia;
            d.b = (uint8_t)ib;
            return b.add_leaf(d) ? FLOCKGPU_OK : pred_full();
        }
        if (l.k != Operand::COL && l.k != Operand::MOD) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: a comparison of two literals");
        // from here on the right side is read as a literal: `a % 3 = b`, `a = b % 3`, `a % 2 = b % 2` have no one-pass leaf (the general
        // evaluator takes them) -- without this guard they would compare against r.i's default 0
        if (r.k != Operand::INT && r.k != Operand::FLT && r.k != Operand::STR)
            return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: a column (or its remainder) against a non-literal has no one-pass leaf");
        const ColType ct = l.col->c.type;
        if (r.k == Operand::STR) {
            if (ct != ColType::UTF8 || l.k == Operand::MOD || (op != CmpOp::EQ && op != CmpOp::NE))
                return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: a Utf8 literal compares with a Utf8 column by = or <>");
            d.kind = (uint8_t)PredLeafKind::Utf8Eq;
            d.negate = op == CmpOp::NE;
            d.lit_len = (int32_t)r.s.size();
            const int ia = b.add_col(l.col->c);
            if (ia < 0 || !b.add_literal(r.s, &d.lit_off)) return pred_full();
            d.a = (uint8_t)ia;
            return b.add_leaf(d) ? FLOCKGPU_OK : pred_full();
        }
        if (ct == ColType::UTF8) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: a Utf8 column against a number");
        if (ct == ColType::F64) {
            if (l.k == Operand::MOD) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: modulo of a Float64 column");
            d.kind = (uint8_t)PredLeafKind::CmpF64Lit;
            const double lit = r.k == Operand::FLT ? r.f : (double)r.i;
            std::memcpy(&d.lit, &lit, sizeof d.lit);
            const int ia = b.add_col(l.col->c);
            if (ia < 0) return pred_full();
            d.a = (uint8_t)ia;
            return b.add_leaf(d) ? FLOCKGPU_OK : pred_full();
        }
        // an integer column (or its remainder) against a number
        int64_t lit = r.i;
        if (r.k == Operand::FLT) {
            // CAST(int AS Float64) op f: exact as an integer comparison while the cast is (Int32 always; wider columns are not taken)
            if (ct != ColType::I32 || l.k == Operand::MOD || r.f != r.f) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: an Int64 column against a Float64 literal");
            const double fl = std::floor(r.f), ce = std::ceil(r.f);
            const bool whole = fl == r.f;
            if (!whole && (op == CmpOp::EQ || op == CmpOp::NE)) return pred_const_valid(b, l.col, op == CmpOp::NE);
            const double pick = (op == CmpOp::LT || op == CmpOp::GE) ? ce : fl;   // x < f <=> x < ceil f;  x <= f <=> x <= floor f;  x > f <=> x > floor f;  x >= f <=> x >= ceil f
            if (pick > 4e18) lit = INT64_MAX; else if (pick < -4e18) lit = INT64_MIN; else lit = (int64_t)pick;
        }
        if (r.k == Operand::INT && r.big && (ct != ColType::U64 || l.k == Operand::MOD))   // above every signed value (and every remainder)
            return pred_const_valid(b, l.col, op == CmpOp::NE || op == CmpOp::LT || op == CmpOp::LE);
        if (ct == ColType::U64 && !r.big && lit < 0)   // a negative literal is below every UInt64
            return pred_const_valid(b, l.col, op == CmpOp::NE || op == CmpOp::GT || op == CmpOp::GE);
        d.kind = (uint8_t)PredLeafKind::CmpIntLit;
        d.uns = ct == ColType::U64;
        if (l.k == Operand::MOD) {
            if (ct == ColType::U64) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: modulo needs a signed integer column");
            if (l.modulus == 0) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: modulo by zero (the reference raises a DataFusion error)");
            // x % -1 == 0 for every x (and INT64_MIN % -1 traps in hardware): the remainder by |m| has the same value
            const int64_t m = l.modulus == INT64_MIN ? l.modulus : (l.modulus < 0 ? -l.modulus : l.modulus);
            d.modulus = m;
            if (ct == ColType::I32 && m > 0 && m <= 0x7fffffffll) {
                d.mod_kind = 1;
                d.mod = umod32_make((uint32_t)m);
            } else {
                d.mod_kind = 2;
            }
        }
        if (ct == ColType::I32 && d.mod_kind != 2 && (lit > INT32_MAX || lit < INT32_MIN)) {
            // no Int32 (and no remainder by |m| < 2^31) reaches the literal: the comparison is the same for every non-NULL row
            const bool above = lit > INT32_MAX;
            const bool v = op == CmpOp::NE ? true : op == CmpOp::EQ ? false : (op == CmpOp::LT || op == CmpOp::LE) ? above : !above;
            return pred_const_valid(b, l.col, v);
        }
        d.lit = lit;
        const int ia = b.add_col(l.col->c);
        if (ia < 0) return pred_full();
        d.a = (uint8_t)ia;
        return b.add_leaf(d) ? FLOCKGPU_OK : pred_full();
    }
    // a comparison whose outcome is `v` for every row whose column is not NULL (NULL where it is): `col IS NOT NULL` when v, else NOT of it AND NULL...
    // expressed with the leaves at hand: v ? (col IS NOT NULL OR NULL) : (col IS NULL AND NULL)
    int pred_const_valid(PredBuilder &b, const TCol *col, bool v) {
        if (!col->c.valid) return pred_const(b, v ? 1 : 0);
        PredLeafDesc d{};
        d.kind = (uint8_t)PredLeafKind::IsNull;
        d.negate = v ? 1 : 0;
        const int ia = b.add_col(col->c);
        if (ia < 0) return pred_full();
        d.a = (uint8_t)ia;
        if (!b.add_leaf(d)) return pred_full();
        FG_TRY(pred_const(b, 2));
        return b.push(v ? PredOpKind::Or : PredOpKind::And) ? FLOCKGPU_OK : pred_full();
    }
    int compile_pred(const Expr *e, const Table &in, PredBuilder &b) {
        switch (e->kind) {
            case EKind::LitB: return pred_const(b, e->i ? 1 : 0);
            case EKind::LitNull: return pred_const(b, 2);
            case EKind::Not:
                FG_TRY(compile_pred(e->l.get(), in, b));
                return b.push(PredOpKind::Not) ? FLOCKGPU_OK : pred_full();
            case EKind::IsNull:
            case EKind::IsNotNull: {
                const Expr *a = uncast_exact(e->l.get(), in);
                if (a->kind != EKind::Col) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: IS [NOT] NULL of something other than a column");
                const TCol &c = in.cols[(size_t)a->col];
                if (c.c.all_null) return pred_const(b, e->kind == EKind::IsNull ? 1 : 0);
                if (!c.present) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: predicate column was not materialised");
                if (!c.c.valid) return pred_const(b, e->kind == EKind::IsNull ? 0 : 1);
                PredLeafDesc d{};
                d.kind = (uint8_t)PredLeafKind::IsNull;
                d.negate = e->kind == EKind::IsNotNull;
                const int ia = b.add_col(c.c);
                if (ia < 0) return pred_full();
                d.a = (uint8_t)ia;
                return b.add_leaf(d) ? FLOCKGPU_OK : pred_full();
            }
            case EKind::InList: {   // x IN (a, b, ...) = (x = a) OR (x = b) OR ...; NOT IN = NOT of that (NULL when x is)
                for (size_t i = 0; i < e->list.size(); ++i) {
                    FG_TRY(pred_compare("Eq", e->l.get(), e->list[i].get(), in, b));
                    if (i > 0 && !b.push(PredOpKind::Or)) return pred_full();
                }
                if (e->negated && !b.push(PredOpKind::Not)) return pred_full();
                return FLOCKGPU_OK;
            }
            case EKind::Bin: {
                if (e->s == "And" || e->s == "Or") {
                    const Expr *first = e->l.get(), *second = e->r.get();
                    if (pred_need(second) > pred_need(first)) std::swap(first, second);
                    FG_TRY(compile_pred(first, in, b));
                    FG_TRY(compile_pred(second, in, b));
                    return b.push(e->s == "And" ? PredOpKind::And : PredOpKind::Or) ? FLOCKGPU_OK : pred_full();
                }
                return pred_compare(e->s, e->l.get(), e->r.get(), in, b);
            }
            default:
                return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: predicate is not a comparison");
        }
    }

    // The Utf8 columns of one take share their row list: up to four of them go through ONE length pass, ONE scan, ONE host wait and
    // ONE emit launch (gather_utf8_multi_*) instead of a take -- and a wait -- per column.  A stage plan's operators materialise small
    // tables many times over (filter, join output, one take per repartition), and at that size the waits ARE the cost.
    // random_rows: the row list is in no order and names most rows (ORDER BY): fixed-width columns are taken as 16-byte records (gather_fixed_packed)
    int take_table(const Node *n, const Table &in, const std::vector<char> &required, const int32_t *rows, int64_t n_rows, int first_out, Table *out, bool random_rows = false) {
        std::vector<size_t> utf8;
        GatherCols batch;
        for (size_t i = 0; i < in.cols.size(); ++i) {
            TCol &o = out->cols[(size_t)first_out + i];
            o.c.type = in.cols[i].c.type;
            o.c.is_ts = in.cols[i].c.is_ts;
            o.c.nullable = in.cols[i].c.nullable;
            if (!required[(size_t)first_out + i] || !in.cols[i].present) continue;
            if (in.cols[i].c.type == ColType::UTF8) {
                utf8.push_back(i);
                continue;
            }
            o.present = true;
            o.subset_of = in.cols[i].subset_of ? in.cols[i].subset_of : in.cols[i].c.values;
            if (!in.cols[i].c.valid && n_rows > 0) {   // a fixed-width column without NULLs: one launch takes up to eight of them (gather_fixed_multi)
                void *pv = nullptr;
                FG_TRY(arena_get(ctx, (node_key(pl, n, "take", first_out + (int)i) + ".val").c_str(), (size_t)n_rows * col_width(in.cols[i].c.type) + 16, &pv));
                o.c = in.cols[i].c;
                o.c.values = pv;
                o.c.offsets = nullptr;
                o.c.bytes = 0;
                batch.src[batch.n] = in.cols[i].c.values;
                batch.out[batch.n] = pv;
                batch.width[batch.n] = (int32_t)col_width(in.cols[i].c.type);
                if (++batch.n == kGatherMulti) {
                    FG_TRY(gather_fixed_multi(ctx, batch, rows, n_rows));
                    batch.n = 0;
                }
                continue;
            }
            FG_TRY(take_column(ctx, node_key(pl, n, "take", first_out + (int)i).c_str(), in.cols[i].c, rows, n_rows, &o.c));
        }
        if (batch.n && random_rows) {   // groups of columns that fill a 16-byte record, in order
            int c0 = 0, grp = 0;
            while (c0 < batch.n) {
                GatherCols part;
                int bytes = 0;
                while (c0 < batch.n && part.n < 4 && bytes + batch.width[c0] <= 16) {
                    part.src[part.n] = batch.src[c0];
                    part.out[part.n] = batch.out[c0];
                    part.width[part.n] = batch.width[c0];
                    bytes += batch.width[c0];
                    ++part.n;
                    ++c0;
                }
                FG_TRY(gather_fixed_packed(ctx, node_key(pl, n, "takerec", first_out + grp++).c_str(), part, in.rows, rows, n_rows));
            }
        } else if (batch.n) {
            FG_TRY(gather_fixed_multi(ctx, batch, rows, n_rows));
        }
        for (size_t g0 = 0; g0 < utf8.size(); g0 += 4) {
            const int k = (int)std::min<size_t>(4, utf8.size() - g0);
            if (k == 1) {
                const size_t i = utf8[g0];
                TCol &o = out->cols[(size_t)first_out + i];
                FG_TRY(take_column(ctx, node_key(pl, n, "take", first_out + (int)i).c_str(), in.cols[i].c, rows, n_rows, &o.c));
                o.present = true;
                continue;
            }
            flockgpu_utf8 srcs[4], outs[4];
            int64_t nb[4];
            for (int j = 0; j < k; ++j) srcs[j] = flockgpu_utf8{in.cols[utf8[g0 + (size_t)j]].c.offsets, static_cast<const uint8_t *>(in.cols[utf8[g0 + (size_t)j]].c.values)};
            Utf8MultiGather g;
            FG_TRY(gather_utf8_multi_begin(ctx, node_key(pl, n, "mtake", first_out + (int)utf8[g0]).c_str(), srcs, k, rows, n_rows, &g));
            FG_TRY(gather_utf8_multi_wait(ctx, g));
            FG_TRY(gather_utf8_multi_finish(ctx, g, outs, nb));
            for (int j = 0; j < k; ++j) {
                const size_t i = utf8[g0 + (size_t)j];
                TCol &o = out->cols[(size_t)first_out + i];
                o.c.values = outs[j].data;
                o.c.offsets = outs[j].offsets;
                o.c.bytes = nb[j];
                o.present = true;
                if (in.cols[i].c.valid) {
                    uint8_t *v = nullptr;
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "mtakev", first_out + (int)i).c_str(), (size_t)std::max<int64_t>(n_rows, 0) + 16, &v));
                    FG_TRY(gather_u8(ctx, in.cols[i].c.valid, rows, n_rows, v));
                    o.c.valid = v;
                }
            }
        }
        return FLOCKGPU_OK;
    }

    // The statistics of a column that is NOT a leaf's cost a pass and a host wait on every execute (~20 us): worth it from several
    // thousand rows on, where the dense paths save more than that; below, the hash table answers (a stage plan's operators run on a
    // few thousand filtered rows, and at that size the waits ARE the cost -- DESIGN section 3a).  A leaf column's are cached.
    // rows of the leaf whose column lives at `values` (-1: no leaf's column)
    int64_t leaf_rows_of(const void *values) const {
        if (!values) return -1;
        for (auto &ld : pl->leaves)
            if (!ld.borrowed)
                for (auto &b : ld.cols)
                    if (b.values && b.values == values) return ld.rows;
        return -1;
    }
    bool stats_worth_it(const TCol &c, int64_t rows) const {
        if (rows >= (int64_t(1) << 13)) return true;   // (up to 4096 build rows the one-workgroup LDS join answers without any statistics)
        return leaf_rows_of(c.c.values) >= 0 || leaf_rows_of(c.subset_of) >= 0;
    }
    // (min, max) of an integer column for sizing the dense paths.  A LEAF column's are exact and remembered until the leaf changes
    // (flockgpu_plan::col_stats); a column taken FROM a leaf column (TCol::subset_of) gets the leaf's as bounds when the range they span is
    // still dense for its row count -- no pass, no wait --, its own exact ones otherwise.
    int leaf_col_stats(const void *values, ColType type, int64_t rows, int64_t *mn, int64_t *mx) {
        auto it = pl->col_stats.find(values);
        if (it != pl->col_stats.end() && it->second.rows == rows) {
            *mn = it->second.mn;
            *mx = it->second.mx;
            return FLOCKGPU_OK;
        }
        DevColumn c;
        c.type = type;
        c.values = values;
        FG_TRY(column_minmax(ctx, c, rows, mn, mx));
        pl->col_stats[values] = flockgpu_plan::ColStat{rows, *mn, *mx};
        return FLOCKGPU_OK;
    }
    int int_col_stats(const TCol &c, int64_t rows, int64_t *mn, int64_t *mx) {
        if (leaf_rows_of(c.c.values) >= 0) return leaf_col_stats(c.c.values, c.c.type, rows, mn, mx);
        const int64_t src_rows = c.subset_of && c.subset_of != c.c.values ? leaf_rows_of(c.subset_of) : -1;
        if (src_rows > 0 && rows > 0) {
            FG_TRY(leaf_col_stats(c.subset_of, c.c.type, src_rows, mn, mx));
            if (dense_range_ok(*mn, *mx, rows, c.c.type == ColType::U64)) return FLOCKGPU_OK;
        }
        return column_minmax(ctx, c.c, rows, mn, mx);
    }

    int key_i64(const Node *n, const TCol &c, int64_t rows, const char *what, int64_t **out) {
        if (!c.present) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: key column was not materialised");
        FG_TRY(arena_get_t(ctx, node_key(pl, n, what).c_str(), (size_t)rows + 2, out));
        return widen_to_i64(ctx, c.c, rows, *out);
    }

    // FilterExec as a row selection: the input table and the rows of it the predicate keeps (input order)
    // ---- the general expression evaluator (valprog.hpp): an expression tree -> its postfix program over the columns of `in`
    int val_unsupported(const char *what) { return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: %s", what); }
    int val_full() { return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: expression too large for one program (%d operators, %d columns, %d constants, depth %d)", kValMaxOps, kValMaxCols, kValMaxConsts, kValMaxStack); }
    // static type of `e` over the table's columns (expr_static_type's codes)
    int val_type_of(const Expr *e, const Table &in) const {
        std::vector<Field> sch(in.cols.size());
        for (size_t i = 0; i < in.cols.size(); ++i) sch[i].type = in.cols[i].c.type;
        return expr_static_type(e, sch);
    }
    // the type two operands meet in: the typed one's; two untyped literals take `hint`, else Float64 if one is written as one, else Int64
    int val_common_type(const Expr *a, const Expr *b, const Table &in, int hint) {
        const int ta = val_type_of(a, in), tb = val_type_of(b, in);
        if (ta == -2 || tb == -2 || ta == 4 || tb == 4) return -2;
        if (ta >= 0 && tb >= 0) return ta == tb ? ta : -2;
        if (ta >= 0) return ta;
        if (tb >= 0) return tb;
        if (hint >= 0 && hint <= 3) return hint;
        return (a->kind == EKind::LitF || b->kind == EKind::LitF) ? 3 : 1;
    }
    // pushes `e` (typed `want` where it is an untyped literal); *vt = what was pushed (0..3, 5)
    int val_compile(const Expr *e, const Table &in, ValBuilder &b, int want, int *vt, bool *may_null) {
        auto arith_op = [](const std::string &op) {
            return op == "Plus" ? ValOpKind::Add : op == "Minus" ? ValOpKind::Sub : op == "Multiply" ? ValOpKind::Mul : op == "Divide" ? ValOpKind::Div : ValOpKind::Mod;
        };
        auto cmp_op = [](const std::string &op, ValOpKind *k) {
            if (op == "Eq") *k = ValOpKind::Eq; else if (op == "NotEq") *k = ValOpKind::Ne; else if (op == "Lt") *k = ValOpKind::Lt;
            else if (op == "LtEq") *k = ValOpKind::Le; else if (op == "Gt") *k = ValOpKind::Gt; else if (op == "GtEq") *k = ValOpKind::Ge;
            else return false;
            return true;
        };
        switch (e->kind) {
            case EKind::Col: {
                const TCol &c = in.cols[(size_t)e->col];
                if (c.c.type == ColType::UTF8) return val_unsupported("a Utf8 column inside a computed expression");
                *vt = (int)c.c.type;
                if (c.c.all_null) {
                    *may_null = true;
                    return b.push(ValOpKind::Null, (ValType)*vt, 0) ? FLOCKGPU_OK : val_full();
                }
                if (!c.present) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: expression column was not materialised");
                *may_null = *may_null || c.c.valid != nullptr;
                return b.push(ValOpKind::Col, (ValType)*vt, 0, b.add_col(c.c)) ? FLOCKGPU_OK : val_full();
            }
            case EKind::LitI: case EKind::LitF: {
                int ty = val_type_of(e, in);
                if (ty < 0) ty = want >= 0 && want <= 3 ? want : (e->kind == EKind::LitF ? 3 : 1);
                uint64_t bits = 0;
                if (e->kind == EKind::LitF) {
                    if (ty != 3) return val_unsupported("a Float64 literal against an integer operand (the planner casts the column)");
                    std::memcpy(&bits, &e->f, 8);
                } else if (ty == 3) {
                    const double d = e->big_unsigned ? (double)(uint64_t)e->i : (double)e->i;
                    std::memcpy(&bits, &d, 8);
                } else {
                    if (e->big_unsigned && ty != 2) return val_unsupported("a UInt64 literal above 2^63 against a signed operand");
                    if (ty == 0 && (e->i < INT32_MIN || e->i > INT32_MAX)) return val_unsupported("a literal beyond the Int32 range of its operand");
                    if (ty == 2 && !e->big_unsigned && e->i < 0) return val_unsupported("a negative literal against a UInt64 operand");
                    bits = (uint64_t)e->i;
                }
                *vt = ty;
                return b.push(ValOpKind::Const, (ValType)ty, 0, b.add_const(bits)) ? FLOCKGPU_OK : val_full();
            }
            case EKind::LitB:
                *vt = 5;
                return b.push(ValOpKind::Const, ValType::BOOL, 0, b.add_const(e->i ? 1 : 0)) ? FLOCKGPU_OK : val_full();
            case EKind::LitNull:
                *vt = want >= 0 ? want : 1;
                *may_null = true;
                return b.push(ValOpKind::Null, (ValType)*vt, 0) ? FLOCKGPU_OK : val_full();
            case EKind::LitS:
                return val_unsupported("a Utf8 literal inside a computed expression");
            case EKind::Cast: {
                int from = -1;
                FG_TRY(val_compile(e->l.get(), in, b, -1, &from, may_null));
                if (from > 3 || (int)e->cast_to > 3) return val_unsupported("CAST between other than numeric types inside a computed expression");
                *vt = (int)e->cast_to;
                if (e->try_cast) *may_null = true;
                if (from == *vt) return FLOCKGPU_OK;
                return b.push(e->try_cast ? ValOpKind::TryCast : ValOpKind::Cast, (ValType)from, 1, 0, (ValType)*vt) ? FLOCKGPU_OK : val_full();
            }
            case EKind::Neg: {
                FG_TRY(val_compile(e->l.get(), in, b, want, vt, may_null));
                if (*vt == 2 || *vt > 3) return val_unsupported("unary minus of an unsigned or Boolean value");
                return b.push(ValOpKind::Neg, (ValType)*vt, 1) ? FLOCKGPU_OK : val_full();
            }
            case EKind::Not: {
                FG_TRY(val_compile(e->l.get(), in, b, 5, vt, may_null));
                if (*vt != 5) return val_unsupported("NOT of a value that is not Boolean");
                return b.push(ValOpKind::Not, ValType::BOOL, 1) ? FLOCKGPU_OK : val_full();
            }
            case EKind::IsNull: case EKind::IsNotNull: {
                int ty = -1;
                bool inner_null = false;
                FG_TRY(val_compile(e->l.get(), in, b, -1, &ty, &inner_null));
                *vt = 5;
                return b.push(e->kind == EKind::IsNull ? ValOpKind::IsNull : ValOpKind::IsNotNull, (ValType)ty, 1) ? FLOCKGPU_OK : val_full();
            }
            case EKind::InList: {   // (x = a) OR (x = b) OR ..., x evaluated once per item; NULL when x is
                for (size_t i = 0; i < e->list.size(); ++i) {
                    const int ty = val_common_type(e->l.get(), e->list[i].get(), in, -1);
                    if (ty < 0 || ty > 3) return val_unsupported("IN over operands without one numeric type");
                    int ta = -1, tb = -1;
                    FG_TRY(val_compile(e->l.get(), in, b, ty, &ta, may_null));
                    FG_TRY(val_compile(e->list[i].get(), in, b, ty, &tb, may_null));
                    if (ta != ty || tb != ty) return val_unsupported("IN over operands of different types");
                    if (!b.push(ValOpKind::Eq, (ValType)ty, 2)) return val_full();
                    if (i > 0 && !b.push(ValOpKind::Or, ValType::BOOL, 2)) return val_full();
                }
                if (e->negated && !b.push(ValOpKind::Not, ValType::BOOL, 1)) return val_full();
                *vt = 5;
                return FLOCKGPU_OK;
            }
            case EKind::Bin: {
                if (e->s == "And" || e->s == "Or") {
                    int ta = -1, tb = -1;
                    FG_TRY(val_compile(e->l.get(), in, b, 5, &ta, may_null));
                    FG_TRY(val_compile(e->r.get(), in, b, 5, &tb, may_null));
                    if (ta != 5 || tb != 5) return val_unsupported("AND / OR of values that are not Boolean");
                    *vt = 5;
                    return b.push(e->s == "And" ? ValOpKind::And : ValOpKind::Or, ValType::BOOL, 2) ? FLOCKGPU_OK : val_full();
                }
                ValOpKind ck;
                const bool is_cmp = cmp_op(e->s, &ck);
                if (!is_cmp && e->s != "Plus" && e->s != "Minus" && e->s != "Multiply" && e->s != "Divide" && e->s != "Modulo")
                    return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: binary operator '%s'", e->s.c_str());
                const int ty = val_common_type(e->l.get(), e->r.get(), in, is_cmp ? -1 : want);
                if (ty == 5 && is_cmp && (ck == ValOpKind::Eq || ck == ValOpKind::Ne)) {
                    // Boolean = Boolean
                } else if (ty < 0 || ty > 3) {
                    return val_unsupported("a binary operator over operands without one numeric type (the planner casts them to one)");
                }
                int ta = -1, tb = -1;
                FG_TRY(val_compile(e->l.get(), in, b, ty, &ta, may_null));
                FG_TRY(val_compile(e->r.get(), in, b, ty, &tb, may_null));
                if (ta != ty || tb != ty) return val_unsupported("a binary operator over operands of different types");
                *vt = is_cmp ? 5 : ty;
                return b.push(is_cmp ? ck : arith_op(e->s), (ValType)(ty == 5 ? 1 : ty), 2) ? FLOCKGPU_OK : val_full();
            }
            case EKind::Case: {   // ELSE (or NULL), then from the LAST branch to the first: WHEN, THEN, Select -- the first TRUE WHEN wins
                int ty = val_type_of(e, in);
                if (ty == -1) ty = want >= 0 ? want : 1;
                if (ty < 0 || ty > 3) return val_unsupported("CASE without one numeric result type");
                int te = -1;
                if (e->r) {
                    FG_TRY(val_compile(e->r.get(), in, b, ty, &te, may_null));
                    if (te != ty) return val_unsupported("CASE branches of different types");
                } else {
                    *may_null = true;
                    if (!b.push(ValOpKind::Null, (ValType)ty, 0)) return val_full();
                }
                for (size_t i = e->list.size(); i >= 2; i -= 2) {
                    const Expr *w = e->list[i - 2].get(), *th = e->list[i - 1].get();
                    int tw = -1, tt = -1;
                    if (e->l) {   // CASE x WHEN v: x = v
                        const int tc = val_common_type(e->l.get(), w, in, -1);
                        if (tc < 0 || tc > 3) return val_unsupported("CASE operand and WHEN value without one numeric type");
                        int t1 = -1, t2 = -1;
                        FG_TRY(val_compile(e->l.get(), in, b, tc, &t1, may_null));
                        FG_TRY(val_compile(w, in, b, tc, &t2, may_null));
                        if (t1 != tc || t2 != tc) return val_unsupported("CASE operand and WHEN value of different types");
                        if (!b.push(ValOpKind::Eq, (ValType)tc, 2)) return val_full();
                    } else {
                        FG_TRY(val_compile(w, in, b, 5, &tw, may_null));
                        if (tw != 5) return val_unsupported("a WHEN that is not Boolean");
                    }
                    FG_TRY(val_compile(th, in, b, ty, &tt, may_null));
                    if (tt != ty) return val_unsupported("CASE branches of different types");
                    if (!b.push(ValOpKind::Select, (ValType)ty, 3)) return val_full();
                }
                *vt = ty;
                return FLOCKGPU_OK;
            }
        }
        return val_unsupported("an expression of an unknown kind");
    }

    int filter_rows(const Node *n, Table *in, int32_t **rows, int64_t *n_out) {
        FG_TRY(exec(n->in[0].get(), in));
        PredBuilder b;
        const int rc = compile_pred(n->pred.get(), *in, b);
        if (rc == FLOCKGPU_OK) return pred_to_rows(ctx, node_key(pl, n, "sel").c_str(), b.p, in->rows, rows, n_out);
        if (rc != FLOCKGPU_ERR_UNSUPPORTED) return rc;
        // a predicate the one-pass program has no leaf for (arithmetic inside a comparison, CASE, casts of computed values): the general
        // evaluator writes the same flag words + wave counts, which go through the same scan -> emit
        ValBuilder vb;
        int vt = -1;
        bool may_null = false;
        FG_TRY(val_compile(n->pred.get(), *in, vb, 5, &vt, &may_null));
        if (vt != 5) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: a filter predicate that is not Boolean");
        return valprog_to_rows(ctx, node_key(pl, n, "vprog").c_str(), vb.p, in->rows, rows, n_out);
    }

    // A FilterExec directly under a join (round 6): the filter's surviving rows as a ROW LIST over its input table -- nothing is taken yet.  The
    // join reads the key column through the list, and its result rows compose with it (via[pair row]), so the columns that only travel
    // through -- q3's three Utf8 columns of a person -- are gathered ONCE, at the join's output, instead of after the filter and again after
    // the join (a Utf8 take is three launches and a host wait).  `via` null: `base` is the table itself.
    struct Lazy {
        Table base;
        const int32_t *via = nullptr;
        int64_t rows = 0;
    };
    int exec_lazy(const Node *n, Lazy *z) {
        if (pl->fused[(size_t)n->id].kind == kNone) {
            if (n->kind == NKind::Repartition) return exec_lazy(n->in[0].get(), z);
            if (n->kind == NKind::Filter) {
                int32_t *rows = nullptr;
                int64_t n_out = 0;
                FG_TRY(filter_rows(n, &z->base, &rows, &n_out));
                z->via = rows;
                z->rows = n_out;
                return FLOCKGPU_OK;
            }
            // a projection that only names columns of its input (the planner puts two of them between q8's DISTINCT and its join): the same rows,
            // the named columns
            if (n->kind == NKind::Project) {
                bool plain = !n->proj.empty();
                for (auto &pe : n->proj) plain = plain && pe.first->kind == EKind::Col;
                if (plain) {
                    Lazy below;
                    FG_TRY(exec_lazy(n->in[0].get(), &below));
                    z->base.rows = below.base.rows;
                    z->base.cols.assign(n->schema.size(), TCol{});
                    for (size_t i = 0; i < n->proj.size() && i < n->schema.size(); ++i) {
                        const size_t c = (size_t)n->proj[i].first->col;
                        if (c >= below.base.cols.size()) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: a projection names a column its input does not have");
                        z->base.cols[i] = below.base.cols[c];
                    }
                    z->via = below.via;
                    z->rows = below.rows;
                    return FLOCKGPU_OK;
                }
            }
            // DISTINCT (Int32, Utf8) -- q8's persons: `SELECT DISTINCT p_id, name` under the join -- is a choice of rows of its input: the input's
            // two columns and the chosen rows go up as they are, and the join takes the strings ONCE, for the rows that found a partner (the
            // aggregate's own take of both columns -- a length pass, a scan and an emit for the names -- was the second-largest item of q8 on the
            // generic operators)
            if (n->kind == NKind::Aggregate && pl->twin_sig[(size_t)n->id].empty()) {
                const Node *agg = n, *partial = nullptr;
                if (final_is_identity(n, &partial)) agg = partial;
                const Node *src = agg->in[0].get();
                if (pl->fused[(size_t)agg->id].kind == kNone && pl->twin_sig[(size_t)agg->id].empty() && agg->group.size() == 2 && agg->aggs.empty() &&
                    (size_t)agg->group[0] < src->schema.size() && (size_t)agg->group[1] < src->schema.size() &&
                    src->schema[(size_t)agg->group[0]].type == ColType::I32 && src->schema[(size_t)agg->group[1]].type == ColType::UTF8) {
                    Table in;
                    FG_TRY(exec(src, &in));
                    const TCol &k = in.cols[(size_t)agg->group[0]], &sc = in.cols[(size_t)agg->group[1]];
                    if (k.c.type != ColType::I32 || sc.c.type != ColType::UTF8 || !k.present || !sc.present)
                        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: two-column GROUP BY other than (Int32, Utf8)");
                    if (k.c.valid || sc.c.valid) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: DISTINCT over columns that hold NULLs");
                    int32_t *rows = nullptr;
                    int64_t n_out = 0;
                    FG_TRY(distinct_i32_utf8(ctx, node_key(pl, agg, "dist").c_str(), static_cast<const int32_t *>(k.c.values),
                                             flockgpu_utf8{sc.c.offsets, static_cast<const uint8_t *>(sc.c.values)}, in.rows, &rows, &n_out));
                    z->base.rows = in.rows;
                    z->base.cols = {k, sc};
                    for (size_t i = 0; i < 2 && i < n->schema.size(); ++i) {
                        z->base.cols[i].c.is_ts = n->schema[i].is_ts;
                        z->base.cols[i].c.nullable = n->schema[i].nullable;
                    }
                    z->via = rows;
                    z->rows = n_out;
                    return FLOCKGPU_OK;
                }
            }
        }
        FG_TRY(exec(n, &z->base));
        z->rows = z->base.rows;
        return FLOCKGPU_OK;
    }
    // column `c` of a lazy table as the join sees it: taken through the row list when there is one
    int lazy_key(const Node *n, const Lazy &z, int c, const char *what, TCol *out) {
        *out = z.base.cols[(size_t)c];
        if (!z.via || !out->present) return FLOCKGPU_OK;
        const TCol &src = z.base.cols[(size_t)c];
        FG_TRY(take_column(ctx, node_key(pl, n, what).c_str(), src.c, z.via, z.rows, &out->c));
        out->subset_of = src.subset_of ? src.subset_of : src.c.values;
        return FLOCKGPU_OK;
    }
    // the join's result rows of one side: the pair rows composed with the side's row list, then ONE take of what the output needs
    int take_lazy(const Node *n, const Lazy &z, const int32_t *pair_rows, int64_t pairs, int first_out, const char *what, Table *t) {
        const int32_t *rows = pair_rows;
        if (z.via && pairs > 0) {
            int32_t *composed = nullptr;
            FG_TRY(arena_get_t(ctx, node_key(pl, n, what).c_str(), (size_t)pairs + 4, &composed));
            FG_TRY(gather_i32(ctx, z.via, pair_rows, pairs, composed));
            rows = composed;
        }
        return take_table(n, z.base, n->required, rows, pairs, first_out, t);
    }

    int exec(const Node *n, Table *t) {
        const FusedInfo &fi = pl->fused[(size_t)n->id];
        if (fi.kind != kNone && !leaf_has_validity(fi.leaf_a) && !leaf_has_validity(fi.leaf_b)) {
            const int rc = run_fused(n, fi, t);
            // q4 / q9's fused kernels need dense, increasing auction ids inside a batch (include/flockgpu.h); any other batch runs
            // on the generic operators below
            if (rc != FLOCKGPU_ERR_UNSUPPORTED || (fi.kind != kQ9 && fi.kind != kQ4)) return rc;
            *t = Table{};
        }
        switch (n->kind) {
            case NKind::Scan:
                return scan_table(n, t);
            case NKind::Repartition:
                return exec(n->in[0].get(), t);  // placement is unobservable in one process; the root case is handled by the caller
            case NKind::Filter: {
                Table in;
                int32_t *rows = nullptr;
                int64_t n_out = 0;
                FG_TRY(filter_rows(n, &in, &rows, &n_out));
                t->rows = n_out;
                t->cols.assign(n->schema.size(), TCol{});
                return take_table(n, in, n->required, rows, n_out, 0, t);
            }
            case NKind::Project: {
                Table in;
                FG_TRY(exec(n->in[0].get(), &in));
                t->rows = in.rows;
                t->cols.assign(n->schema.size(), TCol{});
                for (size_t i = 0; i < n->proj.size(); ++i) {
                    const Expr *e = n->proj[i].first.get();
                    TCol &o = t->cols[i];
                    o.c.type = n->schema[i].type;
                    o.c.is_ts = n->schema[i].is_ts;
                    if (!n->required[i]) continue;
                    if (e->kind == EKind::Col) {
                        o = in.cols[(size_t)e->col];
                        continue;
                    }
                    // literal * CAST(Int32 column AS Float64): q1's currency conversion (planner.rs:90), one IEEE multiply
                    const Expr *l = is_bin(e, "Multiply") ? e->l.get() : nullptr, *r = l ? e->r.get() : nullptr;
                    if (l && l->kind != EKind::LitF && l->kind != EKind::LitI) std::swap(l, r);
                    const Expr *c = r ? uncast(r) : nullptr;
                    if (!l || n->schema[i].type != ColType::F64 || (l->kind != EKind::LitF && l->kind != EKind::LitI) || c->kind != EKind::Col ||
                        in.cols[(size_t)c->col].c.type != ColType::I32 || !in.cols[(size_t)c->col].present) {
                        // anything else: the general evaluator (valprog.hpp), one kernel per output column
                        ValBuilder vb;
                        int vt = -1;
                        bool may_null = false;
                        FG_TRY(val_compile(e, in, vb, (int)n->schema[i].type, &vt, &may_null));
                        if (vt != (int)n->schema[i].type) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: a computed projection whose type is not its column's");
                        void *vals = nullptr;
                        uint8_t *vv = nullptr;
                        FG_TRY(arena_get(ctx, node_key(pl, n, "val", (int)i).c_str(), ((size_t)std::max<int64_t>(in.rows, 0) + 2) * 8, &vals));
                        if (may_null) FG_TRY(arena_get_t(ctx, node_key(pl, n, "valv", (int)i).c_str(), (size_t)std::max<int64_t>(in.rows, 0) + 16, &vv));
                        FG_TRY(valprog_to_column(ctx, node_key(pl, n, "vprog", (int)i).c_str(), vb.p, in.rows, n->schema[i].type, vals, vv));
                        o = dev_col(n->schema[i].type, vals, nullptr, 0, n->schema[i].is_ts);
                        o.c.valid = vv;
                        o.c.nullable = true;
                        continue;
                    }
                    double *d = nullptr;
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "f64", (int)i).c_str(), (size_t)in.rows + 2, &d));
                    flockgpu_bid_cols bc{nullptr, nullptr, static_cast<const int32_t *>(in.cols[(size_t)c->col].c.values), nullptr, in.rows};
                    FG_TRY(flockgpu_q1_project(ctx, &bc, l->kind == EKind::LitF ? l->f : (double)l->i, d));
                    o = dev_col(ColType::F64, d);
                    o.c.valid = in.cols[(size_t)c->col].c.valid;   // literal * NULL is NULL
                }
                return FLOCKGPU_OK;
            }
            case NKind::Aggregate: {
                const std::string &sig = pl->twin_sig[(size_t)n->id];
                if (sig.empty()) return exec_aggregate(n, t);
                // the sub-tree occurs more than once: one run per (signature, the leaves its scans read today); the twins get the table --
                // views of the first run's buffers, which nothing writes again before the next execute
                std::string key = sig;
                for (int leaf : pl->twin_leaves[(size_t)n->id]) key += "#" + std::to_string(resolve_leaf(leaf));
                auto it = memo.find(key);
                if (it != memo.end()) {
                    *t = it->second;
                    return FLOCKGPU_OK;
                }
                FG_TRY(exec_aggregate(n, t));
                memo[key] = *t;
                return FLOCKGPU_OK;
            }
            case NKind::Sort:
                return exec_sort(n, -1, t);
            case NKind::Window: {   // ROW_NUMBER() columns first, then the input's (q6_plan.fmt: the WindowAggr schemas)
                Table in;
                FG_TRY(exec(n->in[0].get(), &in));
                const size_t nw = n->win_part.size();
                t->rows = in.rows;
                t->cols.assign(n->schema.size(), TCol{});
                for (size_t w = 0; w < nw; ++w) {
                    t->cols[w].c.type = ColType::U64;
                    t->cols[w].c.nullable = true;
                    if (!n->required[w]) continue;
                    DevColumn keys[4];
                    if (n->win_part[w].size() > 4) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: PARTITION BY more than four columns");
                    for (size_t c = 0; c < n->win_part[w].size(); ++c) {
                        const TCol &k = in.cols[(size_t)n->win_part[w][c]];
                        if (!k.present && !k.c.all_null) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: PARTITION BY column was not materialised");
                        keys[c] = k.c;
                        if (k.c.all_null) { keys[c].type = ColType::I32; keys[c].values = nullptr; }   // every row NULL: one run as far as this key goes
                    }
                    int nk = 0;
                    DevColumn live[4];
                    for (size_t c = 0; c < n->win_part[w].size(); ++c)
                        if (keys[c].values) live[nk++] = keys[c];
                    uint64_t *rank = nullptr;
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "rank", (int)w).c_str(), (size_t)std::max<int64_t>(in.rows, 0) + 2, &rank));
                    FG_TRY(row_number_runs(ctx, node_key(pl, n, "rn", (int)w).c_str(), live, nk, in.rows, rank));
                    t->cols[w] = dev_col(ColType::U64, rank);
                    t->cols[w].c.nullable = true;
                }
                for (size_t i = 0; i < in.cols.size(); ++i) t->cols[nw + i] = in.cols[i];
                return FLOCKGPU_OK;
            }
            case NKind::Limit: {
                // ORDER BY ... LIMIT n (context.rs:549-550): the order is computed for every row, only the first n are taken
                const Node *c = n->in[0].get();
                if (c->kind == NKind::Sort && pl->fused[(size_t)c->id].kind == kNone) return exec_sort(c, n->limit, t);
                FG_TRY(exec(c, t));
                t->rows = std::min<int64_t>(t->rows, n->limit);   // (Utf8 columns keep their byte buffers: the first rows' offsets still hold)
                return FLOCKGPU_OK;
            }
            case NKind::Join: {
                Lazy ZL, ZR;
                FG_TRY(exec_lazy(n->in[0].get(), &ZL));
                FG_TRY(exec_lazy(n->in[1].get(), &ZR));
                const Table &L = ZL.base, &R = ZR.base;
                t->cols.assign(n->schema.size(), TCol{});
                TCol lk, rk;
                FG_TRY(lazy_key(n, ZL, n->on_l, "lzkl", &lk));
                FG_TRY(lazy_key(n, ZR, n->on_r, "lzkr", &rk));
                const bool text_keys = lk.c.type == ColType::UTF8 && rk.c.type == ColType::UTF8 && n->on_l2 < 0;
                if (!text_keys && ((lk.c.type == ColType::U64) != (rk.c.type == ColType::U64) || lk.c.type == ColType::UTF8 || rk.c.type == ColType::UTF8 ||
                                   lk.c.type == ColType::F64 || rk.c.type == ColType::F64))
                    return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: join keys must be integer columns of one signedness, or two Utf8 columns");
                int64_t nl = ZL.rows, nr = ZR.rows;
                if (lk.c.all_null) nl = 0;  // NULL keys never match
                if (rk.c.all_null) nr = 0;
                // (NULLs in a LEAF's key column were left out at feed: inner-join keys are null-droppable.  A computed key column that carries
                // validity bytes -- a grouped MIN / MAX over nothing but NULLs joined on -- is not taken)
                if (lk.c.valid || rk.c.valid || (n->on_l2 >= 0 && (L.cols[(size_t)n->on_l2].c.valid || R.cols[(size_t)n->on_r2].c.valid)))
                    return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: join on a computed column that holds NULLs");
                int32_t *lrows = nullptr, *rrows = nullptr;
                int64_t pairs = 0;
                // one integer key pair whose build side is dense: chain heads addressed by key - min (relops.hpp "dense integer keys").  The
                // table goes on the smaller side, as in join_key64 (which side is hashed is unobservable in the pair multiset).
                if (!text_keys && n->on_l2 < 0 && lk.present && rk.present && nl > 0 && nr > 0) {
                    bool build_right = nl > 4 * nr && nl > 4096;
                    // Which side is a primary key nothing says -- except the previous executes of this node: a build side that repeated keys
                    // (auctions by seller in q3) while the other side never did (persons by id) hands the table to the other side when that one is
                    // not much larger: unique build keys are probed as a filter, one pass and no chains (relops.hpp).  Each orientation keeps its own
                    // finding (".dups" under its own name).
                    const std::string nm_l = node_key(pl, n, "join"), nm_r = node_key(pl, n, "joinr");
                    auto had_dups = [&](const std::string &nm) {
                        auto it = ctx->host_i64.find(nm + ".dups");
                        return it != ctx->host_i64.end() && !it->second.empty();
                    };
                    if (!build_right && had_dups(nm_l) && !had_dups(nm_r) && nr <= 4 * nl) build_right = true;
                    else if (build_right && had_dups(nm_r) && !had_dups(nm_l) && nl <= 16 * nr) build_right = false;
                    const TCol &bk = build_right ? rk : lk, &pk = build_right ? lk : rk;
                    const int64_t nb = build_right ? nr : nl, np = build_right ? nl : nr;
                    int64_t kmin = 0, kmax = 0;
                    // (a join the one-workgroup LDS kernel takes is ONE launch and one wait -- the dense path's fill, build, count, scan and emit
                    // are eight and a wait, whatever the statistics cost; anything larger is worth the pass, leaf or not)
                    const bool look = !join_is_tiny(nl, nr);
                    if (look) FG_TRY(int_col_stats(bk, nb, &kmin, &kmax));
                    if (look && dense_range_ok(kmin, kmax, nb, bk.c.type == ColType::U64)) {
                        FG_TRY(join_dense(ctx, (build_right ? nm_r : nm_l).c_str(), bk.c, nb, kmin, kmax, pk.c, np, build_right ? &rrows : &lrows, build_right ? &lrows : &rrows,
                                          &pairs));
                        t->rows = pairs;
                        FG_TRY(take_lazy(n, ZL, lrows, pairs, 0, "lzrl", t));
                        return take_lazy(n, ZR, rrows, pairs, (int)L.cols.size(), "lzrr", t);
                    }
                }
                int64_t *kl = nullptr, *kr = nullptr;
                if (text_keys) {  // equal strings <-> equal dictionary codes (exact: full byte compare inside utf8_codes)
                    if (!lk.present || !rk.present) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: key column was not materialised");
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "kl").c_str(), (size_t)nl + 2, &kl));
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "kr").c_str(), (size_t)nr + 2, &kr));
                    FG_TRY(utf8_codes(ctx, node_key(pl, n, "codes").c_str(), lk.c, nl, kl, &rk.c, nr, kr));
                    // the codes are row numbers of the left relation -- dense by construction, [0, nl) -- and a right string the left does not
                    // hold gets a negative one: chain heads addressed by the code itself, no second hash table behind the dictionary's
                    if (nl > 0 && nr > 0 && !join_is_tiny(nl, nr)) {
                        DevColumn cl, cr;
                        cl.type = cr.type = ColType::I64;
                        cl.values = kl;
                        cr.values = kr;
                        FG_TRY(join_dense(ctx, node_key(pl, n, "join").c_str(), cl, nl, 0, nl - 1, cr, nr, &lrows, &rrows, &pairs));
                        t->rows = pairs;
                        FG_TRY(take_lazy(n, ZL, lrows, pairs, 0, "lzrl", t));
                        return take_lazy(n, ZR, rrows, pairs, (int)L.cols.size(), "lzrr", t);
                    }
                } else if (n->on_l2 >= 0) {  // two Int32 pairs compare as one 64-bit key
                    TCol lk2, rk2;
                    FG_TRY(lazy_key(n, ZL, n->on_l2, "lzkl2", &lk2));
                    FG_TRY(lazy_key(n, ZR, n->on_r2, "lzkr2", &rk2));
                    if (lk.c.type != ColType::I32 || rk.c.type != ColType::I32 || lk2.c.type != ColType::I32 || rk2.c.type != ColType::I32)
                        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: a two-key join needs Int32 key columns");
                    if (!lk.present || !rk.present || !lk2.present || !rk2.present)
                        return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: key column was not materialised");
                    if (lk2.c.all_null) nl = 0;
                    if (rk2.c.all_null) nr = 0;
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "kl").c_str(), (size_t)nl + 2, &kl));
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "kr").c_str(), (size_t)nr + 2, &kr));
                    FG_TRY(pack_i32_pair(ctx, static_cast<const int32_t *>(lk.c.values), static_cast<const int32_t *>(lk2.c.values), nl, kl));
                    FG_TRY(pack_i32_pair(ctx, static_cast<const int32_t *>(rk.c.values), static_cast<const int32_t *>(rk2.c.values), nr, kr));
                } else if (!join_is_tiny(nl, nr) && nl > 0 && nr > 0 && (lk.c.type == ColType::I32 || lk.c.type == ColType::I64 || lk.c.type == ColType::U64) &&
                           (rk.c.type == ColType::I32 || rk.c.type == ColType::I64 || rk.c.type == ColType::U64)) {   // one integer key pair, not dense: the hashed table, keys read in their own types
                    if (!lk.present || !rk.present) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: key column was not materialised");
                    FG_TRY(join_hashed(ctx, node_key(pl, n, "join").c_str(), lk.c, nl, rk.c, nr, &lrows, &rrows, &pairs));
                    t->rows = pairs;
                    FG_TRY(take_lazy(n, ZL, lrows, pairs, 0, "lzrl", t));
                    return take_lazy(n, ZR, rrows, pairs, (int)L.cols.size(), "lzrr", t);
                } else {
                    FG_TRY(key_i64(n, lk, nl, "kl", &kl));
                    FG_TRY(key_i64(n, rk, nr, "kr", &kr));
                }
                FG_TRY(join_key64(ctx, node_key(pl, n, "join").c_str(), kl, nl, kr, nr, &lrows, &rrows, &pairs));
                t->rows = pairs;
                FG_TRY(take_lazy(n, ZL, lrows, pairs, 0, "lzrl", t));
                return take_lazy(n, ZR, rrows, pairs, (int)L.cols.size(), "lzrr", t);
            }
        }
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: unknown node");
    }

    // SortExec [+ GlobalLimitExec]: the stable order of relops.hpp's sort_rows, then one take of the columns somebody reads
    int exec_sort(const Node *n, int64_t limit, Table *t) {
        Table in;
        FG_TRY(exec(n->in[0].get(), &in));
        std::vector<SortKey> keys;
        for (auto &sc : n->sort_cols) {
            const TCol &c = in.cols[(size_t)sc.col];
            if (!c.present && !c.c.all_null) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: ORDER BY column was not materialised");
            keys.push_back(SortKey{c.c, sc.descending, sc.nulls_first});
        }
        int32_t *rows = nullptr;
        const int32_t *sorted_key = nullptr;   // the first key's column in sorted order, when the sort carried it (ORDER BY one ascending Int32 column)
        FG_TRY(sort_rows(ctx, node_key(pl, n, "sort").c_str(), keys.data(), (int)keys.size(), in.rows, &rows, &sorted_key));
        t->rows = limit >= 0 ? std::min<int64_t>(in.rows, limit) : in.rows;
        t->cols.assign(n->schema.size(), TCol{});
        std::vector<char> need = n->required;
        const size_t kc = n->sort_cols.empty() ? 0 : (size_t)n->sort_cols[0].col;
        if (sorted_key && kc < need.size() && need[kc]) need[kc] = 0;   // (no take of that column: 4 B / row at sorted-row positions, a quarter of sort.sql's takes)
        FG_TRY(take_table(n, in, need, rows, t->rows, 0, t, true));
        if (sorted_key && kc < need.size() && n->required[kc]) {
            TCol &o = t->cols[kc];
            o = dev_col(ColType::I32, const_cast<int32_t *>(sorted_key));
            o.c.is_ts = in.cols[kc].c.is_ts;
            o.c.nullable = in.cols[kc].c.nullable;
        }
        for (size_t i = 0; i < t->cols.size(); ++i) t->cols[i].c.all_null = in.cols[i].c.all_null;
        return FLOCKGPU_OK;
    }

    // Final / FinalPartitioned directly over the Partial of the SAME plan (only a repartition in between): the Partial saw the
    // whole input of this execute, so its output already holds every group exactly once and its state columns ARE the results
    // (COUNT / SUM / MAX / MIN states have the result's type; AVG's (count, sum) state does not).  The stage plans stage.rs cuts keep
    // Partial -> Hash repartition -> FinalPartitioned in one plan (q5.dag, q8.dag): a second hash pass over the groups for nothing.
    bool final_is_identity(const Node *n, const Node **partial) const {
        if (n->mode == "Partial" || n->group.empty()) return false;
        const Node *c = n->in[0].get();
        while (c->kind == NKind::Repartition) c = c->in[0].get();
        if (c->kind != NKind::Aggregate || c->mode != "Partial" || c->group.size() != n->group.size() || c->aggs.size() != n->aggs.size() ||
            c->schema.size() != n->schema.size())
            return false;
        for (size_t i = 0; i < n->group.size(); ++i)
            if (n->group[i] != (int)i) return false;
        for (size_t a = 0; a < n->aggs.size(); ++a)
            if (n->aggs[a].fn != c->aggs[a].fn || n->aggs[a].fn == "avg" || n->aggs[a].arg != (int)(n->group.size() + a)) return false;
        for (size_t i = 0; i < n->schema.size(); ++i)
            if (c->schema[i].type != n->schema[i].type) return false;
        *partial = c;
        return true;
    }

    int exec_aggregate(const Node *n, Table *t) {
        const Node *partial = nullptr;
        if (final_is_identity(n, &partial)) {
            FG_TRY(exec(partial, t));
            for (size_t i = 0; i < n->schema.size() && i < t->cols.size(); ++i) {
                t->cols[i].c.is_ts = n->schema[i].is_ts;
                t->cols[i].c.nullable = n->schema[i].nullable;
            }
            return FLOCKGPU_OK;
        }
        Table in;
        FG_TRY(exec(n->in[0].get(), &in));
        const bool is_final = n->mode != "Partial";
        t->cols.assign(n->schema.size(), TCol{});
        for (size_t i = 0; i < n->schema.size(); ++i) {
            t->cols[i].c.type = n->schema[i].type;
            t->cols[i].c.is_ts = n->schema[i].is_ts;
            t->cols[i].c.nullable = n->schema[i].nullable;
        }
        // ---- no GROUP BY: MAX of one integer column -> one row (NULL over no input)
        if (n->group.empty()) {
            const ColType at = n->aggs.size() == 1 && n->aggs[0].arg >= 0 ? in.cols[(size_t)n->aggs[0].arg].c.type : ColType::UTF8;
            if (n->aggs.size() != 1 || n->aggs[0].fn != "max" || at == ColType::UTF8 || at == ColType::F64 || !in.cols[(size_t)n->aggs[0].arg].present)
                return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: global aggregate other than MAX over an integer column");
            const TCol &a = in.cols[(size_t)n->aggs[0].arg];
            int64_t mx = 0;
            int any = 0;
            if (!a.c.all_null) FG_TRY(reduce_max(ctx, a.c, in.rows, &mx, &any));
            int64_t *d = nullptr;
            FG_TRY(arena_get_t(ctx, node_key(pl, n, "max").c_str(), 2, &d));
            // the one value travels as kernel arguments (a host-to-device copy call costs ~10 us of host time; Int32 reads the low word)
            FG_TRY(fill_words(ctx, FillList().add(d, (uint32_t)(uint64_t)mx, 1).add(reinterpret_cast<uint32_t *>(d) + 1, (uint32_t)((uint64_t)mx >> 32), 1)));
            t->rows = 1;
            t->cols[0] = dev_col(at, d, nullptr, 0, a.c.is_ts);
            t->cols[0].c.nullable = true;
            t->cols[0].c.all_null = !any;
            return FLOCKGPU_OK;
        }
        // ---- DISTINCT (Int32, Utf8)
        if (n->group.size() == 2 && n->aggs.empty()) {
            const TCol &k = in.cols[(size_t)n->group[0]], &s = in.cols[(size_t)n->group[1]];
            if (k.c.type != ColType::I32 || s.c.type != ColType::UTF8 || !k.present || !s.present)
                return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: two-column GROUP BY other than (Int32, Utf8)");
            if (k.c.valid || s.c.valid) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: DISTINCT over columns that hold NULLs");
            int32_t *rows = nullptr;
            int64_t n_out = 0;
            FG_TRY(distinct_i32_utf8(ctx, node_key(pl, n, "dist").c_str(), static_cast<const int32_t *>(k.c.values),
                                     flockgpu_utf8{s.c.offsets, static_cast<const uint8_t *>(s.c.values)}, in.rows, &rows, &n_out));
            t->rows = n_out;
            FG_TRY(take_column(ctx, node_key(pl, n, "take", 0).c_str(), k.c, rows, n_out, &t->cols[0].c));
            FG_TRY(take_column(ctx, node_key(pl, n, "take", 1).c_str(), s.c, rows, n_out, &t->cols[1].c));
            t->cols[0].present = t->cols[1].present = true;
            return FLOCKGPU_OK;
        }
        // ---- GROUP BY one integer column or two Int32 columns; COUNT / MAX / MIN / SUM / AVG of integer columns.
        // Partial: accumulators over the rows -> state columns (agg_state_cols); Final: the same accumulators over the states.
        const bool pair = n->group.size() == 2;
        if (n->group.empty() || n->group.size() > 2) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: GROUP BY more than two columns");
        const TCol &k = in.cols[(size_t)n->group[0]];
        int64_t *keys = nullptr;
        // NULL group keys form ONE group (DataFusion groups NULLs together): an Int32 key column widened to 64 bits has room for a value
        // no Int32 takes, and so have a Utf8 column's dictionary codes (row numbers); other key types with NULLs are handed back
        constexpr int64_t kNullKey = int64_t(1) << 40;
        const bool null_keys = k.c.valid != nullptr;
        // ... and a 64-bit key (Int64 / UInt64 / Timestamp) hands its validity to the GROUP BY itself, which keeps the NULLs in a slot of their own
        const bool wide_null_keys = null_keys && !pair && k.c.type != ColType::I32 && k.c.type != ColType::UTF8 && k.c.type != ColType::F64;
        if (null_keys && pair) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: NULLs in a two-column GROUP BY key");
        // the generic table's keys: every key column normalised to one int64 per row (not needed by the dense path below)
        auto prepare_keys = [&]() -> int {
            if (pair) {
                const TCol &k2 = in.cols[(size_t)n->group[1]];
                if (k.c.type != ColType::I32 || k2.c.type != ColType::I32 || !k.present || !k2.present)
                    return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: two-column GROUP BY other than (Int32, Int32) / (Int32, Utf8)");
                if (k2.c.valid) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: NULLs in a two-column GROUP BY key");
                FG_TRY(arena_get_t(ctx, node_key(pl, n, "gk").c_str(), (size_t)in.rows + 2, &keys));
                FG_TRY(pack_i32_pair(ctx, static_cast<const int32_t *>(k.c.values), static_cast<const int32_t *>(k2.c.values), in.rows, keys));
            } else if (k.c.type == ColType::UTF8) {  // group on the strings' dictionary codes; the key column is taken from the first rows
                if (!k.present) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: key column was not materialised");
                FG_TRY(arena_get_t(ctx, node_key(pl, n, "gk").c_str(), (size_t)in.rows + 2, &keys));
                FG_TRY(utf8_codes(ctx, node_key(pl, n, "codes").c_str(), k.c, in.rows, keys, nullptr, 0, nullptr));
                // (a NULL's bytes are whatever its slot holds -- usually nothing, which is also the empty string's code: NULLs get their own key;
                // the group's key comes out NULL through the validity of its first row, take_column below)
                if (null_keys) FG_TRY(replace_invalid_i64(ctx, keys, k.c.valid, in.rows, kNullKey));
            } else {
                if (k.c.type == ColType::F64) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: GROUP BY a Float64 column");
                FG_TRY(key_i64(n, k, in.rows, "gk", &keys));
                if (null_keys && !wide_null_keys) FG_TRY(replace_invalid_i64(ctx, keys, k.c.valid, in.rows, kNullKey));
            }
            return FLOCKGPU_OK;
        };
        AggSpec specs[kMaxGroupAggs];
        int n_specs = 0;
        struct Out { int first = 0, count = 1; };  // accumulators of aggregate a
        std::vector<Out> outs;
        auto int_col = [&](int c, const char *what) -> const TCol * {
            if (c < 0 || !in.cols[(size_t)c].present || in.cols[(size_t)c].c.type == ColType::UTF8 || in.cols[(size_t)c].c.type == ColType::F64) {
                fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: %s needs an integer column", what);
                return nullptr;
            }
            return &in.cols[(size_t)c];
        };
        for (auto &a : n->aggs) {
            Out o;
            o.first = n_specs;
            o.count = a.fn == "avg" ? 2 : 1;
            if (n_specs + o.count > kMaxGroupAggs) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: more than %d accumulators in one GROUP BY", kMaxGroupAggs);
            // (every accumulator carries its argument's validity: NULLs are skipped, a group without a valid value comes out NULL)
            if (a.fn == "count") {
                if (is_final) {
                    const TCol *st = int_col(a.arg, "the COUNT state");
                    if (!st) return FLOCKGPU_ERR_UNSUPPORTED;
                    specs[n_specs++] = AggSpec{AggOp::SUM_INT, st->c.values, st->c.type, nullptr};
                } else {   // COUNT(*) / COUNT(UInt8(1)) counts rows, COUNT(col) the rows whose col is not NULL
                    specs[n_specs++] = AggSpec{AggOp::COUNT, nullptr, ColType::I64, a.arg >= 0 ? in.cols[(size_t)a.arg].c.valid : nullptr};
                }
            } else if ((a.fn == "max" || a.fn == "min") && a.arg >= 0 && in.cols[(size_t)a.arg].present && in.cols[(size_t)a.arg].c.type == ColType::F64) {
                specs[n_specs++] = AggSpec{a.fn == "max" ? AggOp::MAX_F64 : AggOp::MIN_F64, in.cols[(size_t)a.arg].c.values, ColType::F64, in.cols[(size_t)a.arg].c.valid};
            } else if (a.fn == "max" || a.fn == "min" || a.fn == "sum") {
                const TCol *v = int_col(a.arg, a.fn.c_str());
                if (!v) return FLOCKGPU_ERR_UNSUPPORTED;
                const bool uns = v->c.type == ColType::U64;
                const AggOp op = a.fn == "sum" ? AggOp::SUM_INT : (a.fn == "max" ? (uns ? AggOp::MAX_U : AggOp::MAX_S) : (uns ? AggOp::MIN_U : AggOp::MIN_S));
                specs[n_specs++] = AggSpec{op, v->c.values, v->c.type, v->c.valid};
            } else {  // avg: (count, sum)
                if (is_final) {
                    const TCol *cnt = int_col(a.arg, "the AVG count state");
                    if (!cnt) return FLOCKGPU_ERR_UNSUPPORTED;
                    const TCol &sm = in.cols[(size_t)a.arg2];
                    if (!sm.present || sm.c.type != ColType::F64) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: the AVG sum state must be Float64");
                    specs[n_specs++] = AggSpec{AggOp::SUM_INT, cnt->c.values, cnt->c.type, nullptr};
                    specs[n_specs++] = AggSpec{AggOp::SUM_F64, sm.c.values, ColType::F64, nullptr};
                } else {
                    const TCol *v = int_col(a.arg, "AVG");
                    if (!v) return FLOCKGPU_ERR_UNSUPPORTED;
                    specs[n_specs++] = AggSpec{AggOp::COUNT, nullptr, ColType::I64, v->c.valid};
                    specs[n_specs++] = AggSpec{AggOp::SUM_INT, v->c.values, v->c.type, v->c.valid};
                }
            }
            outs.push_back(o);
        }
        GroupResultN g;
        // A dense integer key without NULLs under integer accumulators without NULLs: the perfect-hash GROUP BY (relops.hpp "dense integer
        // keys") -- slot = key - min over the column's exact range, no hashing, no int64 copy of the key column.  Everything else (Utf8 /
        // two-column keys, NULLs, Float64 accumulators, keys spread wider than their row count) takes the hash table.
        bool dense = !pair && !null_keys && k.present && in.rows > 0 && (k.c.type == ColType::I32 || k.c.type == ColType::I64 || k.c.type == ColType::U64);
        for (int a = 0; a < n_specs && dense; ++a)
            dense = !specs[a].valid && specs[a].op != AggOp::SUM_F64 && specs[a].op != AggOp::MAX_F64 && specs[a].op != AggOp::MIN_F64;
        int64_t kmin = 0, kmax = 0;
        if (dense && !stats_worth_it(k, in.rows)) dense = false;
        if (dense) {
            FG_TRY(int_col_stats(k, in.rows, &kmin, &kmax));
            dense = dense_range_ok(kmin, kmax, in.rows, k.c.type == ColType::U64);
        }
        // A Utf8 key's dictionary codes are row numbers of its own relation: dense by construction.  The code of a group IS a row that
        // carries the group's string, so it also stands in for the first row the key column is taken from.
        bool dense_codes = !pair && !null_keys && k.present && k.c.type == ColType::UTF8 && in.rows > 0 && in.rows < (int64_t(1) << 31);
        for (int a = 0; a < n_specs && dense_codes; ++a)
            dense_codes = !specs[a].valid && specs[a].op != AggOp::SUM_F64 && specs[a].op != AggOp::MAX_F64 && specs[a].op != AggOp::MIN_F64;
        if (dense_codes) {
            FG_TRY(prepare_keys());   // (utf8_codes)
            DevColumn codes;
            codes.type = ColType::I64;
            codes.values = keys;
            FG_TRY(group_by_dense(ctx, node_key(pl, n, "grp").c_str(), codes, in.rows, 0, in.rows - 1, specs, n_specs, &g));
            int32_t *rep = nullptr;
            FG_TRY(arena_get_t(ctx, node_key(pl, n, "rep").c_str(), (size_t)g.n_groups + 4, &rep));
            FG_TRY(narrow_i64_to_i32(ctx, g.keys, g.n_groups, rep));
            g.first_row = rep;
        } else if (dense) {
            FG_TRY(group_by_dense(ctx, node_key(pl, n, "grp").c_str(), k.c, in.rows, kmin, kmax, specs, n_specs, &g));
        } else {
            FG_TRY(prepare_keys());
            FG_TRY(group_by_key64_n(ctx, node_key(pl, n, "grp").c_str(), keys, in.rows, specs, n_specs, &g, wide_null_keys ? k.c.valid : nullptr));
        }
        t->rows = g.n_groups;
        // ---- key columns
        if (pair) {
            int32_t *ka = nullptr, *kb = nullptr;
            FG_TRY(arena_get_t(ctx, node_key(pl, n, "nk").c_str(), (size_t)g.n_groups + 4, &ka));
            FG_TRY(arena_get_t(ctx, node_key(pl, n, "nk2").c_str(), (size_t)g.n_groups + 4, &kb));
            FG_TRY(unpack_i32_pair(ctx, g.keys, g.n_groups, ka, kb));
            t->cols[0] = dev_col(ColType::I32, ka);
            t->cols[1] = dev_col(ColType::I32, kb);
            t->cols[1].c.nullable = n->schema[1].nullable;
        } else if (k.c.type == ColType::UTF8) {
            FG_TRY(take_column(ctx, node_key(pl, n, "take", 0).c_str(), k.c, g.first_row, g.n_groups, &t->cols[0].c));
            t->cols[0].present = true;
        } else if (k.c.type == ColType::I32) {
            int32_t *nk = nullptr;
            FG_TRY(arena_get_t(ctx, node_key(pl, n, "nk").c_str(), (size_t)g.n_groups + 4, &nk));
            FG_TRY(narrow_i64_to_i32(ctx, g.keys, g.n_groups, nk));
            t->cols[0] = dev_col(ColType::I32, nk);
            if (!null_keys) t->cols[0].subset_of = k.subset_of ? k.subset_of : k.c.values;   // (a group's key is one of the input's keys)
            if (null_keys) {   // the NULL group's key is NULL again
                uint8_t *kv = nullptr;
                FG_TRY(arena_get_t(ctx, node_key(pl, n, "nkv").c_str(), (size_t)g.n_groups + 16, &kv));
                FG_TRY(valid_from_i64(ctx, g.keys, g.n_groups, kNullKey, kv));
                t->cols[0].c.valid = kv;
            }
        } else {
            t->cols[0] = dev_col(k.c.type, g.keys, nullptr, 0, k.c.is_ts);
            if (wide_null_keys) t->cols[0].c.valid = g.key_valid;
            else if (!null_keys) t->cols[0].subset_of = k.subset_of ? k.subset_of : k.c.values;
        }
        t->cols[0].c.nullable = n->schema[0].nullable;
        // ---- aggregate / state columns
        size_t oc = n->group.size();
        for (size_t ai = 0; ai < n->aggs.size(); ++ai) {
            const Agg &a = n->aggs[ai];
            const Out &o = outs[ai];
            auto narrow_if_i32 = [&](uint64_t *acc, ColType want, bool ts, size_t col) -> int {
                if (want == ColType::I32) {
                    int32_t *v = nullptr;
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "av", (int)col).c_str(), (size_t)g.n_groups + 4, &v));
                    FG_TRY(narrow_i64_to_i32(ctx, reinterpret_cast<const int64_t *>(acc), g.n_groups, v));
                    t->cols[col] = dev_col(ColType::I32, v);
                } else {
                    t->cols[col] = dev_col(want, acc, nullptr, 0, ts);
                }
                t->cols[col].c.nullable = true;
                return FLOCKGPU_OK;
            };
            if (a.fn == "avg") {
                if (is_final) {
                    double *avg = nullptr;
                    uint8_t *av = nullptr;
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "av", (int)oc).c_str(), (size_t)g.n_groups + 2, &avg));
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "avv", (int)oc).c_str(), (size_t)g.n_groups + 16, &av));
                    FG_TRY(avg_finish(ctx, reinterpret_cast<const double *>(g.agg[o.first + 1]), g.agg[o.first], g.n_groups, avg));
                    FG_TRY(valid_from_i64(ctx, reinterpret_cast<const int64_t *>(g.agg[o.first]), g.n_groups, 0, av));   // AVG over no valid value is NULL
                    t->cols[oc] = dev_col(ColType::F64, avg);
                    t->cols[oc].c.nullable = true;
                    t->cols[oc].c.valid = av;
                    oc += 1;
                } else {
                    double *sum = nullptr;
                    FG_TRY(arena_get_t(ctx, node_key(pl, n, "av", (int)oc + 1).c_str(), (size_t)g.n_groups + 2, &sum));
                    FG_TRY(i64_to_f64(ctx, reinterpret_cast<const int64_t *>(g.agg[o.first + 1]), g.n_groups, sum));
                    t->cols[oc] = dev_col(ColType::U64, g.agg[o.first]);
                    t->cols[oc + 1] = dev_col(ColType::F64, sum);
                    t->cols[oc].c.nullable = t->cols[oc + 1].c.nullable = true;
                    oc += 2;
                }
            } else {
                const ColType want = n->schema[oc].type;
                const bool f64_acc = specs[o.first].op == AggOp::MAX_F64 || specs[o.first].op == AggOp::MIN_F64;
                if (want == ColType::UTF8 || (want == ColType::F64) != f64_acc)
                    return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: %s of a column into a column of another kind", a.fn.c_str());
                FG_TRY(narrow_if_i32(g.agg[o.first], want, n->schema[oc].is_ts, oc));
                if (a.fn != "count") t->cols[oc].c.valid = g.agg_valid[o.first];   // MIN / MAX / SUM over nothing but NULLs is NULL; COUNT(col) is 0
                oc += 1;
            }
        }
        return FLOCKGPU_OK;
    }
};

// ------------------------------------------------------------------ Arrow export
struct BatchPriv {
    std::shared_ptr<HostBlock> block;
    std::vector<const void *> buffers;
    std::vector<ArrowArray *> child_ptrs;
    std::vector<std::unique_ptr<ArrowArray>> children;
    std::vector<std::unique_ptr<BatchPriv>> child_priv;
};
void release_child(ArrowArray *a) {
    if (a) a->release = nullptr;  // memory belongs to the parent struct array
}
void release_batch(ArrowArray *a) {
    if (!a || !a->release) return;
    BatchPriv *p = static_cast<BatchPriv *>(a->private_data);
    if (p) {
        for (auto &c : p->children)
            if (c && c->release) c->release(c.get());
        delete p;
    }
    a->release = nullptr;
}
constexpr const char *kPartitionScheme = "flockgpu/fmix32-mulhi/v1";
constexpr const char *kPartitionSchemeKey = "flockgpu.partition_scheme";
// Arrow C Data Interface metadata: int32 pair count, then per pair int32 key length, key, int32 value length, value (native endian)
std::string encode_metadata(const char *key, const char *value) {
    std::string m;
    auto put = [&](int32_t v) { m.append(reinterpret_cast<const char *>(&v), 4); };
    put(1);
    put((int32_t)strlen(key));
    m += key;
    put((int32_t)strlen(value));
    m += value;
    return m;
}
// value of `key` in an Arrow metadata blob; false when absent (or the blob is malformed)
bool find_metadata(const char *meta, const char *key, std::string *value) {
    if (!meta) return false;
    int32_t n = 0;
    std::memcpy(&n, meta, 4);
    const char *p = meta + 4;
    for (int32_t i = 0; i < n && n < 4096; ++i) {
        int32_t kl = 0, vl = 0;
        std::memcpy(&kl, p, 4);
        if (kl < 0 || kl > (1 << 20)) return false;
        const char *k = p + 4;
        std::memcpy(&vl, k + kl, 4);
        if (vl < 0 || vl > (1 << 20)) return false;
        const char *v = k + kl + 4;
        if ((size_t)kl == strlen(key) && !std::memcmp(k, key, (size_t)kl)) {
            value->assign(v, (size_t)vl);
            return true;
        }
        p = v + vl;
    }
    return false;
}

struct SchemaPriv {
    std::string format, name, metadata;
    std::vector<ArrowSchema *> child_ptrs;
    std::vector<std::unique_ptr<ArrowSchema>> children;
};
void release_schema(ArrowSchema *s) {
    if (!s || !s->release) return;
    SchemaPriv *p = static_cast<SchemaPriv *>(s->private_data);
    if (p) {
        for (auto &c : p->children)
            if (c && c->release) c->release(c.get());
        delete p;
    }
    s->release = nullptr;
}
void make_schema(ArrowSchema *s, const char *format, const char *name, bool nullable) {
    SchemaPriv *p = new SchemaPriv();
    p->format = format;
    p->name = name;
    std::memset(s, 0, sizeof *s);
    s->format = p->format.c_str();
    s->name = p->name.c_str();
    s->flags = nullable ? ARROW_FLAG_NULLABLE : 0;
    s->release = release_schema;
    s->private_data = p;
}
void add_schema_child(ArrowSchema *parent, const char *format, const char *name, bool nullable) {
    SchemaPriv *p = static_cast<SchemaPriv *>(parent->private_data);
    p->children.emplace_back(new ArrowSchema());
    make_schema(p->children.back().get(), format, name, nullable);
    p->child_ptrs.push_back(p->children.back().get());
    parent->children = p->child_ptrs.data();
    parent->n_children = (int64_t)p->child_ptrs.size();
}
void export_schema(const Node *root, ArrowSchema *out, bool shuffled = false) {
    make_schema(out, "+s", "", false);
    if (shuffled) {   // the partitions of a shuffling stage say how their rows were placed (flockgpu_plan.h "hash placement")
        SchemaPriv *p = static_cast<SchemaPriv *>(out->private_data);
        p->metadata = encode_metadata(kPartitionSchemeKey, kPartitionScheme);
        out->metadata = p->metadata.data();
    }
    for (auto &f : root->schema) {
        DevColumn c;
        c.type = f.type;
        c.is_ts = f.is_ts;
        add_schema_child(out, format_of(c), f.name.c_str(), f.nullable);
    }
}

// Copies the table's buffers into ONE pinned block (async, then a single synchronisation) and exposes `n_parts`
// record batches over it: batch p = rows [part_off[p], part_off[p + 1]) through the Arrow `offset` field of its children.
int export_batches(flockgpu_ctx *ctx, const Table &t, const std::vector<int64_t> &part_off, ArrowArray *out_batches) {
    const int n_parts = (int)part_off.size() - 1;
    struct Slot { size_t at = 0, bytes = 0; const void *dev = nullptr; };
    std::vector<Slot> values(t.cols.size()), offsets(t.cols.size()), validity(t.cols.size()), valid_bytes(t.cols.size());
    size_t total = 0;
    auto reserve = [&](Slot &s, const void *dev, size_t bytes) {
        s.at = total;
        s.bytes = bytes;
        s.dev = dev;
        total += (bytes + 63) & ~size_t(63);
    };
    for (size_t i = 0; i < t.cols.size(); ++i) {
        const TCol &c = t.cols[i];
        if (!c.present) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: output column %zu was not materialised", i);
        if (c.c.type == ColType::UTF8) {
            reserve(offsets[i], c.c.offsets, (size_t)(t.rows + 1) * 4);
            reserve(values[i], c.c.values, (size_t)std::max<int64_t>(c.c.bytes, 0));
        } else {
            reserve(values[i], c.c.values, (size_t)t.rows * col_width(c.c.type));
        }
        if (c.c.all_null || c.c.valid) reserve(validity[i], nullptr, (size_t)(t.rows + 7) / 8 + 8);
        if (c.c.valid && !c.c.all_null) reserve(valid_bytes[i], c.c.valid, (size_t)t.rows);   // one byte per row on the device: packed into bits below
    }
    auto block = std::make_shared<HostBlock>();
    block->ptr = pool().get(std::max<size_t>(total, 64), &block->cap);
    if (!block->ptr) return fail(ctx, FLOCKGPU_ERR_OOM, "plan execute: pinned host allocation of %zu bytes failed", total);
    uint8_t *base = static_cast<uint8_t *>(block->ptr);
    for (auto *group : {&values, &offsets, &valid_bytes})
        for (auto &s : *group)
            if (s.bytes && s.dev) FG_HIP(ctx, hipMemcpyAsync(base + s.at, s.dev, s.bytes, hipMemcpyDeviceToHost, ctx->stream));
    for (auto &s : validity)
        if (s.bytes) std::memset(base + s.at, 0, s.bytes);
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < t.cols.size(); ++i)   // validity bytes -> the Arrow bitmap (bit r of the column = row r; batches address it through `offset`)
        if (valid_bytes[i].bytes) {
            const uint8_t *vb = base + valid_bytes[i].at;
            uint8_t *bits = base + validity[i].at;
            for (int64_t r = 0; r < t.rows; ++r)
                if (vb[r]) bits[r >> 3] |= (uint8_t)(1u << (r & 7));
        }
    for (int p = 0; p < n_parts; ++p) {
        ArrowArray *a = &out_batches[p];
        BatchPriv *bp = new BatchPriv();
        bp->block = block;
        std::memset(a, 0, sizeof *a);
        const int64_t lo = part_off[(size_t)p], len = part_off[(size_t)p + 1] - lo;
        a->length = len;
        bp->buffers = {nullptr};
        a->n_buffers = 1;
        a->buffers = bp->buffers.data();
        for (size_t i = 0; i < t.cols.size(); ++i) {
            const TCol &c = t.cols[i];
            bp->children.emplace_back(new ArrowArray());
            bp->child_priv.emplace_back(new BatchPriv());
            ArrowArray *ch = bp->children.back().get();
            BatchPriv *cp = bp->child_priv.back().get();
            std::memset(ch, 0, sizeof *ch);
            ch->length = len;
            ch->offset = lo;
            const bool has_bitmap = c.c.all_null || (c.c.valid && t.rows > 0);
            const void *valid = has_bitmap ? base + validity[i].at : nullptr;
            ch->null_count = c.c.all_null ? len : 0;
            if (!c.c.all_null && valid_bytes[i].bytes) {
                const uint8_t *vb = base + valid_bytes[i].at;
                int64_t nulls = 0;
                for (int64_t r = lo; r < lo + len; ++r) nulls += vb[r] ? 0 : 1;
                ch->null_count = nulls;
            }
            if (c.c.type == ColType::UTF8) cp->buffers = {valid, base + offsets[i].at, base + values[i].at};
            else cp->buffers = {valid, base + values[i].at};
            ch->n_buffers = (int64_t)cp->buffers.size();
            ch->buffers = cp->buffers.data();
            ch->release = release_child;
            bp->child_ptrs.push_back(ch);
        }
        a->children = bp->child_ptrs.data();
        a->n_children = (int64_t)bp->child_ptrs.size();
        a->release = release_batch;
        a->private_data = bp;
    }
    return FLOCKGPU_OK;
}

int run_plan(flockgpu_plan *plan, bool partitioned, ArrowSchema *out_schema, ArrowArray *out_batches, int capacity, int *n_out) {
    flockgpu_ctx *ctx = plan->ctx;
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const Node *root = plan->ir.root.get();
    Exec ex{plan, ctx};
    Table t;
    std::vector<int64_t> part_off;
    if (partitioned && root->kind == NKind::Repartition) {
        if (root->n_parts > capacity) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: %d output partitions, room for %d", root->n_parts, capacity);
        // Filter -> Repartition (stage 0 of planner.rs:152-171: the rows a filter keeps are what travels): the filter's row selection
        // and the partition's send order are COMPOSED and the columns taken once, from the filter's input -- materialising the
        // filtered table first moved every column twice (the three Utf8 columns of q3's persons among them)
        const Node *below = root->in[0].get();
        while (below->kind == NKind::Repartition) below = below->in[0].get();
        const bool compose = below->kind == NKind::Filter && plan->fused[(size_t)below->id].kind == kNone && below->schema.size() == root->schema.size() &&
                             root->in[0]->schema.size() == root->schema.size();
        Table in;
        int32_t *sel = nullptr;      // compose: rows of `in` the filter keeps
        int64_t n_sel = 0;
        if (compose) FG_TRY(ex.filter_rows(below, &in, &sel, &n_sel));
        else FG_TRY(ex.exec(root->in[0].get(), &in));
        const int64_t n_rows = compose ? n_sel : in.rows;
        // RepartitionExec Hash(exprs, n): equal keys must meet in one partition; hashing the first key column already
        // guarantees that, and which partition a key lands on is unobservable
        TCol k = in.cols[(size_t)root->hash_cols[0]];
        if (!k.present) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: key column was not materialised");
        if (compose) FG_TRY(take_column(ctx, node_key(plan, root, "pkcol").c_str(), in.cols[(size_t)root->hash_cols[0]].c, sel, n_sel, &k.c));
        int64_t *keys = nullptr;
        if (k.c.type == ColType::UTF8) {
            FG_TRY(arena_get_t(ctx, node_key(plan, root, "pk").c_str(), (size_t)n_rows + 2, &keys));
            FG_TRY(hash_utf8_i64(ctx, k.c, n_rows, keys));
        } else {
            FG_TRY(ex.key_i64(root, k, n_rows, "pk", &keys));
        }
        if (k.c.valid) FG_TRY(replace_invalid_i64(ctx, keys, k.c.valid, n_rows, 0));   // (NULL keys: one key as far as placement goes; the slot's bytes are unspecified)
        int32_t *rows = nullptr;
        if (root->hash_diff) {
            // HashDiff(exprs, n): every DISTINCT key a partition of its own (session.rs:236-253: n is COUNT(DISTINCT key), counted by the host
            // just before).  A stable sort of the rows by the key groups them, input order kept inside a key; the partitions come out in key
            // order (which partition a key gets is as unobservable as under Hash).  More distinct keys than n: the host's count does not fit
            // the data -- refused; fewer: the partitions behind the last key are empty.
            if (k.c.type == ColType::UTF8 || k.c.type == ColType::F64) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: HashDiff on a key that is not an integer column");
            if (k.c.valid) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: HashDiff on a key column that holds NULLs");
            if (root->hash_cols.size() != 1) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan execute: HashDiff on more than one key column");
            const SortKey sk{k.c, false, false};
            FG_TRY(sort_rows(ctx, node_key(plan, root, "hdsort").c_str(), &sk, 1, n_rows, &rows));
            int32_t *starts = nullptr;
            int64_t n_keys = 0;
            FG_TRY(key_run_starts(ctx, node_key(plan, root, "hdruns").c_str(), keys, rows, n_rows, &starts, &n_keys));
            if (n_keys > root->n_parts)
                return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: HashDiff names %d partitions, the key column holds %lld distinct keys", root->n_parts, (long long)n_keys);
            std::vector<int32_t> h_starts((size_t)n_keys);
            if (n_keys) {
                FG_HIP(ctx, hipMemcpyAsync(h_starts.data(), starts, sizeof(int32_t) * (size_t)n_keys, hipMemcpyDeviceToHost, ctx->stream));
                FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
            part_off.assign((size_t)root->n_parts + 1, n_rows);
            for (int64_t g = 0; g < n_keys; ++g) part_off[(size_t)g] = h_starts[(size_t)g];
        } else
        FG_TRY(partition_rows_key64(ctx, node_key(plan, root, "part").c_str(), keys, n_rows, root->n_parts, &rows, &part_off));
        if (compose) {   // send order over the filter's input: sel[rows[i]]
            int32_t *composed = nullptr;
            FG_TRY(arena_get_t(ctx, node_key(plan, root, "rows").c_str(), (size_t)n_rows + 4, &composed));
            FG_TRY(gather_i32(ctx, sel, rows, n_rows, composed));
            rows = composed;
        }
        t.rows = n_rows;
        t.cols.assign(root->schema.size(), TCol{});
        FG_TRY(ex.take_table(root, in, std::vector<char>(root->schema.size(), 1), rows, n_rows, 0, &t));
    } else {
        FG_TRY(ex.exec(root, &t));
        part_off = {0, t.rows};
    }
    if ((int)part_off.size() - 1 > capacity) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: no room for the output batch");
    FG_TRY(export_batches(ctx, t, part_off, out_batches));
    plan->fed_bytes = 0;  // export synchronised the stream: every borrowed buffer has been read
    export_schema(root, out_schema, partitioned && root->kind == NKind::Repartition);
    *n_out = (int)part_off.size() - 1;
    return FLOCKGPU_OK;
}

}  // namespace

extern "C" {

int flockgpu_plan_create_ex(flockgpu_ctx *ctx, const char *plan_json, size_t len, uint32_t flags, flockgpu_plan **out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!plan_json || !out) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_create: null argument");
    if (flags & ~FLOCKGPU_PLAN_GENERIC_ONLY) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_create: unknown flag bits 0x%x", flags & ~FLOCKGPU_PLAN_GENERIC_ONLY);
    *out = nullptr;
    std::unique_ptr<flockgpu_plan> pl(new flockgpu_plan());
    pl->ctx = ctx;
    FG_TRY(parse_and_build(ctx, plan_json, len, pl.get(), flags));
    *out = pl.release();
    return FLOCKGPU_OK;
}

int flockgpu_plan_create(flockgpu_ctx *ctx, const char *plan_json, size_t len, flockgpu_plan **out) {
    return flockgpu_plan_create_ex(ctx, plan_json, len, 0, out);
}

int flockgpu_plan_recognise(const char *plan_json, size_t len, int *query) {
    if (!plan_json || !query) return FLOCKGPU_ERR_INVALID;
    flockgpu_plan pl;
    const int rc = parse_and_build(nullptr, plan_json, len, &pl);
    if (rc != FLOCKGPU_OK) return rc;
    *query = pl.query;
    return FLOCKGPU_OK;
}

int flockgpu_plan_explain(const char *plan_json, size_t len, char *out, size_t capacity) {
    if (!plan_json || !out || !capacity) return FLOCKGPU_ERR_INVALID;
    flockgpu_plan pl;
    const int rc = parse_and_build(nullptr, plan_json, len, &pl);
    const std::string text = rc == FLOCKGPU_OK ? pl.description : (rc == FLOCKGPU_ERR_UNSUPPORTED ? "unsupported: " + pl.ir.why : std::string("plan JSON error"));
    snprintf(out, capacity, "%s", text.c_str());
    return rc;
}

void flockgpu_plan_destroy(flockgpu_plan *plan) {
    if (!plan) return;
#ifdef FLOCKGPU_EXPERIMENTAL
    if (plan->ring_ppw) ApiClock::dump();
#endif
    flockgpu_ctx *ctx = plan->ctx;
    prefetch_drop(plan);
    if (plan->copy_stream) {
        (void)hipStreamSynchronize(plan->copy_stream);
        (void)hipStreamDestroy(plan->copy_stream);
        (void)hipEventDestroy(plan->copy_done);
        (void)hipEventDestroy(plan->append_done);
    }
    if (plan->async_pending) {   // an execute nobody waited for: let it finish, drop what it produced
        if (ctx_wait(ctx) == FLOCKGPU_OK) {
            for (int i = 0; i < plan->async_n; ++i)
                if (plan->async_batches[(size_t)i].release) plan->async_batches[(size_t)i].release(&plan->async_batches[(size_t)i]);
            if (plan->async_schema.release) plan->async_schema.release(&plan->async_schema);
        }
        plan->async_pending = false;
        if (ctx->plan_in_flight == plan) ctx->plan_in_flight = nullptr;
    }
    (void)hipStreamSynchronize(ctx->stream);
    char prefix[64];
    snprintf(prefix, sizeof prefix, "plan%p.", (const void *)plan);
    for (auto it = ctx->arena.begin(); it != ctx->arena.end();) {
        if (it->first.compare(0, strlen(prefix), prefix) == 0) {
            if (it->second.ptr) dev_free(ctx, it->second.ptr);
            it = ctx->arena.erase(it);
        } else {
            ++it;
        }
    }
    // host-side caches keyed by the same names (tile schedules of scan.hpp) describe buffers that no longer exist
    for (auto it = ctx->host_i64.begin(); it != ctx->host_i64.end();) it = it->first.compare(0, strlen(prefix), prefix) == 0 ? ctx->host_i64.erase(it) : std::next(it);
    for (auto it = ctx->host_u64.begin(); it != ctx->host_u64.end();) it = it->first.compare(0, strlen(prefix), prefix) == 0 ? ctx->host_u64.erase(it) : std::next(it);
    for (auto it = ctx->pinned.begin(); it != ctx->pinned.end();) {
        if (it->first.compare(0, strlen(prefix), prefix) == 0) {
            if (it->second.ptr) (void)hipHostFree(it->second.ptr);
            it = ctx->pinned.erase(it);
        } else {
            ++it;
        }
    }
    for (int k = 0; k < kStageChunks; ++k) {
        if (plan->stage[k]) (void)hipHostFree(plan->stage[k]);
        if (plan->stage_done[k]) (void)hipEventDestroy(plan->stage_done[k]);
    }
    delete plan;
}

int flockgpu_plan_query(const flockgpu_plan *plan) { return plan ? plan->query : 0; }
const char *flockgpu_plan_description(const flockgpu_plan *plan) { return plan ? plan->description.c_str() : nullptr; }
int flockgpu_plan_num_inputs(const flockgpu_plan *plan) { return plan ? (int)plan->ir.leaves.size() : 0; }
const char *flockgpu_plan_input_name(const flockgpu_plan *plan, int input) {
    if (!plan || input < 0 || input >= (int)plan->ir.leaves.size()) return nullptr;
    return plan->ir.leaves[(size_t)input].relation.c_str();
}
int flockgpu_plan_input_matches(const flockgpu_plan *plan, int input, const struct ArrowSchema *schema) {
    if (!plan || !schema || input < 0 || input >= (int)plan->ir.leaves.size()) return 0;
    const Leaf &lf = plan->ir.leaves[(size_t)input];
    for (size_t c = 0; c < lf.schema.size(); ++c)
        if (lf.needed[c] && find_child(schema, lf.schema[c].name) < 0) return 0;
    return 1;
}
int flockgpu_plan_output_partitions(const flockgpu_plan *plan) {
    if (!plan) return 0;
    return plan->ir.root->kind == NKind::Repartition ? plan->ir.root->n_parts : 1;
}
int flockgpu_plan_is_shuffling(const flockgpu_plan *plan) { return plan && plan->ir.root->kind == NKind::Repartition ? 1 : 0; }

// feed_data_sources for one leaf.  validate_only: every check of the feed, nothing moved (flockgpu_plan_feed_pane asks before it
// opens a new pane, so that a refused feed leaves the ring as it was).
static int feed_impl(flockgpu_plan *plan, int input, const struct ArrowSchema *schema, const struct ArrowArray *const *batches, int n_batches,
                     bool validate_only) {
    flockgpu_ctx *ctx = plan->ctx;
    if (!schema || input < 0 || input >= (int)plan->ir.leaves.size() || n_batches < 0 || (n_batches && !batches))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: bad argument");
    if (plan->async_pending) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: an asynchronous execute is in flight (flockgpu_plan_wait first)");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    const Leaf &lf = plan->ir.leaves[(size_t)input];
    LeafData &ld = plan->leaves[(size_t)input];
    if (ld.borrowed) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: input %d shares another plan's relation; reset the plan first", input);
    // hash placement (flockgpu_plan.h): the co-partitioned inputs of one invocation must all have been placed by the same scheme
    int scheme_state = plan->scheme_state;
    std::string scheme_tag = plan->scheme_tag;
    if (lf.co_partitioned) {
        bool any_rows = false;
        for (int b = 0; b < n_batches; ++b) any_rows = any_rows || (batches[b] && batches[b]->length > 0);
        if (any_rows) {
            std::string tag;
            const int tagged = find_metadata(schema->metadata, kPartitionSchemeKey, &tag) ? 1 : 0;
            if (tagged && tag != kPartitionScheme)
                return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: partitions placed by scheme '%s', this library places by '%s': one key would meet two consumers",
                            tag.c_str(), kPartitionScheme);
            if (scheme_state >= 0 && scheme_state != tagged)
                return fail(ctx, FLOCKGPU_ERR_INVALID,
                            "plan_feed: mixed hash placement -- this stage was fed partitions tagged '%s' and untagged ones (another engine's hash): "
                            "every producer of a stage must place rows the same way", kPartitionScheme);
            scheme_state = tagged;
            scheme_tag = tag;
        }
    }
    std::vector<int> child(lf.schema.size(), -1);
    for (size_t c = 0; c < lf.schema.size(); ++c) {
        if (!lf.needed[c]) continue;
        child[c] = find_child(schema, lf.schema[c].name);
        if (child[c] < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: column '%s' missing from the fed schema", lf.schema[c].name.c_str());
        if (!format_ok(lf.schema[c], schema->children[child[c]]->format))
            return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed: column '%s' has Arrow format '%s', the plan scans %s", lf.schema[c].name.c_str(),
                        schema->children[child[c]]->format, type_name(lf.schema[c]));
    }
    // validate everything before the first copy, so that a rejected feed leaves the leaf as it was
    int64_t add_rows = 0;
    std::vector<int64_t> add_bytes(lf.schema.size(), 0);
    std::vector<std::vector<int64_t>> keep((size_t)n_batches);  // per batch: the rows that survive NULL dropping (empty: all)
    std::vector<char> filtered((size_t)n_batches, 0);
    std::vector<char> with_valid((size_t)n_batches * lf.schema.size(), 0);   // (batch, column): NULLs that stay -> validity bytes go along
    for (int b = 0; b < n_batches; ++b) {
        const ArrowArray *rb = batches[b];
        if (!rb || rb->n_children < schema->n_children) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: batch %d does not match the schema", b);
        const int64_t n = rb->length;
        if (n == 0) continue;
        for (size_t c = 0; c < lf.schema.size(); ++c) {
            if (child[c] < 0) continue;
            const ArrowArray *a = rb->children[child[c]];
            if (!a || a->length < rb->offset + n) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: column '%s' is shorter than its batch", lf.schema[c].name.c_str());
            if (lf.schema[c].type == ColType::UTF8) {
                if (a->n_buffers < 3 || !a->buffers[1]) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: Utf8 column '%s' without 3 buffers", lf.schema[c].name.c_str());
            } else if (a->n_buffers < 2 || !a->buffers[1]) {
                return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: column '%s' without a data buffer", lf.schema[c].name.c_str());
            }
            const int64_t off = a->offset + rb->offset;
            if (!validity_has_nulls(a, off, n)) continue;
            // NULLs.  Where dropping the row cannot change the result (inner-join keys, columns only compared under AND, MAX arguments of a
            // global aggregate) the row is left out here; everywhere else the column travels with a validity byte per row and the
            // operators honour it (NULLs skipped by COUNT(col) / MIN / MAX / SUM / AVG, one group for NULL keys, NULLs in the output)
            if (!lf.null_droppable[c]) {
                if (plan->ring_ppw && plan->ring_q5)   // (q5's ring keeps Partial COUNT groups of plain columns; the rows ring carries validity along)
                    return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed: column '%s' holds NULLs that reach the output; q5's pane ring keeps no validity", lf.schema[c].name.c_str());
                with_valid[(size_t)b * lf.schema.size() + c] = 1;
                continue;
            }
            const uint8_t *bits = static_cast<const uint8_t *>(a->buffers[0]);
            std::vector<int64_t> &k = keep[(size_t)b];
            if (!filtered[(size_t)b]) {
                k.resize((size_t)n);
                for (int64_t i = 0; i < n; ++i) k[(size_t)i] = i;
                filtered[(size_t)b] = 1;
            }
            size_t w = 0;
            for (int64_t i : k)
                if ((bits[(off + i) >> 3] >> ((off + i) & 7)) & 1) k[w++] = i;
            k.resize(w);
        }
        const int64_t n_keep = filtered[(size_t)b] ? (int64_t)keep[(size_t)b].size() : n;
        for (size_t c = 0; c < lf.schema.size(); ++c) {
            if (child[c] < 0 || lf.schema[c].type != ColType::UTF8) continue;
            const ArrowArray *a = rb->children[child[c]];
            const int32_t *so = static_cast<const int32_t *>(a->buffers[1]) + a->offset + rb->offset;
            if (!filtered[(size_t)b]) add_bytes[c] += (int64_t)so[n] - so[0];
            else for (int64_t i : keep[(size_t)b]) add_bytes[c] += so[i + 1] - so[i];
        }
        add_rows += n_keep;
    }
    if (ld.rows + add_rows >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed: more than 2^31 rows per relation");
    for (size_t c = 0; c < lf.schema.size(); ++c)
        if (child[c] >= 0 && lf.schema[c].type == ColType::UTF8 && ld.cols[c].bytes + add_bytes[c] >= (int64_t(1) << 31))
            return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed: Utf8 column exceeds 2^31 bytes");
    if (validate_only) return FLOCKGPU_OK;
    plan->has_retained = false;
    plan->col_stats.clear();
    plan->scheme_state = scheme_state;
    plan->scheme_tag = scheme_tag;
    for (size_t c = 0; c < lf.schema.size(); ++c) {
        if (child[c] < 0) continue;
        DevBuf &dc = ld.cols[c];
        void *p = nullptr;
        if (lf.schema[c].type == ColType::UTF8) {
            FG_TRY(grow(ctx, leaf_key(plan, input, (int)c, "off"), (size_t)(ld.rows + 1) * 4, (size_t)(ld.rows + add_rows + 1) * 4 + 16, &p));
            if (!dc.offsets) FG_HIP(ctx, hipMemsetAsync(p, 0, 4, ctx->stream));
            dc.offsets = static_cast<int32_t *>(p);
            FG_TRY(grow(ctx, leaf_key(plan, input, (int)c, "bytes"), (size_t)dc.bytes, (size_t)(dc.bytes + add_bytes[c]) + 16, &p));
            dc.values = p;
        } else {
            const size_t w = col_width(lf.schema[c].type);
            FG_TRY(grow(ctx, leaf_key(plan, input, (int)c, "val"), (size_t)ld.rows * w, (size_t)(ld.rows + add_rows) * w + 16, &p));
            dc.values = p;
        }
        // a column that already carries validity keeps room for every row of the leaf: the scan fills the rows fed after the last NULL with
        // "valid" up to the leaf's row count (a batch without NULLs behind one with NULLs wrote past the validity buffer before: refused by
        // the runtime as an invalid memset, found by the rows ring's exactly-sized buffers)
        if (dc.valid && dc.valid_rows > 0) {
            void *vp = nullptr;
            FG_TRY(grow(ctx, leaf_key(plan, input, (int)c, "valid"), (size_t)dc.valid_rows, (size_t)(ld.rows + add_rows) + 64, &vp));
            dc.valid = static_cast<uint8_t *>(vp);
        }
    }
    const int64_t rows_end = ld.rows + add_rows;   // the leaf's rows once this feed is in
    std::vector<uint8_t> tmp_vals, tmp_bytes;
    std::vector<int32_t> tmp_off;
    struct Rebase { int32_t *data; int64_t n; int32_t delta; };
    std::vector<Rebase> rebases;  // Utf8 offsets are rebased on the device once their copy is queued
    for (int b = 0; b < n_batches; ++b) {
        const ArrowArray *rb = batches[b];
        int64_t n = rb->length;
        if (n == 0) continue;
        const bool filt = filtered[(size_t)b] != 0;
        const std::vector<int64_t> &k = keep[(size_t)b];
        for (size_t c = 0; c < lf.schema.size(); ++c) {
            if (child[c] < 0) continue;
            DevBuf &dc = ld.cols[c];
            const ArrowArray *a = rb->children[child[c]];
            const int64_t off = a->offset + rb->offset;
            if (with_valid[(size_t)b * lf.schema.size() + c]) {   // this batch's validity bytes (of the rows that stay), behind what the column holds
                const uint8_t *bits = static_cast<const uint8_t *>(a->buffers[0]);
                std::vector<uint8_t> vb;
                if (filt) for (int64_t i : k) vb.push_back((bits[(off + i) >> 3] >> ((off + i) & 7)) & 1);
                else for (int64_t i = 0; i < n; ++i) vb.push_back((bits[(off + i) >> 3] >> ((off + i) & 7)) & 1);
                void *vp = nullptr;
                FG_TRY(grow(ctx, leaf_key(plan, input, (int)c, "valid"), (size_t)dc.valid_rows, (size_t)rows_end + 64, &vp));   // (room for every row of this feed)
                dc.valid = static_cast<uint8_t *>(vp);
                if (dc.valid_rows < ld.rows) FG_HIP(ctx, hipMemsetAsync(dc.valid + dc.valid_rows, 1, (size_t)(ld.rows - dc.valid_rows), ctx->stream));
                FG_TRY(h2d(plan, dc.valid + ld.rows, vb.data(), vb.size()));
                FG_TRY(flush_jobs(plan));   // (the bytes live in a temporary)
                dc.valid_rows = ld.rows + (int64_t)vb.size();
            }
            if (lf.schema[c].type == ColType::UTF8) {
                const int32_t *so = static_cast<const int32_t *>(a->buffers[1]) + off;
                const uint8_t *src = static_cast<const uint8_t *>(a->buffers[2]);
                int64_t b0 = so[0], nbytes = (int64_t)so[n] - b0, rows_in = n;
                if (filt) {  // the surviving rows, compacted on the host (small: the outputs of aggregate stages)
                    tmp_off.assign(1, 0);
                    tmp_bytes.clear();
                    for (int64_t i : k) {
                        tmp_bytes.insert(tmp_bytes.end(), src + so[i], src + so[i + 1]);
                        tmp_off.push_back((int32_t)tmp_bytes.size());
                    }
                    so = tmp_off.data();
                    src = tmp_bytes.data();
                    b0 = 0;
                    nbytes = (int64_t)tmp_bytes.size();
                    rows_in = (int64_t)k.size();
                }
                // offsets[rows + 1 .. rows + n] = source offsets rebased onto the column's byte cursor, on the device
                FG_TRY(h2d(plan, dc.offsets + ld.rows + 1, so + 1, (size_t)rows_in * 4));
                rebases.push_back(Rebase{dc.offsets + ld.rows + 1, rows_in, (int32_t)(dc.bytes - b0)});
                if (nbytes) FG_TRY(h2d(plan, static_cast<uint8_t *>(dc.values) + dc.bytes, src + b0, (size_t)nbytes));
                if (filt) FG_TRY(flush_jobs(plan));  // the compacted copies live in temporaries that the next column reuses
                dc.bytes += nbytes;
            } else {
                const size_t w = col_width(lf.schema[c].type);
                const uint8_t *src = static_cast<const uint8_t *>(a->buffers[1]) + (size_t)off * w;
                int64_t rows_in = n;
                if (filt) {
                    tmp_vals.resize(k.size() * w);
                    for (size_t i = 0; i < k.size(); ++i) std::memcpy(tmp_vals.data() + i * w, src + (size_t)k[i] * w, w);
                    src = tmp_vals.data();
                    rows_in = (int64_t)k.size();
                }
                FG_TRY(h2d(plan, static_cast<uint8_t *>(dc.values) + (size_t)ld.rows * w, src, (size_t)rows_in * w));
                if (filt) FG_TRY(flush_jobs(plan));
            }
        }
        ld.rows += filt ? (int64_t)k.size() : n;
        if (filt) ld.dropped += n - (int64_t)k.size();
    }
    FG_TRY(flush_jobs(plan));  // every pageable buffer of this feed, several staging lanes side by side
    for (auto &r : rebases) FG_TRY(add_i32(ctx, r.data, r.n, r.delta));  // (stream-ordered behind the copies above)
    return FLOCKGPU_OK;  // no host wait: the copies are ordered before the plan's kernels on the ctx stream
}

int flockgpu_plan_feed(flockgpu_plan *plan, int input, const struct ArrowSchema *schema, const struct ArrowArray *const *batches,
                       int n_batches) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    if (plan->ring_ppw) return fail(plan->ctx, FLOCKGPU_ERR_INVALID, "plan_feed: the plan has an open pane ring (flockgpu_plan_feed_pane, or flockgpu_plan_ring_close first)");
    return feed_impl(plan, input, schema, batches, n_batches, false);
}

// ------------------------------------------------------------------ pane ring
// Drops the oldest pane: the rows (bytes) of the panes that stay move to the front of a second buffer, which then trades places
// with the first (source and destination of one hipMemcpyAsync must not overlap); q5's group arrays likewise.
// The retained tail of a leaf buffer moves to the front of the buffer's twin, and the two swap names.  The twin is asked for the KEPT bytes
// only: what the next pane adds is the feed's business (grow), so in the steady state both twins have reached the size two panes need and
// nothing is allocated.  (Asking for the current buffer's capacity made every window reallocate: the arena over-allocates by an eighth when
// it grows, so each twin was always an eighth smaller than the other -- 0.39 ms of hipFree + hipMalloc per window, growing without bound.)
// (min_bytes: room the twin must have beyond what is kept -- a validity column is written up to the leaf's row count when it is scanned)
static int ring_swap_in(flockgpu_ctx *ctx, const std::string &key, const void *src, size_t keep_bytes, size_t min_bytes, void **out) {
    const std::string alt = key + ".alt";
    void *p = nullptr;
    FG_TRY(arena_get(ctx, alt.c_str(), std::max<size_t>(std::max(keep_bytes, min_bytes), 64), &p));
    if (keep_bytes) FG_HIP(ctx, hipMemcpyAsync(p, src, keep_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    std::swap(ctx->arena[key], ctx->arena[alt]);
    *out = p;
    return FLOCKGPU_OK;
}
static int ring_drop_oldest(flockgpu_plan *plan) {
    flockgpu_ctx *ctx = plan->ctx;
    if (plan->ring_n == 0) return FLOCKGPU_OK;
    API_CLOCK("ring_drop_oldest");
    if (plan->ring_q5) {
        const int64_t d = plan->ring_groups.empty() ? 0 : plan->ring_groups[0];
        int64_t total = 0;
        for (int64_t g : plan->ring_groups) total += g;
        if (d && total > d) {
            void *a = nullptr, *c = nullptr;
            FG_TRY(ring_swap_in(ctx, leaf_key(plan, 0, 0, "ring.q5a"), plan->ring_auction + d, (size_t)(total - d) * 4, 0, &a));
            FG_TRY(ring_swap_in(ctx, leaf_key(plan, 0, 0, "ring.q5c"), plan->ring_count + d, (size_t)(total - d) * 4, 0, &c));
            plan->ring_auction = static_cast<int32_t *>(a);
            plan->ring_count = static_cast<uint32_t *>(c);
        }
        if (!plan->ring_groups.empty()) plan->ring_groups.erase(plan->ring_groups.begin());
    }
    for (size_t l = 0; l < plan->leaves.size(); ++l) {
        LeafData &ld = plan->leaves[l];
        const Leaf &lf = plan->ir.leaves[l];
        // (q5's state ring: the leaf holds the newest pane's rows only -- one entry, dropped only when that pane is the oldest, too)
        if (ld.pane_rows.empty() || (plan->ring_q5 && plan->ring_n > 1)) continue;
        const int64_t d = ld.pane_rows[0], keep = ld.rows - d;
        for (size_t c = 0; c < ld.cols.size(); ++c) {
            DevBuf &dc = ld.cols[c];
            if (!dc.values) continue;
            if (dc.valid && dc.valid_rows > 0) {   // the validity bytes of the rows that stay (rows beyond valid_rows are valid by convention)
                const int64_t left = std::max<int64_t>(dc.valid_rows - d, 0);
                if (left > 0 && d > 0) {
                    void *nv = nullptr;
                    FG_TRY(ring_swap_in(ctx, leaf_key(plan, (int)l, (int)c, "valid"), dc.valid + d, (size_t)left, (size_t)keep + 64, &nv));
                    dc.valid = static_cast<uint8_t *>(nv);
                }
                dc.valid_rows = left;
            }
            if (lf.schema[c].type == ColType::UTF8) {
                const int64_t db = ld.pane_bytes[0][c];
                if (keep > 0 && d > 0) {
                    void *no = nullptr, *nb = nullptr;
                    FG_TRY(ring_swap_in(ctx, leaf_key(plan, (int)l, (int)c, "off"), dc.offsets + d, (size_t)(keep + 1) * 4, 0, &no));
                    FG_TRY(ring_swap_in(ctx, leaf_key(plan, (int)l, (int)c, "bytes"), static_cast<uint8_t *>(dc.values) + db, (size_t)(dc.bytes - db), 0, &nb));
                    dc.offsets = static_cast<int32_t *>(no);
                    dc.values = nb;
                    if (db) FG_TRY(add_i32(ctx, dc.offsets, keep + 1, (int32_t)-db));
                } else if (keep == 0 && dc.offsets) {
                    FG_HIP(ctx, hipMemsetAsync(dc.offsets, 0, 4, ctx->stream));
                }
                dc.bytes -= db;
            } else if (keep > 0 && d > 0) {
                const size_t w = col_width(lf.schema[c].type);
                void *nv = nullptr;
                FG_TRY(ring_swap_in(ctx, leaf_key(plan, (int)l, (int)c, "val"), static_cast<uint8_t *>(dc.values) + (size_t)d * w, (size_t)keep * w, 0, &nv));
                dc.values = nv;
            }
        }
        ld.rows = keep;
        ld.pane_rows.erase(ld.pane_rows.begin());
        ld.pane_bytes.erase(ld.pane_bytes.begin());
    }
    plan->ring_first += 1;
    plan->ring_n -= 1;
    if (plan->ring_n == 0) plan->ring_newest_done = false;
    return FLOCKGPU_OK;
}

// q5's state ring: the newest pane's Partial aggregate -- HashAggregateExec(Partial) COUNT GROUP BY auction over the pane's rows,
// counted ONCE -- appended to the retained groups.
static int ring_q5_partial(flockgpu_plan *plan) {
    flockgpu_ctx *ctx = plan->ctx;
    if (plan->ring_newest_done || plan->ring_n == 0) return FLOCKGPU_OK;
    API_CLOCK("ring_q5_partial");
    // the bids sit in whichever of the plan's two `bid` leaves was fed (both scan the same relation)
    const LeafData *ld = nullptr;
    int col = -1;
    for (size_t l = 0; l < plan->leaves.size() && !ld; ++l)
        for (size_t c = 0; c < plan->ir.leaves[l].schema.size(); ++c)
            if (plan->ir.leaves[l].needed[c] && plan->ir.leaves[l].schema[c].type == ColType::I32 && plan->leaves[l].rows > 0 && plan->leaves[l].cols[c].values) {
                ld = &plan->leaves[l];
                col = (int)c;
                break;
            }
    int64_t before = 0;
    for (size_t i = 0; i + 1 < plan->ring_groups.size(); ++i) before += plan->ring_groups[i];
    int64_t groups = 0;
    if (ld) {
        int64_t off[2] = {0, ld->rows};
        int32_t lo[1] = {0}, hi[1] = {1};
        flockgpu_bid_cols bc{static_cast<const int32_t *>(ld->cols[(size_t)col].values), nullptr, nullptr, nullptr, ld->rows};
        flockgpu_windows w{off, 1, lo, hi, 1};
        // The Partial stage per 8192-row TILE (q5.hip: q5_partial_tile_kernel -- one pass, one wait; a tile stands for one input partition
        // of HashAggregateExec(Partial), so an auction may leave the pane in several (auction, count) groups, which the Final side merges).
        // Going through flockgpu_q5_partial_counts instead made every window TWO full q5 calls of different kinds on one ctx -- each
        // re-uploading the other's schedule and declining its speculation: 1.48 ms per window against 1.21 ms for the whole-window feed
        // (bench, round 4).
        Q5TilePartial part;
        FG_TRY(q5_partial_by_tile(ctx, &bc, &w, &part));
        groups = part.offsets[1] - part.offsets[0];
        void *a = nullptr, *c = nullptr;
        FG_TRY(grow(ctx, leaf_key(plan, 0, 0, "ring.q5a"), (size_t)before * 4, (size_t)(before + groups) * 4 + 64, &a));
        FG_TRY(grow(ctx, leaf_key(plan, 0, 0, "ring.q5c"), (size_t)before * 4, (size_t)(before + groups) * 4 + 64, &c));
        plan->ring_auction = static_cast<int32_t *>(a);
        plan->ring_count = static_cast<uint32_t *>(c);
        if (groups) {
            FG_HIP(ctx, hipMemcpyAsync(plan->ring_auction + before, part.auction + part.offsets[0], (size_t)groups * 4, hipMemcpyDeviceToDevice, ctx->stream));
            FG_HIP(ctx, hipMemcpyAsync(plan->ring_count + before, part.count + part.offsets[0], (size_t)groups * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    plan->ring_groups.back() = groups;
    plan->ring_newest_done = true;
    return FLOCKGPU_OK;
}

int flockgpu_plan_ring_open(flockgpu_plan *plan, int panes_per_window) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    flockgpu_ctx *ctx = plan->ctx;
    if (panes_per_window < 1 || panes_per_window > 4096) return fail(ctx, FLOCKGPU_ERR_INVALID, "ring_open: %d panes per window", panes_per_window);
    if (plan->ring_ppw) return fail(ctx, FLOCKGPU_ERR_INVALID, "ring_open: the plan already has an open ring");
    for (auto &ld : plan->leaves)
        if (ld.rows || ld.borrowed) return fail(ctx, FLOCKGPU_ERR_INVALID, "ring_open: the plan holds inputs (flockgpu_plan_reset first)");
    plan->ring_ppw = panes_per_window;
    plan->ring_first = 0;
    plan->ring_n = 0;
    plan->ring_q5 = plan->query == 5 && !plan->generic_only && !exp_env("FLOCKGPU_RING_ROWS");   // (A/B knob: the rows ring for q5, too)
    plan->ring_groups.clear();
    plan->ring_newest_done = false;
    return FLOCKGPU_OK;
}

int flockgpu_plan_ring_state(const flockgpu_plan *plan, int64_t *first_pane, int *n_panes, int *panes_per_window) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    if (first_pane) *first_pane = plan->ring_first;
    if (n_panes) *n_panes = plan->ring_n;
    if (panes_per_window) *panes_per_window = plan->ring_ppw;
    return FLOCKGPU_OK;
}

// Ends a prefetch's host side: the staging threads are joined (every copy is queued on the copy stream by then).  Returns their status.
static int prefetch_join(flockgpu_plan *plan) {
    int rc = FLOCKGPU_OK;
    for (auto &w : plan->pre.workers)
        if (w.joinable()) w.join();
    plan->pre.workers.clear();
    for (int r : plan->pre.rc)
        if (r != FLOCKGPU_OK) rc = r;
    plan->pre.rc.clear();
    return rc;
}
// Drops a pending prefetch (ring close, reset of a plan without a ring, destroy): nothing of it reaches the leaf.
static void prefetch_drop(flockgpu_plan *plan) {
    if (!plan->pre.active) return;
    (void)prefetch_join(plan);
    if (plan->copy_stream) (void)hipStreamSynchronize(plan->copy_stream);
    plan->pre.active = false;
}

int flockgpu_plan_prefetch_pane(flockgpu_plan *plan, int input, int64_t pane_id, const struct ArrowSchema *schema, const struct ArrowArray *const *batches,
                                int n_batches) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    flockgpu_ctx *ctx = plan->ctx;
    if (!plan->ring_ppw) return fail(ctx, FLOCKGPU_ERR_INVALID, "prefetch_pane: the plan has no open ring (flockgpu_plan_ring_open)");
    if (input < 0 || input >= (int)plan->leaves.size() || !schema || !batches || n_batches < 1) return fail(ctx, FLOCKGPU_ERR_INVALID, "prefetch_pane: bad argument");
    if (plan->pre.active) return fail(ctx, FLOCKGPU_ERR_INVALID, "prefetch_pane: pane %lld of input %d is already on its way", (long long)plan->pre.pane, plan->pre.input);
    const int64_t newest = plan->ring_first + plan->ring_n - 1;
    if (plan->ring_n > 0 && pane_id != newest + 1)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "prefetch_pane: pane %lld is not the next one (the ring holds [%lld, %lld])", (long long)pane_id, (long long)plan->ring_first,
                    (long long)newest);
    const Leaf &lf = plan->ir.leaves[(size_t)input];
    if (plan->leaves[(size_t)input].borrowed) return fail(ctx, FLOCKGPU_ERR_INVALID, "prefetch_pane: input %d shares another plan's relation", input);
    FG_HIP(ctx, hipSetDevice(ctx->device));
    // what the pane holds: fixed-width columns without NULLs (anything else takes the ordinary feed, which knows how to rebase and filter)
    std::vector<int> child(lf.schema.size(), -1);
    int64_t rows = 0;
    for (int b = 0; b < n_batches; ++b) {
        if (!batches[b] || batches[b]->n_children < schema->n_children) return fail(ctx, FLOCKGPU_ERR_INVALID, "prefetch_pane: batch %d does not match the schema", b);
        rows += batches[b]->length;
    }
    for (size_t c = 0; c < lf.schema.size(); ++c) {
        if (!lf.needed[c]) continue;
        child[c] = find_child(schema, lf.schema[c].name);
        if (child[c] < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "prefetch_pane: column '%s' missing from the fed schema", lf.schema[c].name.c_str());
        if (!format_ok(lf.schema[c], schema->children[child[c]]->format))
            return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "prefetch_pane: column '%s' has Arrow format '%s', the plan scans %s", lf.schema[c].name.c_str(),
                        schema->children[child[c]]->format, type_name(lf.schema[c]));
        for (int b = 0; b < n_batches; ++b) {
            const ArrowArray *rb = batches[b], *a = rb->children[child[c]];
            if (!a || a->length < rb->offset + rb->length || a->n_buffers < (lf.schema[c].type == ColType::UTF8 ? 3 : 2) || (rb->length > 0 && !a->buffers[1]))
                return fail(ctx, FLOCKGPU_ERR_INVALID, "prefetch_pane: column '%s' of batch %d is malformed", lf.schema[c].name.c_str(), b);
            if (validity_has_nulls(a, a->offset + rb->offset, rb->length))
                return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "prefetch_pane: column '%s' holds NULLs (feed the pane the ordinary way)", lf.schema[c].name.c_str());
        }
    }
    if (rows >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "prefetch_pane: more than 2^31 rows");
    if (!plan->copy_stream) {
        FG_HIP(ctx, hipStreamCreateWithFlags(&plan->copy_stream, hipStreamNonBlocking));
        FG_HIP(ctx, hipEventCreateWithFlags(&plan->copy_done, hipEventDisableTiming));
        FG_HIP(ctx, hipEventCreateWithFlags(&plan->append_done, hipEventDisableTiming));
    }
    // the side buffers and the pinned mirror are the previous prefetch's, too: its copies must have left the mirror (host wait: they were
    // queued a whole window ago) and its appends must have read the side buffers (device-side wait of the copy stream)
    if (plan->copy_done_set) FG_HIP(ctx, hipEventSynchronize(plan->copy_done));
    if (plan->append_done_set) FG_HIP(ctx, hipStreamWaitEvent(plan->copy_stream, plan->append_done, 0));
    // side buffers; pinned sources go straight to the DMA engine, pageable ones through a pinned mirror filled by staging threads
    plan->pre.dev.assign(lf.schema.size(), nullptr);
    plan->pre.dev_off.assign(lf.schema.size(), nullptr);
    plan->pre.bytes.assign(lf.schema.size(), 0);
    plan->pre.rebase.clear();
    std::vector<flockgpu_plan::CopyJob> pieces;   // pageable pieces: dst (device), src (host), bytes
    size_t pageable = 0;
    auto queue_runs = [&](std::vector<flockgpu_plan::CopyJob> &runs) -> int {   // pinned sources straight to the DMA engine, pageable ones in staged chunks
        for (auto &r : runs) {
            const uint8_t *src = static_cast<const uint8_t *>(r.src);
            uint8_t *dst = static_cast<uint8_t *>(r.dst);
            if (host_is_pinned(src) && host_is_pinned(src + r.bytes - 1)) {
                FG_HIP(ctx, hipMemcpyAsync(dst, src, r.bytes, hipMemcpyHostToDevice, plan->copy_stream));
            } else {
                for (size_t done = 0; done < r.bytes; done += kStageChunk) {
                    const size_t n = std::min(kStageChunk, r.bytes - done);
                    pieces.push_back(flockgpu_plan::CopyJob{dst + done, src + done, n});
                    pageable += n;
                }
            }
        }
        return FLOCKGPU_OK;
    };
    for (size_t c = 0; c < lf.schema.size(); ++c) {
        if (child[c] < 0) continue;
        if (lf.schema[c].type == ColType::UTF8) {
            // offsets: every batch's END offsets, raw, one after the other (rebased onto the column's byte cursor when the pane is appended: a run
            // of batches that are slices of one allocation shares its delta); bytes: every batch's byte range, back to back
            int64_t total = 0;
            for (int b = 0; b < n_batches; ++b) {
                const ArrowArray *rb = batches[b], *a = rb->children[child[c]];
                if (!rb->length) continue;
                const int32_t *so = static_cast<const int32_t *>(a->buffers[1]) + a->offset + rb->offset;
                total += (int64_t)so[rb->length] - so[0];
            }
            if (total >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "prefetch_pane: Utf8 column exceeds 2^31 bytes");
            void *d_off = nullptr, *d_bytes = nullptr;
            FG_TRY(arena_get(ctx, leaf_key(plan, input, (int)c, "pre.off").c_str(), (size_t)rows * 4 + 16, &d_off));
            FG_TRY(arena_get(ctx, leaf_key(plan, input, (int)c, "pre").c_str(), (size_t)total + 16, &d_bytes));
            plan->pre.dev[c] = d_bytes;
            plan->pre.dev_off[c] = d_off;
            plan->pre.bytes[c] = total;
            std::vector<flockgpu_plan::CopyJob> off_runs, byte_runs;
            int64_t row_at = 0, byte_at = 0;
            for (int b = 0; b < n_batches; ++b) {
                const ArrowArray *rb = batches[b], *a = rb->children[child[c]];
                const int64_t n = rb->length;
                if (!n) continue;
                const int32_t *so = static_cast<const int32_t *>(a->buffers[1]) + a->offset + rb->offset;
                const uint8_t *sb = static_cast<const uint8_t *>(a->buffers[2]);
                const int64_t nbytes = (int64_t)so[n] - so[0], delta = byte_at - so[0];
                if (nbytes && !sb) return fail(ctx, FLOCKGPU_ERR_INVALID, "prefetch_pane: Utf8 column '%s' without a data buffer", lf.schema[c].name.c_str());
                const uint8_t *osrc = reinterpret_cast<const uint8_t *>(so + 1);
                uint8_t *odst = static_cast<uint8_t *>(d_off) + (size_t)row_at * 4;
                if (!off_runs.empty() && static_cast<const uint8_t *>(off_runs.back().src) + off_runs.back().bytes == osrc) off_runs.back().bytes += (size_t)n * 4;
                else off_runs.push_back(flockgpu_plan::CopyJob{odst, osrc, (size_t)n * 4});
                if (nbytes) {
                    const uint8_t *bsrc = sb + so[0];
                    uint8_t *bdst = static_cast<uint8_t *>(d_bytes) + byte_at;
                    if (!byte_runs.empty() && static_cast<const uint8_t *>(byte_runs.back().src) + byte_runs.back().bytes == bsrc) byte_runs.back().bytes += (size_t)nbytes;
                    else byte_runs.push_back(flockgpu_plan::CopyJob{bdst, bsrc, (size_t)nbytes});
                }
                auto &rbs = plan->pre.rebase;
                if (!rbs.empty() && rbs.back().col == (int)c && rbs.back().delta == delta && rbs.back().row0 + rbs.back().n == row_at) rbs.back().n += n;
                else rbs.push_back(flockgpu_plan::Prefetch::Rebase{(int)c, row_at, n, delta});
                row_at += n;
                byte_at += nbytes;
            }
            FG_TRY(queue_runs(off_runs));
            FG_TRY(queue_runs(byte_runs));
            continue;
        }
        const size_t w = col_width(lf.schema[c].type);
        void *d = nullptr;
        FG_TRY(arena_get(ctx, leaf_key(plan, input, (int)c, "pre").c_str(), (size_t)rows * w + 16, &d));
        plan->pre.dev[c] = d;
        // the column's batches as runs: batches that continue each other in host memory (slices of one allocation -- the reference's own
        // event_bytes_to_batch output is) travel as ONE copy (26 copies of 0.7 MB are 0.4 ms of set-up on the calling thread)
        std::vector<flockgpu_plan::CopyJob> runs;
        size_t at = 0;
        for (int b = 0; b < n_batches; ++b) {
            const ArrowArray *rb = batches[b], *a = rb->children[child[c]];
            const size_t bytes = (size_t)rb->length * w;
            if (!bytes) continue;
            const uint8_t *src = static_cast<const uint8_t *>(a->buffers[1]) + (size_t)(a->offset + rb->offset) * w;
            uint8_t *dst = static_cast<uint8_t *>(d) + at;
            at += bytes;
            if (!runs.empty() && static_cast<const uint8_t *>(runs.back().src) + runs.back().bytes == src) runs.back().bytes += bytes;
            else runs.push_back(flockgpu_plan::CopyJob{dst, src, bytes});
        }
        FG_TRY(queue_runs(runs));
    }
    uint8_t *mirror = nullptr;   // (before the prefetch counts as under way: a refusal here must leave nothing behind that a feed would append)
    if (!pieces.empty()) FG_TRY(pinned_get_t(ctx, leaf_key(plan, input, 0, "pre.stage").c_str(), pageable + 64, &mirror));
    plan->pre.input = input;
    plan->pre.pane = pane_id;
    plan->pre.rows = rows;
    plan->pre.active = true;
    if (pieces.empty()) return FLOCKGPU_OK;
    // (the mirror may still feed the previous prefetch's copies: they were waited for when that pane was appended -- feed_pane orders the
    // ctx stream behind the copy stream, and a host synchronises the ctx stream in every execute)
    const int n_lanes = (int)std::min<size_t>(4, pieces.size());
    auto shared = std::make_shared<std::vector<flockgpu_plan::CopyJob>>(std::move(pieces));
    plan->pre.rc.assign((size_t)n_lanes, FLOCKGPU_OK);
    std::vector<size_t> offs(shared->size());
    size_t run = 0;
    for (size_t i = 0; i < shared->size(); ++i) {
        offs[i] = run;
        run += (*shared)[i].bytes;
    }
    auto offsets = std::make_shared<std::vector<size_t>>(std::move(offs));
    const int device = ctx->device;
    hipStream_t stream = plan->copy_stream;
    for (int l = 0; l < n_lanes; ++l)
        plan->pre.workers.emplace_back([plan, shared, offsets, mirror, l, n_lanes, device, stream] {
            if (hipSetDevice(device) != hipSuccess) {
                plan->pre.rc[(size_t)l] = FLOCKGPU_ERR_HIP;
                return;
            }
            for (size_t i = (size_t)l; i < shared->size(); i += (size_t)n_lanes) {
                const flockgpu_plan::CopyJob &j = (*shared)[i];
                std::memcpy(mirror + (*offsets)[i], j.src, j.bytes);
                if (hipMemcpyAsync(j.dst, mirror + (*offsets)[i], j.bytes, hipMemcpyHostToDevice, stream) != hipSuccess) plan->pre.rc[(size_t)l] = FLOCKGPU_ERR_HIP;
            }
        });
    return FLOCKGPU_OK;
}

int flockgpu_plan_feed_pane(flockgpu_plan *plan, int input, int64_t pane_id, const struct ArrowSchema *schema,
                            const struct ArrowArray *const *batches, int n_batches) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    flockgpu_ctx *ctx = plan->ctx;
    if (!plan->ring_ppw) return fail(ctx, FLOCKGPU_ERR_INVALID, "feed_pane: the plan has no open ring (flockgpu_plan_ring_open)");
    API_CLOCK("feed_pane");
    plan->col_stats.clear();
    if (input < 0 || input >= (int)plan->leaves.size()) return fail(ctx, FLOCKGPU_ERR_INVALID, "feed_pane: bad argument");
    const int64_t newest = plan->ring_first + plan->ring_n - 1;
    const bool begin = plan->ring_n == 0 || pane_id == newest + 1;
    if (!begin && pane_id != newest)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "feed_pane: pane %lld out of order -- the ring holds panes [%lld, %lld], only pane %lld or %lld can be fed",
                    (long long)pane_id, (long long)plan->ring_first, (long long)newest, (long long)newest, (long long)(newest + 1));
    if (begin && plan->ring_n == plan->ring_ppw)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "feed_pane: the ring is full (%d panes): flockgpu_plan_reset retires the oldest before pane %lld begins",
                    plan->ring_ppw, (long long)pane_id);
    // a pane that flockgpu_plan_prefetch_pane already brought over: its side buffers are appended below, device to device
    const bool from_prefetch = n_batches == 0 && !schema && plan->pre.active && plan->pre.input == input && plan->pre.pane == pane_id;
    if (plan->pre.active && !from_prefetch && plan->pre.input == input)
        return fail(ctx, FLOCKGPU_ERR_INVALID, "feed_pane: pane %lld of input %d was prefetched: feed it with (NULL, NULL, 0) first", (long long)plan->pre.pane, plan->pre.input);
    if (from_prefetch) {
        const int rc_pre = prefetch_join(plan);
        if (rc_pre != FLOCKGPU_OK) {
            plan->pre.active = false;
            return fail(ctx, rc_pre, "feed_pane: the prefetch of pane %lld failed while staging", (long long)pane_id);
        }
    }
    // every check of the feed first: a refused feed must leave the ring as it was
    {
        API_CLOCK2(c1, "feed_pane.validate");
        if (n_batches > 0 || schema) FG_TRY(feed_impl(plan, input, schema, batches, n_batches, true));
    }
    // what `begin` changes, kept so that a feed that fails AFTER its validation (allocation, HIP error, staging) leaves the rows ring as it
    // was.  q5's state ring cannot go back -- the closing pane's rows are folded into its groups by then -- so there a failed
    // first feed leaves the new pane begun and empty: the same pane id is simply fed again.
    const int ring_n_before = plan->ring_n;
    const int64_t ring_first_before = plan->ring_first;
    const bool newest_done_before = plan->ring_newest_done;
    auto undo_begin = [&]() {
        if (!begin || plan->ring_q5) return;
        plan->ring_n = ring_n_before;
        plan->ring_first = ring_first_before;
        plan->ring_newest_done = newest_done_before;
        for (auto &l : plan->leaves) {
            if (!l.pane_rows.empty()) l.pane_rows.pop_back();
            if (!l.pane_bytes.empty()) l.pane_bytes.pop_back();
        }
    };
    if (begin) {
        if (plan->ring_q5 && plan->ring_n > 0) {   // the pane that closes leaves its groups behind, its rows go
            FG_TRY(ring_q5_partial(plan));
            for (auto &ld : plan->leaves) {
                ld.rows = 0;
                for (auto &c : ld.cols) c.bytes = 0;
                ld.pane_rows.clear();
                ld.pane_bytes.clear();
            }
        }
        if (plan->ring_n == 0) plan->ring_first = pane_id;
        plan->ring_n += 1;
        for (size_t l = 0; l < plan->leaves.size(); ++l) {
            plan->leaves[l].pane_rows.push_back(0);
            plan->leaves[l].pane_bytes.emplace_back(plan->leaves[l].cols.size(), 0);
        }
        if (plan->ring_q5) plan->ring_groups.push_back(0);
        plan->ring_newest_done = false;
    }
    if (from_prefetch) {
      // a later column's failure must not leave an earlier column's byte cursor advanced: the retry would append behind the orphaned bytes
      // while offsets[ld.rows] still names the old end
      std::vector<int64_t> pre_bytes_before(plan->leaves[(size_t)input].cols.size());
      for (size_t c = 0; c < pre_bytes_before.size(); ++c) pre_bytes_before[c] = plan->leaves[(size_t)input].cols[c].bytes;
      const int rc_append = [&]() -> int {
        LeafData &ld = plan->leaves[(size_t)input];
        const Leaf &lf = plan->ir.leaves[(size_t)input];
        FG_HIP(ctx, hipEventRecord(plan->copy_done, plan->copy_stream));           // every copy of the pane is queued (prefetch_join)
        plan->copy_done_set = true;
        FG_HIP(ctx, hipStreamWaitEvent(ctx->stream, plan->copy_done, 0));          // ... and the appends wait for them on the device
        for (size_t c = 0; c < lf.schema.size(); ++c) {
            if (!plan->pre.dev[c]) continue;
            // a column that carries validity from an earlier, ordinarily fed pane keeps room for every row of the leaf, exactly as feed_impl
            // does: the scan fills the rows behind the last NULL with "valid" up to the leaf's row count (the prefetched rows were
            // appended to the values only, and that fill ran past a validity buffer sized for the panes before)
            DevBuf &dc = ld.cols[c];
            if (dc.valid && dc.valid_rows > 0) {
                void *vp = nullptr;
                FG_TRY(grow(ctx, leaf_key(plan, input, (int)c, "valid"), (size_t)dc.valid_rows, (size_t)(ld.rows + plan->pre.rows) + 64, &vp));
                dc.valid = static_cast<uint8_t *>(vp);
            }
            if (lf.schema[c].type == ColType::UTF8) {   // offsets behind the leaf's, rebased onto its byte cursor run by run; bytes behind its bytes
                void *po = nullptr, *pb = nullptr;
                FG_TRY(grow(ctx, leaf_key(plan, input, (int)c, "off"), (size_t)(ld.rows + 1) * 4, (size_t)(ld.rows + plan->pre.rows + 1) * 4 + 16, &po));
                if (!dc.offsets) FG_HIP(ctx, hipMemsetAsync(po, 0, 4, ctx->stream));
                dc.offsets = static_cast<int32_t *>(po);
                FG_TRY(grow(ctx, leaf_key(plan, input, (int)c, "bytes"), (size_t)dc.bytes, (size_t)(dc.bytes + plan->pre.bytes[c]) + 16, &pb));
                dc.values = pb;
                if (dc.bytes + plan->pre.bytes[c] >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "feed_pane: Utf8 column exceeds 2^31 bytes");
                if (plan->pre.rows) {
                    FG_HIP(ctx, hipMemcpyAsync(dc.offsets + ld.rows + 1, plan->pre.dev_off[c], (size_t)plan->pre.rows * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    for (auto &rb : plan->pre.rebase)
                        if (rb.col == (int)c) FG_TRY(add_i32(ctx, dc.offsets + ld.rows + 1 + rb.row0, rb.n, (int32_t)(dc.bytes + rb.delta)));
                }
                if (plan->pre.bytes[c])
                    FG_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(pb) + dc.bytes, plan->pre.dev[c], (size_t)plan->pre.bytes[c], hipMemcpyDeviceToDevice, ctx->stream));
                dc.bytes += plan->pre.bytes[c];
                ld.pane_bytes.back()[c] += plan->pre.bytes[c];
                continue;
            }
            const size_t w = col_width(lf.schema[c].type);
            void *p = nullptr;
            FG_TRY(grow(ctx, leaf_key(plan, input, (int)c, "val"), (size_t)ld.rows * w, (size_t)(ld.rows + plan->pre.rows) * w + 16, &p));
            ld.cols[c].values = p;
            if (plan->pre.rows)
                FG_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(p) + (size_t)ld.rows * w, plan->pre.dev[c], (size_t)plan->pre.rows * w, hipMemcpyDeviceToDevice, ctx->stream));
        }
        FG_HIP(ctx, hipEventRecord(plan->append_done, ctx->stream));
        plan->append_done_set = true;
        ld.rows += plan->pre.rows;
        ld.pane_rows.back() += plan->pre.rows;
        plan->has_retained = false;
        plan->fed_bytes += 1;   // (reset waits for the stream before the host's buffers may go)
        plan->pre.active = false;
        plan->ring_newest_done = false;
        if (plan->ring_q5) plan->ring_groups.back() = 0;
        return FLOCKGPU_OK;
      }();
      if (rc_append != FLOCKGPU_OK) {   // (the prefetch stays pending: the same call can be repeated)
          LeafData &ld = plan->leaves[(size_t)input];
          for (size_t c = 0; c < pre_bytes_before.size(); ++c) {
              if (!ld.pane_bytes.empty()) ld.pane_bytes.back()[c] -= ld.cols[c].bytes - pre_bytes_before[c];
              ld.cols[c].bytes = pre_bytes_before[c];
          }
          undo_begin();
      }
      return rc_append;
    }
    if (n_batches == 0) return FLOCKGPU_OK;   // an empty pane still advances the ring
    LeafData &ld = plan->leaves[(size_t)input];
    const int64_t rows_before = ld.rows;
    std::vector<int64_t> bytes_before(ld.cols.size());
    for (size_t c = 0; c < ld.cols.size(); ++c) bytes_before[c] = ld.cols[c].bytes;
    {
        API_CLOCK2(c2, "feed_pane.feed");
        const int rc_feed = feed_impl(plan, input, schema, batches, n_batches, false);
        if (rc_feed != FLOCKGPU_OK) {
            // (rows of batches that did get through stay appended to the leaf's buffers but are not counted: ld.rows is what the ring reads)
            if (!plan->ring_q5) {
                ld.rows = rows_before;
                for (size_t c = 0; c < ld.cols.size(); ++c) {
                    ld.cols[c].bytes = bytes_before[c];
                    ld.cols[c].valid_rows = std::min(ld.cols[c].valid_rows, rows_before);
                }
            }
            undo_begin();
            return rc_feed;
        }
    }
    ld.pane_rows.back() += ld.rows - rows_before;
    for (size_t c = 0; c < ld.cols.size(); ++c) ld.pane_bytes.back()[c] += ld.cols[c].bytes - bytes_before[c];
    plan->ring_newest_done = false;   // (more rows of the newest pane: its groups are counted again from its rows)
    if (plan->ring_q5) plan->ring_groups.back() = 0;
    return FLOCKGPU_OK;
}

int flockgpu_plan_ring_close(flockgpu_plan *plan) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    if (plan->async_pending) return fail(plan->ctx, FLOCKGPU_ERR_INVALID, "ring_close: an asynchronous execute is in flight");
    prefetch_drop(plan);
#ifdef FLOCKGPU_EXPERIMENTAL
    ApiClock::dump();
#endif
    plan->ring_ppw = 0;
    plan->ring_n = 0;
    plan->ring_q5 = false;
    plan->ring_groups.clear();
    plan->ring_newest_done = false;
    return flockgpu_plan_reset(plan);
}

int flockgpu_plan_reset(flockgpu_plan *plan) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    if (plan->async_pending) return fail(plan->ctx, FLOCKGPU_ERR_INVALID, "plan_reset: an asynchronous execute is in flight (flockgpu_plan_wait first)");
    plan->col_stats.clear();
    API_CLOCK("reset");
    // borrowed pinned buffers may still be read by the DMA engine: the caller is about to drop them
    if (plan->fed_bytes) (void)hipStreamSynchronize(plan->ctx->stream);
    plan->fed_bytes = 0;
    plan->has_retained = false;   // (its columns may alias the leaves, and the arena buffers behind it are reused by the next execute)
    plan->scheme_state = -1;
    plan->scheme_tag.clear();
    if (plan->ring_ppw) {   // the window ends: the oldest pane of a full ring goes, the others stay on the device
        if (plan->ring_n == plan->ring_ppw) FG_TRY(ring_drop_oldest(plan));
        return FLOCKGPU_OK;
    }
    for (auto &ld : plan->leaves) {
        ld.rows = 0;
        ld.dropped = 0;
        ld.pane_rows.clear();
        ld.pane_bytes.clear();
        for (auto &c : ld.cols) {
            c.bytes = 0;
            c.valid_rows = 0;
            if (ld.borrowed) c.values = nullptr, c.offsets = nullptr, c.valid = nullptr;   // the donor's memory: never grown or appended to from here
        }
        ld.borrowed = false;
    }
    return FLOCKGPU_OK;
}

int flockgpu_plan_feed_shared(flockgpu_plan *plan, int input, const flockgpu_plan *donor, int donor_input) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    flockgpu_ctx *ctx = plan->ctx;
    if (!donor || donor == plan || input < 0 || input >= (int)plan->ir.leaves.size() || donor_input < 0 || donor_input >= (int)donor->ir.leaves.size())
        return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_shared: bad argument");
    if (donor->ctx != ctx) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_shared: the two plans live on different contexts (streams)");
    if (plan->ring_ppw || donor->ring_ppw) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_shared: a plan with an open pane ring feeds panes (flockgpu_plan_feed_pane)");
    const Leaf &lf = plan->ir.leaves[(size_t)input], &df = donor->ir.leaves[(size_t)donor_input];
    LeafData &ld = plan->leaves[(size_t)input];
    const LeafData &dd = donor->leaves[(size_t)donor_input];
    if (ld.rows != 0 || ld.borrowed) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_shared: input %d already holds rows", input);
    if (dd.borrowed) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_shared: the donor's relation is itself shared");
    // what the donor left out because of NULLs it was allowed to drop need not be droppable here
    if (dd.dropped) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed_shared: the donor dropped %lld rows with NULLs", (long long)dd.dropped);
    std::vector<int> from(lf.schema.size(), -1);
    for (size_t c = 0; c < lf.schema.size(); ++c) {
        if (!lf.needed[c]) continue;
        for (size_t d = 0; d < df.schema.size(); ++d)
            if (df.needed[d] && df.schema[d].name == lf.schema[c].name && df.schema[d].type == lf.schema[c].type) from[c] = (int)d;
        if (from[c] < 0)
            return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed_shared: the donor does not hold column '%s' as %s", lf.schema[c].name.c_str(), type_name(lf.schema[c]));
        if (dd.rows > 0 && !dd.cols[(size_t)from[c]].values) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_shared: the donor has not been fed");
    }
    for (size_t c = 0; c < lf.schema.size(); ++c)
        if (from[c] >= 0) ld.cols[c] = dd.cols[(size_t)from[c]];
    ld.rows = dd.rows;
    ld.borrowed = true;
    return FLOCKGPU_OK;   // same stream as the donor's copies: ordered behind them without a wait
}

int flockgpu_plan_execute_retain(flockgpu_plan *plan, int64_t *rows) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    flockgpu_ctx *ctx = plan->ctx;
    FG_HIP(ctx, hipSetDevice(ctx->device));
    plan->has_retained = false;
    Exec ex{plan, ctx};
    Table t;
    // a root hash repartition is not computed: which partition a key lands in is unobservable once every partition is handed to
    // the same consumer, and that is the only hand-over this call serves
    FG_TRY(ex.exec(plan->ir.root.get(), &t));
    const Node *root = plan->ir.root.get();
    if (t.cols.size() != root->schema.size()) return fail(ctx, FLOCKGPU_ERR_HIP, "plan execute: %zu result columns for a schema of %zu", t.cols.size(), root->schema.size());
    for (size_t c = 0; c < t.cols.size(); ++c)
        if (!t.cols[c].present) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan execute: output column '%s' was not materialised", root->schema[c].name.c_str());
    // A result column must outlive the executes of OTHER plans on this context.  Buffers named after this plan (its leaves, its
    // generic operators' outputs) do; a fused pipeline that hands out context-wide buffers ("q5.out_auction", ...) would have them
    // overwritten by the same pipeline inside another plan -- q5's two stage plans both run the fused Partial COUNT (today its
    // columns reach the plan's result through plan-named buffers; this keeps it that way for whatever is fused next).  Such a column
    // is copied into a buffer of this plan (device to device, stream-ordered).
    char prefix[40];
    snprintf(prefix, sizeof prefix, "plan%p.", (const void *)plan);
    auto mine = [&](const void *ptr) {
        if (!ptr) return true;
        const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
        for (auto it = ctx->arena.lower_bound(prefix); it != ctx->arena.end() && it->first.compare(0, strlen(prefix), prefix) == 0; ++it) {
            const uintptr_t b = reinterpret_cast<uintptr_t>(it->second.ptr);
            if (it->second.ptr && a >= b && a < b + it->second.cap) return true;
        }
        return false;
    };
    for (size_t c = 0; c < t.cols.size(); ++c) {
        DevColumn &col = t.cols[c].c;
        if (t.rows == 0 || col.all_null) continue;
        if (col.valid && !mine(col.valid)) {
            uint8_t *v = nullptr;
            FG_TRY(arena_get_t(ctx, node_key(plan, root, "keepn", (int)c).c_str(), (size_t)t.rows + 16, &v));
            FG_HIP(ctx, hipMemcpyAsync(v, col.valid, (size_t)t.rows, hipMemcpyDeviceToDevice, ctx->stream));
            col.valid = v;
        }
        if (col.type == ColType::UTF8) {
            if (mine(col.values) && mine(col.offsets)) continue;
            void *bytes = nullptr;
            int32_t *offs = nullptr;
            FG_TRY(arena_get(ctx, node_key(plan, root, "keepb", (int)c).c_str(), (size_t)col.bytes + 16, &bytes));
            FG_TRY(arena_get_t(ctx, node_key(plan, root, "keepo", (int)c).c_str(), (size_t)t.rows + 4, &offs));
            if (col.bytes) FG_HIP(ctx, hipMemcpyAsync(bytes, col.values, (size_t)col.bytes, hipMemcpyDeviceToDevice, ctx->stream));
            FG_HIP(ctx, hipMemcpyAsync(offs, col.offsets, sizeof(int32_t) * ((size_t)t.rows + 1), hipMemcpyDeviceToDevice, ctx->stream));
            col.values = bytes;
            col.offsets = offs;
        } else {
            if (mine(col.values)) continue;
            void *vals = nullptr;
            FG_TRY(arena_get(ctx, node_key(plan, root, "keepv", (int)c).c_str(), (size_t)t.rows * col_width(col.type) + 16, &vals));
            FG_HIP(ctx, hipMemcpyAsync(vals, col.values, (size_t)t.rows * col_width(col.type), hipMemcpyDeviceToDevice, ctx->stream));
            col.values = vals;
        }
    }
    plan->retained = t;
    plan->has_retained = true;
    if (rows) *rows = t.rows;
    return FLOCKGPU_OK;   // no host wait: the consumer's kernels follow on the same stream
}

int flockgpu_plan_feed_from(flockgpu_plan *plan, int input, const flockgpu_plan *producer) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    flockgpu_ctx *ctx = plan->ctx;
    if (!producer || producer == plan || input < 0 || input >= (int)plan->ir.leaves.size()) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_from: bad argument");
    if (producer->ctx != ctx) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_from: the two plans live on different contexts (streams)");
    if (plan->ring_ppw) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_from: a plan with an open pane ring feeds panes (flockgpu_plan_feed_pane)");
    if (!producer->has_retained) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_from: the producer holds no retained result (flockgpu_plan_execute_retain)");
    const Leaf &lf = plan->ir.leaves[(size_t)input];
    LeafData &ld = plan->leaves[(size_t)input];
    if (ld.rows != 0 || ld.borrowed) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed_from: input %d already holds rows", input);
    const std::vector<Field> &ps = producer->ir.root->schema;
    const Table &t = producer->retained;
    std::vector<int> from(lf.schema.size(), -1);
    bool empty = false;   // a NULL the plan may drop (MAX over no rows): the relation is empty
    for (size_t c = 0; c < lf.schema.size(); ++c) {
        if (!lf.needed[c]) continue;
        for (size_t d = 0; d < ps.size(); ++d)
            if (ps[d].name == lf.schema[c].name && ps[d].type == lf.schema[c].type) from[c] = (int)d;
        if (from[c] < 0) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed_from: the producer's result has no column '%s' of type %s", lf.schema[c].name.c_str(), type_name(lf.schema[c]));
        if (t.cols[(size_t)from[c]].c.all_null) {
            if (!lf.null_droppable[c]) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed_from: column '%s' is NULL and would reach the output", lf.schema[c].name.c_str());
            empty = true;
        }
    }
    if (empty || t.rows == 0) return FLOCKGPU_OK;   // an unfed leaf is an empty relation
    for (size_t c = 0; c < lf.schema.size(); ++c) {
        if (from[c] < 0) continue;
        const DevColumn &col = t.cols[(size_t)from[c]].c;
        ld.cols[c] = DevBuf{const_cast<void *>(col.values), const_cast<int32_t *>(col.offsets), col.bytes, const_cast<uint8_t *>(col.valid), col.valid ? t.rows : 0};
    }
    ld.rows = t.rows;
    ld.borrowed = true;
    return FLOCKGPU_OK;
}

int flockgpu_plan_execute(flockgpu_plan *plan, struct ArrowSchema *out_schema, struct ArrowArray *out_batch) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    if (!out_schema || !out_batch) return fail(plan->ctx, FLOCKGPU_ERR_INVALID, "plan_execute: null output");
    if (plan->async_pending) return fail(plan->ctx, FLOCKGPU_ERR_INVALID, "plan_execute: an asynchronous execute is in flight (flockgpu_plan_wait first)");
    int n = 0;
    API_CLOCK("execute");
    return run_plan(plan, false, out_schema, out_batch, 1, &n);
}

int flockgpu_plan_execute_partitioned(flockgpu_plan *plan, struct ArrowSchema *out_schema, struct ArrowArray *out_batches, int capacity,
                                      int *n_partitions) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    if (!out_schema || !out_batches || !n_partitions || capacity < 1) return fail(plan->ctx, FLOCKGPU_ERR_INVALID, "plan_execute_partitioned: bad argument");
    if (plan->async_pending) return fail(plan->ctx, FLOCKGPU_ERR_INVALID, "plan_execute_partitioned: an asynchronous execute is in flight (flockgpu_plan_wait first)");
    return run_plan(plan, true, out_schema, out_batches, capacity, n_partitions);
}

// execute on the ctx's worker thread (the reference: one tokio task per plan, context.rs:172-191); the outputs wait in the plan
int flockgpu_plan_execute_async(flockgpu_plan *plan, int partitioned) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    flockgpu_ctx *ctx = plan->ctx;
    if (plan->async_pending) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_execute_async: a call is already in flight on this plan");
    const int cap = partitioned && plan->ir.root->kind == NKind::Repartition ? plan->ir.root->n_parts : 1;
    plan->async_batches.assign((size_t)cap, ArrowArray{});
    plan->async_schema = ArrowSchema{};
    plan->async_n = 0;
    FG_TRY(ctx_submit(ctx, [plan, partitioned, cap] {
        return run_plan(plan, partitioned != 0, &plan->async_schema, plan->async_batches.data(), cap, &plan->async_n);
    }));
    plan->async_pending = true;
    ctx->plan_in_flight = plan;
    return FLOCKGPU_OK;
}

int flockgpu_plan_wait(flockgpu_plan *plan, struct ArrowSchema *out_schema, struct ArrowArray *out_batches, int capacity, int *n_partitions) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    flockgpu_ctx *ctx = plan->ctx;
    if (!plan->async_pending) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_wait: no asynchronous execute was started on this plan");
    const int rc = ctx_wait(ctx);
    plan->async_pending = false;
    ctx->plan_in_flight = nullptr;
    auto drop = [&] {
        for (int i = 0; i < plan->async_n; ++i)
            if (plan->async_batches[(size_t)i].release) plan->async_batches[(size_t)i].release(&plan->async_batches[(size_t)i]);
        if (plan->async_schema.release) plan->async_schema.release(&plan->async_schema);
    };
    if (rc != FLOCKGPU_OK) {   // (a failed run_plan exports nothing; whatever it did export is released, never leaked)
        drop();
        return rc;
    }
    if (!out_schema || !out_batches || capacity < plan->async_n) {
        drop();
        return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_wait: %d output batches, room for %d", plan->async_n, capacity);
    }
    // (Arrow C Data Interface: a struct is moved by copying it bitwise and marking the source released)
    *out_schema = plan->async_schema;
    plan->async_schema.release = nullptr;
    for (int i = 0; i < plan->async_n; ++i) {
        out_batches[i] = plan->async_batches[(size_t)i];
        plan->async_batches[(size_t)i].release = nullptr;
    }
    if (n_partitions) *n_partitions = plan->async_n;
    return FLOCKGPU_OK;
}

const char *flockgpu_plan_partition_scheme(void) { return kPartitionScheme; }
int flockgpu_plan_check_partition_scheme(const char *scheme) {
    return scheme && !strcmp(scheme, kPartitionScheme) ? FLOCKGPU_OK : FLOCKGPU_ERR_UNSUPPORTED;
}

int flockgpu_host_alloc(size_t bytes, void **out) {
    if (!out) return FLOCKGPU_ERR_INVALID;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 64, hipHostMallocDefault) == hipSuccess ? FLOCKGPU_OK : FLOCKGPU_ERR_OOM;
}
int flockgpu_host_free(void *ptr) { return !ptr || hipHostFree(ptr) == hipSuccess ? FLOCKGPU_OK : FLOCKGPU_ERR_HIP; }
int flockgpu_host_register(void *ptr, size_t bytes) {
    if (!ptr || !bytes) return FLOCKGPU_ERR_INVALID;
    return hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess ? FLOCKGPU_OK : FLOCKGPU_ERR_HIP;
}
int flockgpu_host_unregister(void *ptr) { return ptr && hipHostUnregister(ptr) == hipSuccess ? FLOCKGPU_OK : FLOCKGPU_ERR_HIP; }

}  // extern "C"
