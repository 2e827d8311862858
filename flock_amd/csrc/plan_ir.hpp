// Physical-plan IR of the plan-level C ABI (include/flockgpu_plan.h): the serde_json text of the reference's
// `Arc<dyn ExecutionPlan>` (flock/src/runtime/context.rs:477-480; dialect: SURVEY.md appendix C, fixtures
// flock/src/tests/data/plan/*.json) parsed into a small operator tree with derived schemas.  Host-side only.
//
// Kept nodes: memory_exec (Scan), filter_exec, projection_exec, hash_aggregate_exec, hash_join_exec,
// repartition_exec with Hash partitioning, sort_exec (ORDER BY over columns: the reference's own boundary goldens end in it,
// flock/src/runtime/context.rs:471,549; the splitter cuts stages at it, distributed_plan/stage.rs:337) and global_limit_exec /
// local_limit_exec.  coalesce_batches_exec, repartition_exec RoundRobinBatch, merge_exec / coalesce_partitions_exec change
// neither the row multiset nor the schema and are dropped.  Anything else (window functions, outer
// joins, unknown expressions / types) makes the plan UNSUPPORTED: the host keeps its own engine for it.
#pragma once
#include <algorithm>
#include <cctype>
#include <set>
#include <sstream>

#include "plan_json.hpp"
#include "relops.hpp"

namespace flockgpu {
namespace ir {

struct Field {
    std::string name;
    ColType type = ColType::I32;
    bool is_ts = false;
    bool nullable = false;
};

// Expression dialect (SURVEY.md appendix C + the unary / list expressions of the fork's expressions/*.rs, upstream DataFusion ~6:
// IsNullExpr{arg}, IsNotNullExpr{arg}, NotExpr{arg}, NegativeExpr{arg}, InListExpr{expr, list, negated}).
enum class EKind { Col, LitI, LitF, LitS, LitB, LitNull, Bin, Cast, Not, IsNull, IsNotNull, Neg, InList, Case };
struct Expr {
    EKind kind = EKind::Col;
    int col = -1;  // Col: index into the input schema
    int64_t i = 0;   // LitI value / LitB 0 | 1
    double f = 0;
    std::string s;   // LitS value / Bin operator (Rust enum ident: Eq, NotEq, Lt, LtEq, Gt, GtEq, And, Or, Modulo, Multiply)
    ColType cast_to = ColType::I64;
    std::unique_ptr<Expr> l, r;  // Bin operands; the operand of Cast / Not / IsNull / IsNotNull / Neg / InList in l
    std::vector<std::unique_ptr<Expr>> list;   // InList: the literals; Case: WHEN, THEN, WHEN, THEN, ... (base expression in l, ELSE in r; either may be null)
    bool negated = false;                      // InList: NOT IN
    bool big_unsigned = false;                 // LitI: `i` is the bit pattern of a UInt64 above INT64_MAX
    bool cast_ts = false;                      // Cast: the target is Timestamp(Millisecond) (Int64 storage)
    bool try_cast = false;                     // Cast: try_cast_expr (a value that does not fit becomes NULL instead of failing the call)
    std::string lit_kind;                      // literals: the ScalarValue variant ("Int32", "Float64", ...; empty: a bare JSON value)
};

// Static type of an expression over `schema`: 0..3 = ColType I32 / I64 / U64 / F64, 4 = Utf8, 5 = Boolean, -1 = an untyped literal (it takes
// the type of whatever it meets), -2 = no consistent type.  (valprog.hpp: both operands of a binary operator have one type.)
inline int expr_static_type(const Expr *e, const std::vector<Field> &schema) {
    auto arith = [](const std::string &op) { return op == "Plus" || op == "Minus" || op == "Multiply" || op == "Divide" || op == "Modulo"; };
    switch (e->kind) {
        case EKind::Col: return e->col >= 0 && (size_t)e->col < schema.size() ? (int)schema[(size_t)e->col].type : -2;
        case EKind::LitI: return e->lit_kind == "Int32" ? 0 : e->lit_kind == "Int64" ? 1 : e->lit_kind == "UInt64" ? 2 : -1;
        case EKind::LitF: return 3;
        case EKind::LitS: return 4;
        case EKind::LitB: return 5;
        case EKind::LitNull: return -1;
        case EKind::Cast: return (int)e->cast_to;
        case EKind::Neg: return expr_static_type(e->l.get(), schema);
        case EKind::Not: case EKind::IsNull: case EKind::IsNotNull: case EKind::InList: return 5;
        case EKind::Bin: {
            if (!arith(e->s)) return 5;
            const int a = expr_static_type(e->l.get(), schema), b = expr_static_type(e->r.get(), schema);
            if (a == -2 || b == -2 || a >= 4 || b >= 4) return -2;
            if (a == -1) return b;
            if (b == -1 || a == b) return a;
            // (fixtures of older fork revisions write q1's conversion as `Float64 literal * Int32 column` without the cast the planner inserts)
            if (e->s == "Multiply" && (e->l->kind == EKind::LitF || e->r->kind == EKind::LitF)) return 3;
            return -2;
        }
        case EKind::Case: {
            int t = -1;
            for (size_t i = 1; i < e->list.size() && t == -1; i += 2) t = expr_static_type(e->list[i].get(), schema);
            if (t == -1 && e->r) t = expr_static_type(e->r.get(), schema);
            return t;
        }
    }
    return -2;
}

enum class NKind { Scan, Filter, Project, Aggregate, Join, Repartition, Sort, Limit, Window };
struct SortCol {
    int col = -1;            // input column
    bool descending = false;
    bool nulls_first = false;  // parsed and carried; device columns hold no NULLs (a NULL that could reach a sort is refused at feed)
};
struct Agg {
    std::string fn;  // "count" | "max" | "min" | "sum" | "avg"
    int arg = -1;    // Partial: input column of the argument (-1: a literal, COUNT(UInt8(1))); Final: the first state column
    int arg2 = -1;   // Final AVG: its second state column (the sum; `arg` is the count)
    std::string name;
    ColType type = ColType::U64;  // type of the finished aggregate
};
// State columns a Partial stage emits per aggregate, as DataFusion ~6 lays them out (Accumulator::state / state_fields,
// SURVEY.md appendix D): COUNT -> [count UInt64]; MAX / MIN / SUM -> [value]; AVG -> [count UInt64, sum Float64].
inline int agg_state_cols(const std::string &fn) { return fn == "avg" ? 2 : 1; }
struct Node {
    NKind kind = NKind::Scan;
    int id = 0;
    std::vector<std::unique_ptr<Node>> in;
    std::vector<Field> schema;
    int leaf = -1;                  // Scan: index into Plan::leaves
    std::unique_ptr<Expr> pred;     // Filter
    std::vector<std::pair<std::unique_ptr<Expr>, std::string>> proj;  // Project
    std::string mode;               // Aggregate: Partial | Final | FinalPartitioned
    std::vector<int> group;         // Aggregate: input columns of the group keys
    std::vector<Agg> aggs;
    int on_l = -1, on_r = -1;       // Join: key columns (left input, right input)
    int on_l2 = -1, on_r2 = -1;     // Join: second key pair (q9: auction = id AND price = final), -1 when there is one
    bool join_partitioned = false;  // Join: mode=Partitioned (both inputs arrive hash-partitioned on the keys)
    std::vector<int> hash_cols;     // Repartition
    int n_parts = 0;
    bool hash_diff = false;         // Repartition: HashDiff -- one partition per DISTINCT key (n_parts = what the host counted)
    std::vector<SortCol> sort_cols; // Sort: ORDER BY keys, most significant first
    int64_t limit = -1;             // Limit: rows kept
    std::vector<std::vector<int>> win_part;   // Window: per ROW_NUMBER() column (they come FIRST in the schema), the input columns of its PARTITION BY
    std::vector<char> required;     // per output column: needed by an ancestor (or by the plan output)
};

struct Leaf {
    std::vector<Field> schema;  // the columns the MemoryExec scans (after its projection)
    std::string relation;       // bid | auction | person | side_input | "" (guessed from the column names)
    std::vector<char> needed;   // per column: read by the plan (others are never uploaded)
    // per column: a row whose value here is NULL can be dropped at the scan without changing the plan's result (the column
    // only feeds inner-join keys, MAX arguments or comparisons) -- how the NULL `maxn` of an empty partition is ingested
    std::vector<char> null_droppable;
    // the leaf feeds a HashJoinExec mode=Partitioned / a FinalPartitioned aggregate without a hash repartition of this plan in
    // between: its batches were placed by the PRODUCING stage's hash, and every such leaf of the plan must have been placed by
    // the same one (flockgpu_plan.h "hash placement")
    bool co_partitioned = false;
};

struct Plan {
    std::unique_ptr<Node> root;
    std::vector<Leaf> leaves;
    int n_nodes = 0;
    std::string why;  // reason when unsupported
};

// ---------------------------------------------------------------------------------------------------------------------
inline const std::string &tag(const JValue *n) {
    static const std::string empty;
    const JValue *t = n ? n->get("execution_plan") : nullptr;
    return t && t->kind == JValue::Str ? t->str : empty;
}
inline const std::string &etag(const JValue *e) {
    static const std::string empty;
    const JValue *t = e ? e->get("physical_expr") : nullptr;
    return t && t->kind == JValue::Str ? t->str : empty;
}

inline bool parse_type(const JValue *dt, ColType *t, bool *is_ts) {
    *is_ts = false;
    if (!dt) return false;
    if (dt->kind == JValue::Str) {
        if (dt->str == "Int32") { *t = ColType::I32; return true; }
        if (dt->str == "Int64") { *t = ColType::I64; return true; }
        if (dt->str == "UInt64") { *t = ColType::U64; return true; }
        if (dt->str == "Float64") { *t = ColType::F64; return true; }
        if (dt->str == "Utf8") { *t = ColType::UTF8; return true; }
        return false;
    }
    if (dt->kind == JValue::Obj && dt->obj.size() == 1 && dt->obj[0].first == "Timestamp") {
        const JValue *a = dt->obj[0].second.get();
        if (a->kind == JValue::Arr && !a->arr.empty() && a->arr[0]->kind == JValue::Str && a->arr[0]->str == "Millisecond") {
            *t = ColType::I64;
            *is_ts = true;
            return true;
        }
    }
    return false;
}

inline const char *type_name(const Field &f) {
    if (f.is_ts) return "Timestamp(ms)";
    switch (f.type) {
        case ColType::I32: return "Int32";
        case ColType::I64: return "Int64";
        case ColType::U64: return "UInt64";
        case ColType::F64: return "Float64";
        default: return "Utf8";
    }
}

struct Builder {
    Plan *plan;
    std::string err;
    bool fail(const std::string &m) {
        if (err.empty()) err = m;
        return false;
    }

    bool fields_of(const JValue *schema, std::vector<Field> *out) {
        const JValue *fields = schema ? schema->get("fields") : nullptr;
        if (!fields || fields->kind != JValue::Arr) return fail("node without schema.fields");
        for (auto &f : fields->arr) {
            Field fd;
            fd.name = f->s("name");
            const JValue *nl = f->get("nullable");
            fd.nullable = nl && nl->kind == JValue::Bool && nl->b;
            if (!parse_type(f->get("data_type"), &fd.type, &fd.is_ts)) return fail("column '" + fd.name + "': data type outside {Int32, Int64, UInt64, Float64, Utf8, Timestamp(ms)}");
            out->push_back(fd);
        }
        return true;
    }

    // column{name[,index]} against `schema`: by index when it names the same column, else by name (older fork
    // revisions serialise the name only)
    int resolve(const JValue *e, const std::vector<Field> &schema) {
        const std::string name = e->s("name");
        const JValue *ix = e->get("index");
        if (ix && ix->kind == JValue::Num && ix->is_int && ix->inum >= 0 && (size_t)ix->inum < schema.size() &&
            (name.empty() || schema[(size_t)ix->inum].name == name))
            return (int)ix->inum;
        for (size_t i = 0; i < schema.size(); ++i)
            if (schema[i].name == name) return (int)i;
        // qualified name ("bid.price", "CountBids.num") against an unqualified schema
        const size_t dot = name.rfind('.');
        if (dot != std::string::npos)
            for (size_t i = 0; i < schema.size(); ++i)
                if (schema[i].name == name.substr(dot + 1)) return (int)i;
        return -1;
    }

    std::unique_ptr<Expr> expr(const JValue *e, const std::vector<Field> &schema) {
        std::unique_ptr<Expr> x(new Expr());
        const std::string &t = etag(e);
        if (t == "column") {
            x->kind = EKind::Col;
            x->col = resolve(e, schema);
            if (x->col < 0) { fail("column '" + e->s("name") + "' not in the input schema"); return nullptr; }
            return x;
        }
        if (t == "literal") {
            const JValue *val = e->get("value");
            std::string kind;
            if (val && val->kind == JValue::Obj && val->obj.size() == 1) {
                kind = val->obj[0].first;
                val = val->obj[0].second.get();
            }
            if (!val) { fail("literal without value"); return nullptr; }
            x->lit_kind = kind;
            if (val->kind == JValue::Null) { x->kind = EKind::LitNull; return x; }   // ScalarValue::Int32(None) and its siblings
            if (val->kind == JValue::Bool) { x->kind = EKind::LitB; x->i = val->b ? 1 : 0; return x; }
            if (val->kind == JValue::Str) { x->kind = EKind::LitS; x->s = val->str; return x; }
            if (val->kind == JValue::Num) {
                if (val->is_int && kind.find("Float") == std::string::npos) { x->kind = EKind::LitI; x->i = val->inum; x->big_unsigned = val->is_big_unsigned; }
                else { x->kind = EKind::LitF; x->f = val->num; }
                return x;
            }
            fail("literal of an unsupported kind");
            return nullptr;
        }
        if (t == "cast_expr" || t == "try_cast_expr") {
            x->kind = EKind::Cast;
            x->try_cast = t == "try_cast_expr";
            bool ts = false;
            if (!parse_type(e->get("cast_type"), &x->cast_to, &ts)) { fail("cast to an unsupported type"); return nullptr; }
            x->cast_ts = ts;
            x->l = expr(e->get("expr"), schema);
            return x->l ? std::move(x) : nullptr;
        }
        if (t == "binary_expr") {
            x->kind = EKind::Bin;
            x->s = e->s("op");
            x->l = expr(e->get("left"), schema);
            x->r = expr(e->get("right"), schema);
            return x->l && x->r ? std::move(x) : nullptr;
        }
        if (t == "not_expr" || t == "is_null_expr" || t == "is_not_null_expr" || t == "negative_expr") {
            x->kind = t == "not_expr" ? EKind::Not : t == "is_null_expr" ? EKind::IsNull : t == "is_not_null_expr" ? EKind::IsNotNull : EKind::Neg;
            const JValue *arg = e->get("arg");
            x->l = expr(arg ? arg : e->get("expr"), schema);
            return x->l ? std::move(x) : nullptr;
        }
        if (t == "in_list_expr") {
            x->kind = EKind::InList;
            x->l = expr(e->get("expr"), schema);
            const JValue *list = e->get("list"), *neg = e->get("negated");
            if (!x->l || !list || list->kind != JValue::Arr || list->arr.empty()) { if (x->l) fail("in_list_expr without a list"); return nullptr; }
            x->negated = neg && neg->kind == JValue::Bool && neg->b;
            for (auto &item : list->arr) {
                auto li = expr(item.get(), schema);
                if (!li) return nullptr;
                x->list.push_back(std::move(li));
            }
            return x;
        }
        if (t == "case_expr") {   // CaseExpr { expr: Option, when_then_expr: Vec<(when, then)>, else_expr: Option }
            x->kind = EKind::Case;
            const JValue *base = e->get("expr"), *wt = e->get("when_then_expr"), *el = e->get("else_expr");
            if (!wt || wt->kind != JValue::Arr || wt->arr.empty()) { fail("case_expr without when_then_expr"); return nullptr; }
            if (base && base->kind != JValue::Null && !(x->l = expr(base, schema))) return nullptr;
            if (el && el->kind != JValue::Null && !(x->r = expr(el, schema))) return nullptr;
            for (auto &pair : wt->arr) {
                if (pair->kind != JValue::Arr || pair->arr.size() != 2) { fail("malformed when_then_expr"); return nullptr; }
                auto w = expr(pair->arr[0].get(), schema), th = expr(pair->arr[1].get(), schema);
                if (!w || !th) return nullptr;
                x->list.push_back(std::move(w));
                x->list.push_back(std::move(th));
            }
            return x;
        }
        fail("physical_expr '" + t + "' is not supported");
        return nullptr;
    }

    // An EXPRESSION where an operator reads a column (GROUP BY a % 10, SUM(price * 2), ORDER BY a + b): `in` gets a projection on top (once:
    // `wrapped`) that carries every column through and the expression's value beside them (the general evaluator, valprog.hpp); returns the
    // new column's index in the wrapped schema, -1 when refused (`err` says why).
    int computed_column(std::unique_ptr<Node> &in, bool &wrapped, const JValue *e, const char *what) {
        if (!wrapped) {
            std::unique_ptr<Node> w(new Node());
            w->kind = NKind::Project;
            w->id = plan->n_nodes++;
            w->schema = in->schema;
            for (size_t i = 0; i < in->schema.size(); ++i) {
                std::unique_ptr<Expr> c(new Expr());
                c->kind = EKind::Col;
                c->col = (int)i;
                w->proj.emplace_back(std::move(c), in->schema[i].name);
            }
            w->in.push_back(std::move(in));
            in = std::move(w);
            wrapped = true;
        }
        const std::vector<Field> &below = in->in[0]->schema;
        auto x = expr(e, below);
        if (!x) return -1;
        const int ty = expr_static_type(x.get(), below);
        if (ty < 0 || ty > 3) { fail(std::string(what) + " over an expression without a numeric type"); return -1; }
        Field f;
        f.name = "#" + std::to_string(in->schema.size());
        f.type = (ColType)ty;
        f.nullable = true;
        f.is_ts = x->kind == EKind::Cast && x->cast_ts;
        in->proj.emplace_back(std::move(x), f.name);
        in->schema.push_back(f);
        return (int)in->schema.size() - 1;
    }

    static std::string guess_relation(const std::vector<Field> &f) {
        auto has = [&](const char *n) { return std::any_of(f.begin(), f.end(), [&](const Field &x) { return x.name == n; }); };
        if (has("auction") || has("bidder") || has("price")) return "bid";
        if (has("a_id") || has("seller") || has("category")) return "auction";
        if (has("p_id") || has("state") || has("city")) return "person";
        if (has("key") && has("value")) return "side_input";
        if (has("ad_id") || has("event_type")) return "ad_event";
        if (has("c_ad_id") || has("campaign_id")) return "campaign";
        return "";
    }

    std::unique_ptr<Node> node(const JValue *j, int depth = 0) {
        if (!j || j->kind != JValue::Obj || depth > 64) { fail("malformed plan node"); return nullptr; }
        const std::string &t = tag(j);
        // transparent nodes
        if (t == "coalesce_batches_exec" || t == "merge_exec" || t == "coalesce_partitions_exec") return node(j->get("input"), depth + 1);
        std::unique_ptr<Node> n(new Node());
        if (t == "repartition_exec") {
            const JValue *part = j->get("partitioning");
            const JValue *hash = part ? part->get("Hash") : nullptr;
            // HashDiff(exprs, n): the fork's own variant (flock-function/src/aws/window/session.rs:252, global.rs:234; datasource/nexmark/queries/
            // q6.rs:128, q11.rs:168, q12.rs:136): n = COUNT(DISTINCT key), "each partition has a unique key after repartition execution"
            if (!hash && part && part->get("HashDiff")) {
                hash = part->get("HashDiff");
                n->hash_diff = true;
            }
            if (!hash) return node(j->get("input"), depth + 1);  // RoundRobinBatch(n)
            if (hash->kind != JValue::Arr || hash->arr.size() != 2 || hash->arr[0]->kind != JValue::Arr || hash->arr[1]->kind != JValue::Num) {
                fail("malformed Hash partitioning");
                return nullptr;
            }
            n->kind = NKind::Repartition;
            auto in = node(j->get("input"), depth + 1);
            if (!in) return nullptr;
            n->schema = in->schema;
            for (auto &e : hash->arr[0]->arr) {
                if (etag(e.get()) != "column") { fail("Hash partitioning on a computed expression"); return nullptr; }
                const int c = resolve(e.get(), n->schema);
                if (c < 0) { fail("Hash partitioning column not in the schema"); return nullptr; }
                n->hash_cols.push_back(c);
            }
            n->n_parts = (int)hash->arr[1]->inum;
            if (n->hash_cols.empty() || n->n_parts < 1) { fail("Hash partitioning without columns / partitions"); return nullptr; }
            n->in.push_back(std::move(in));
        } else if (t == "memory_exec") {
            n->kind = NKind::Scan;
            std::vector<Field> all;
            if (!fields_of(j->get("schema"), &all)) return nullptr;
            const JValue *proj = j->get("projection");
            if (proj && proj->kind == JValue::Arr && !proj->arr.empty()) {
                bool ok = true;
                for (auto &i : proj->arr) {
                    if (i->kind == JValue::Num && i->inum >= 0 && (size_t)i->inum < all.size()) n->schema.push_back(all[(size_t)i->inum]);
                    else ok = false;
                }
                // fixtures of older fork revisions list only the projected fields: indices then exceed the list
                if (!ok) n->schema = all;
            } else {
                n->schema = all;
            }
            Leaf lf;
            lf.schema = n->schema;
            lf.relation = guess_relation(lf.schema);
            n->leaf = (int)plan->leaves.size();
            plan->leaves.push_back(lf);
        } else if (t == "filter_exec") {
            n->kind = NKind::Filter;
            auto in = node(j->get("input"), depth + 1);
            if (!in) return nullptr;
            n->schema = in->schema;
            n->pred = expr(j->get("predicate"), n->schema);
            if (!n->pred) return nullptr;
            n->in.push_back(std::move(in));
        } else if (t == "projection_exec") {
            n->kind = NKind::Project;
            auto in = node(j->get("input"), depth + 1);
            if (!in) return nullptr;
            const JValue *ex = j->get("expr");
            if (!ex || ex->kind != JValue::Arr) { fail("projection_exec without expr"); return nullptr; }
            for (auto &pair : ex->arr) {
                if (pair->kind != JValue::Arr || pair->arr.size() < 2 || pair->arr[1]->kind != JValue::Str) { fail("malformed projection expr"); return nullptr; }
                auto e = expr(pair->arr[0].get(), in->schema);
                if (!e) return nullptr;
                Field f;
                f.name = pair->arr[1]->str;
                if (e->kind == EKind::Col) {
                    const Field &src = in->schema[(size_t)e->col];
                    f.type = src.type; f.is_ts = src.is_ts; f.nullable = src.nullable;
                } else {   // computed: q1's `literal * column` kernel, or the general evaluator (valprog.hpp) -- a numeric result either way
                    const int ty = expr_static_type(e.get(), in->schema);
                    if (ty < 0 || ty > 3) { fail(ty == 5 ? "projection of a Boolean expression (no Boolean columns at this boundary)" : "projection expression without a numeric type"); return nullptr; }
                    f.type = (ColType)ty;
                    f.nullable = true;
                    f.is_ts = e->kind == EKind::Cast && e->cast_ts;   // CAST(x AS Timestamp(Millisecond))
                }
                n->proj.emplace_back(std::move(e), f.name);
                n->schema.push_back(f);
            }
            n->in.push_back(std::move(in));
        } else if (t == "hash_aggregate_exec") {
            n->kind = NKind::Aggregate;
            n->mode = j->s("mode");
            if (n->mode != "Partial" && n->mode != "Final" && n->mode != "FinalPartitioned") { fail("aggregate mode '" + n->mode + "'"); return nullptr; }
            auto in = node(j->get("input"), depth + 1);
            if (!in) return nullptr;
            const bool is_final = n->mode != "Partial";
            // A group key or an aggregate argument that is an EXPRESSION (GROUP BY a % 10, SUM(price * 2)): the stage that evaluates it
            // (Partial) gets a projection underneath that carries every input column through and the expression's value beside them
            // (the general evaluator, valprog.hpp); the aggregate then reads a column, as ever.  -1: refused (plan->why says why).
            bool wrapped = false;
            auto computed = [&](const JValue *e, const char *what) -> int { return computed_column(in, wrapped, e, what); };
            const JValue *ge = j->get("group_expr");
            size_t gi = 0;
            if (ge && ge->kind == JValue::Arr)
                for (auto &pair : ge->arr) {
                    if (pair->kind != JValue::Arr || pair->arr.size() < 2) { fail("malformed group_expr"); return nullptr; }
                    // the final stage reads the group keys by POSITION from the partial stage's output
                    // (DataFusion's merge expressions); the partial / single stage evaluates the expression
                    int c = -1;
                    if (is_final && gi < in->schema.size()) c = (int)gi;
                    else if (etag(pair->arr[0].get()) == "column") c = resolve(pair->arr[0].get(), in->schema);
                    else if (!is_final) c = computed(pair->arr[0].get(), "GROUP BY");
                    if (c < 0) { fail("GROUP BY on something that is neither an input column nor a numeric expression"); return nullptr; }
                    n->group.push_back(c);
                    Field f = in->schema[(size_t)c];
                    f.name = pair->arr[1]->kind == JValue::Str ? pair->arr[1]->str : f.name;
                    n->schema.push_back(f);
                    ++gi;
                }
            const JValue *ae = j->get("aggr_expr");
            int state_at = (int)n->group.size();  // Final: position of the next aggregate's first state column
            if (ae && ae->kind == JValue::Arr)
                for (auto &x : ae->arr) {
                    Agg a;
                    a.fn = x->s("aggregate_expr");
                    a.name = x->s("name");
                    bool ts = false;
                    if (!parse_type(x->get("data_type"), &a.type, &ts)) { fail("aggregate '" + a.name + "' of an unsupported type"); return nullptr; }
                    if (a.fn != "count" && a.fn != "max" && a.fn != "min" && a.fn != "sum" && a.fn != "avg") {
                        fail("aggregate function '" + a.fn + "' (supported: count, max, min, sum, avg)");
                        return nullptr;
                    }
                    if (is_final) {
                        a.arg = state_at;  // state columns, by position
                        if (a.fn == "avg") a.arg2 = state_at + 1;
                        state_at += agg_state_cols(a.fn);
                        if ((size_t)state_at > in->schema.size()) { fail("final aggregate without its state column"); return nullptr; }
                    } else {
                        const JValue *arg = x->get("expr");
                        if (arg && etag(arg) == "column") {
                            a.arg = resolve(arg, in->schema);
                            if (a.arg < 0) { fail("aggregate argument not in the input schema"); return nullptr; }
                        } else if (arg && etag(arg) != "literal") {   // SUM(price * 2), COUNT(CASE ...): the expression becomes a column underneath
                            a.arg = computed(arg, ("aggregate '" + a.fn + "'").c_str());
                            if (a.arg < 0) return nullptr;
                        } else if (a.fn != "count") {
                            fail("aggregate '" + a.fn + "' over a literal");
                            return nullptr;
                        }
                    }
                    if (a.fn == "count") a.type = ColType::U64;
                    if (a.fn == "avg") a.type = ColType::F64;
                    Field f;
                    f.nullable = true;
                    f.is_ts = ts && (a.fn == "max" || a.fn == "min");   // MIN / MAX of a Timestamp column is a Timestamp (q11's start_time / end_time)
                    if (is_final) {
                        f.name = a.name;
                        f.type = a.type;
                        n->schema.push_back(f);
                    } else if (a.fn == "avg") {
                        f.name = a.name + "[count]";
                        f.type = ColType::U64;
                        n->schema.push_back(f);
                        f.name = a.name + "[sum]";
                        f.type = ColType::F64;
                        n->schema.push_back(f);
                    } else {
                        f.name = a.name + "[" + a.fn + "]";
                        f.type = a.type;
                        n->schema.push_back(f);
                    }
                    n->aggs.push_back(a);
                }
            n->in.push_back(std::move(in));
        } else if (t == "hash_join_exec") {
            n->kind = NKind::Join;
            if (j->s("join_type") != "Inner") { fail("only Inner joins"); return nullptr; }
            n->join_partitioned = j->s("mode") == "Partitioned";
            auto l = node(j->get("left"), depth + 1);
            if (!l) return nullptr;
            auto r = node(j->get("right"), depth + 1);
            if (!r) return nullptr;
            const JValue *on = j->get("on");
            if (!on || on->kind != JValue::Arr || on->arr.empty() || on->arr.size() > 2) {
                fail("join on other than one or two key pairs");
                return nullptr;
            }
            for (auto &pair : on->arr)
                if (pair->kind != JValue::Arr || pair->arr.size() != 2) { fail("malformed join key pair"); return nullptr; }
            auto keycol = [&](const JValue *k, const std::vector<Field> &schema) {
                if (k->kind == JValue::Str) {  // older fork revision: bare names
                    for (size_t i = 0; i < schema.size(); ++i)
                        if (schema[i].name == k->str) return (int)i;
                    return -1;
                }
                return resolve(k, schema);
            };
            n->on_l = keycol(on->arr[0]->arr[0].get(), l->schema);
            n->on_r = keycol(on->arr[0]->arr[1].get(), r->schema);
            if (n->on_l < 0 || n->on_r < 0) { fail("join key not in the input schemas"); return nullptr; }
            if (on->arr.size() == 2) {
                n->on_l2 = keycol(on->arr[1]->arr[0].get(), l->schema);
                n->on_r2 = keycol(on->arr[1]->arr[1].get(), r->schema);
                if (n->on_l2 < 0 || n->on_r2 < 0) { fail("join key not in the input schemas"); return nullptr; }
            }
            n->schema = l->schema;
            n->schema.insert(n->schema.end(), r->schema.begin(), r->schema.end());
            n->in.push_back(std::move(l));
            n->in.push_back(std::move(r));
        } else if (t == "sort_exec") {
            n->kind = NKind::Sort;
            auto in = node(j->get("input"), depth + 1);
            if (!in) return nullptr;
            const size_t n_in = in->schema.size();
            bool wrapped = false;
            const JValue *ex = j->get("expr");
            if (!ex || ex->kind != JValue::Arr || ex->arr.empty()) { fail("sort_exec without expr"); return nullptr; }
            for (auto &k : ex->arr) {
                const JValue *e = k->get("expr");
                if (!e) { fail("sort_exec key without expr"); return nullptr; }
                SortCol sc;
                // ORDER BY an expression: its value becomes a column underneath (computed_column) and is dropped again above the sort
                sc.col = etag(e) == "column" ? resolve(e, in->schema) : computed_column(in, wrapped, e, "ORDER BY");
                if (sc.col < 0) { if (err.empty()) fail("ORDER BY column '" + e->s("name") + "' not in the input schema"); return nullptr; }
                const JValue *opt = k->get("options");
                const JValue *d = opt ? opt->get("descending") : nullptr, *nf = opt ? opt->get("nulls_first") : nullptr;
                sc.descending = d && d->kind == JValue::Bool && d->b;
                sc.nulls_first = nf && nf->kind == JValue::Bool && nf->b;
                n->sort_cols.push_back(sc);
            }
            n->schema = in->schema;
            n->in.push_back(std::move(in));
            if (wrapped) {   // the sort's own columns only: a projection on top takes the computed keys out again
                n->id = plan->n_nodes++;
                std::unique_ptr<Node> top(new Node());
                top->kind = NKind::Project;
                for (size_t i = 0; i < n_in; ++i) {
                    std::unique_ptr<Expr> c(new Expr());
                    c->kind = EKind::Col;
                    c->col = (int)i;
                    top->proj.emplace_back(std::move(c), n->schema[i].name);
                    top->schema.push_back(n->schema[i]);
                }
                top->in.push_back(std::move(n));
                n = std::move(top);
            }
        } else if (t == "window_agg_exec") {
            // WindowAggExec (q6.sql: ROW_NUMBER() OVER (PARTITION BY a_id ORDER BY price DESC), benchmarks/src/nexmark/query/q6_plan.fmt:6,11).
            // The physical planner sorts the input by (PARTITION BY, ORDER BY) underneath (a sort_exec); the operator numbers the rows of every
            // RUN of equal partition keys 1, 2, ... in the order they arrive -- what DataFusion's partition points over a sorted batch give.
            // Output: the window columns first (UInt64), then the input's (q6_plan.fmt's schemas).  Only ROW_NUMBER is taken.
            n->kind = NKind::Window;
            auto in = node(j->get("input"), depth + 1);
            if (!in) return nullptr;
            const JValue *we = j->get("window_expr");
            if (!we || we->kind != JValue::Arr || we->arr.empty()) { fail("window_agg_exec without window_expr"); return nullptr; }
            std::vector<Field> wf;
            for (auto &w : we->arr) {
                if (w->kind != JValue::Obj) { fail("malformed window_expr"); return nullptr; }
                std::string fun;
                for (const char *key : {"fun", "function", "window_function", "built_in", "expr", "name"}) {
                    const JValue *f = w->get(key);
                    if (!f) continue;
                    std::string v = f->kind == JValue::Str ? f->str : (f->kind == JValue::Obj && !f->obj.empty() ? f->obj[0].first : std::string());
                    for (auto &ch : v) ch = (char)std::tolower((unsigned char)ch);
                    v.erase(std::remove(v.begin(), v.end(), '_'), v.end());
                    if (v.find("rownumber") != std::string::npos) { fun = "row_number"; break; }
                    if (fun.empty() && !v.empty()) fun = v;
                }
                if (fun != "row_number") { fail("window function '" + fun + "' (supported: ROW_NUMBER)"); return nullptr; }
                std::vector<int> part;
                const JValue *pb = w->get("partition_by");
                if (pb && pb->kind == JValue::Arr)
                    for (auto &e : pb->arr) {
                        if (etag(e.get()) != "column") { fail("PARTITION BY on something other than a column"); return nullptr; }
                        const int c = resolve(e.get(), in->schema);
                        if (c < 0) { fail("PARTITION BY column not in the input schema"); return nullptr; }
                        part.push_back(c);
                    }
                n->win_part.push_back(part);
                Field f;
                const JValue *nm = w->get("name");
                f.name = nm && nm->kind == JValue::Str ? nm->str : "ROW_NUMBER()";
                f.type = ColType::U64;
                f.nullable = true;
                wf.push_back(f);
            }
            n->schema = wf;
            n->schema.insert(n->schema.end(), in->schema.begin(), in->schema.end());
            n->in.push_back(std::move(in));
        } else if (t == "global_limit_exec" || t == "local_limit_exec") {
            n->kind = NKind::Limit;
            auto in = node(j->get("input"), depth + 1);
            if (!in) return nullptr;
            n->schema = in->schema;
            const JValue *l = j->get("limit");
            if (!l || l->kind != JValue::Num || !l->is_int || l->inum < 0) { fail("limit without a row count"); return nullptr; }
            n->limit = l->inum;
            n->in.push_back(std::move(in));
        } else {
            fail("execution_plan '" + t + "' is not supported");
            return nullptr;
        }
        n->id = plan->n_nodes++;
        return n;
    }
};

inline void expr_cols(const Expr *e, std::set<int> *out) {
    if (!e) return;
    if (e->kind == EKind::Col) out->insert(e->col);
    expr_cols(e->l.get(), out);
    expr_cols(e->r.get(), out);
    for (auto &li : e->list) expr_cols(li.get(), out);
}

// Marks, top-down, the output columns of every node that something above it reads; leaves learn which of their
// columns have to reach the device at all.
inline void mark_required(Plan *p, Node *n, const std::vector<char> &req) {
    n->required = req;
    auto need = [](std::vector<char> &v, int c) { if (c >= 0 && (size_t)c < v.size()) v[(size_t)c] = 1; };
    switch (n->kind) {
        case NKind::Scan: {
            Leaf &lf = p->leaves[(size_t)n->leaf];
            if (lf.needed.size() != n->schema.size()) lf.needed.assign(n->schema.size(), 0);
            for (size_t i = 0; i < req.size(); ++i) lf.needed[i] |= req[i];
            break;
        }
        case NKind::Filter: {
            std::vector<char> r = req;
            std::set<int> cs;
            expr_cols(n->pred.get(), &cs);
            for (int c : cs) need(r, c);
            mark_required(p, n->in[0].get(), r);
            break;
        }
        case NKind::Project: {
            std::vector<char> r(n->in[0]->schema.size(), 0);
            for (size_t i = 0; i < n->proj.size(); ++i)
                if (req[i]) {
                    std::set<int> cs;
                    expr_cols(n->proj[i].first.get(), &cs);
                    for (int c : cs) need(r, c);
                }
            mark_required(p, n->in[0].get(), r);
            break;
        }
        case NKind::Aggregate: {
            std::vector<char> r(n->in[0]->schema.size(), 0);
            for (int c : n->group) need(r, c);
            for (auto &a : n->aggs) { need(r, a.arg); need(r, a.arg2); }
            mark_required(p, n->in[0].get(), r);
            break;
        }
        case NKind::Join: {
            const size_t nl = n->in[0]->schema.size();
            std::vector<char> l(req.begin(), req.begin() + nl), r(req.begin() + nl, req.end());
            need(l, n->on_l);
            need(r, n->on_r);
            need(l, n->on_l2);
            need(r, n->on_r2);
            mark_required(p, n->in[0].get(), l);
            mark_required(p, n->in[1].get(), r);
            break;
        }
        case NKind::Repartition: {
            std::vector<char> r = req;
            for (int c : n->hash_cols) need(r, c);
            mark_required(p, n->in[0].get(), r);
            break;
        }
        case NKind::Sort: {
            std::vector<char> r = req;
            for (auto &k : n->sort_cols) need(r, k.col);
            mark_required(p, n->in[0].get(), r);
            break;
        }
        case NKind::Limit:
            mark_required(p, n->in[0].get(), req);
            break;
        case NKind::Window: {
            const size_t nw = n->win_part.size();
            std::vector<char> r(req.begin() + (long)nw, req.end());
            for (auto &part : n->win_part)
                for (int c : part) need(r, c);
            mark_required(p, n->in[0].get(), r);
            break;
        }
    }
}

// The columns whose NULL makes `e` NULL: reached through arithmetic, comparisons, casts and unary minus only.  A column under CASE, IS [NOT] NULL,
// IN, NOT, AND / OR does not count -- `CASE WHEN f <= f THEN 100 ELSE i END` has a value where f is NULL.
inline void strict_cols(const Expr *e, std::set<int> *out) {
    if (!e) return;
    switch (e->kind) {
        case EKind::Col: out->insert(e->col); return;
        case EKind::Cast: case EKind::Neg: strict_cols(e->l.get(), out); return;
        case EKind::Bin:
            if (e->s == "And" || e->s == "Or") return;
            strict_cols(e->l.get(), out);
            strict_cols(e->r.get(), out);
            return;
        default: return;
    }
}

// droppable[c]: dropping the rows of `n`'s output whose column c is NULL leaves the plan's result unchanged.
inline void mark_null_droppable(Plan *p, const Node *n, const std::vector<char> &droppable) {
    switch (n->kind) {
        case NKind::Scan: {
            Leaf &lf = p->leaves[(size_t)n->leaf];
            if (lf.null_droppable.size() != n->schema.size()) lf.null_droppable.assign(n->schema.size(), 1);
            for (size_t i = 0; i < droppable.size(); ++i) lf.null_droppable[i] &= droppable[i];
            break;
        }
        case NKind::Filter: {
            std::vector<char> d = droppable;
            // comparisons with NULL are NULL, and a NULL predicate drops the row -- through AND, not through OR
            std::vector<const Expr *> stack{n->pred.get()};
            while (!stack.empty()) {
                const Expr *e = stack.back();
                stack.pop_back();
                if (e->kind == EKind::Bin && e->s == "And") { stack.push_back(e->l.get()); stack.push_back(e->r.get()); continue; }
                if (e->kind == EKind::Bin && e->s != "Or") {
                    std::set<int> cs;
                    strict_cols(e, &cs);
                    for (int c : cs) d[(size_t)c] = 1;
                }
            }
            mark_null_droppable(p, n->in[0].get(), d);
            break;
        }
        case NKind::Project: {
            std::vector<char> d(n->in[0]->schema.size(), 0);
            for (size_t i = 0; i < n->proj.size(); ++i)
                if (droppable[i] && n->proj[i].first->kind == EKind::Col) d[(size_t)n->proj[i].first->col] = 1;
            mark_null_droppable(p, n->in[0].get(), d);
            break;
        }
        case NKind::Aggregate: {
            std::vector<char> d(n->in[0]->schema.size(), 0);
            if (n->group.empty())  // MAX ignores NULLs; over nothing but NULLs it is NULL, as over no rows
                for (auto &a : n->aggs)
                    if (a.fn == "max" && a.arg >= 0) d[(size_t)a.arg] = 1;
            mark_null_droppable(p, n->in[0].get(), d);
            break;
        }
        case NKind::Join: {
            const size_t nl = n->in[0]->schema.size();
            std::vector<char> l(droppable.begin(), droppable.begin() + nl), r(droppable.begin() + nl, droppable.end());
            l[(size_t)n->on_l] = 1;  // NULL keys never match in an inner join
            r[(size_t)n->on_r] = 1;
            if (n->on_l2 >= 0) { l[(size_t)n->on_l2] = 1; r[(size_t)n->on_r2] = 1; }
            mark_null_droppable(p, n->in[0].get(), l);
            mark_null_droppable(p, n->in[1].get(), r);
            break;
        }
        case NKind::Repartition:
        case NKind::Sort:   // (the order of the rows that remain does not depend on the rows that were dropped)
            mark_null_droppable(p, n->in[0].get(), droppable);
            break;
        case NKind::Limit:  // WHICH rows make the first n depends on every row below: nothing may be dropped early
        case NKind::Window: // ... and so does every row's number
            mark_null_droppable(p, n->in[0].get(), std::vector<char>(n->in[0]->schema.size(), 0));
            break;
    }
}

inline void mark_co_partitioned(Plan *p, const Node *n, bool under) {
    switch (n->kind) {
        case NKind::Scan:
            if (under) p->leaves[(size_t)n->leaf].co_partitioned = true;
            return;
        case NKind::Repartition:   // this plan places the rows itself from here on
            under = false;
            break;
        case NKind::Join:
            if (n->join_partitioned) under = true;
            break;
        case NKind::Aggregate:
            if (n->mode == "FinalPartitioned") under = true;
            break;
        default:
            break;
    }
    for (auto &c : n->in) mark_co_partitioned(p, c.get(), under);
}

inline bool build_plan(const JValue *root, Plan *plan) {
    Builder b{plan, {}};
    plan->root = b.node(root);
    if (!plan->root) {
        plan->why = b.err.empty() ? "unsupported plan" : b.err;
        return false;
    }
    mark_required(plan, plan->root.get(), std::vector<char>(plan->root->schema.size(), 1));
    mark_null_droppable(plan, plan->root.get(), std::vector<char>(plan->root->schema.size(), 0));
    mark_co_partitioned(plan, plan->root.get(), false);
    return true;
}

}  // namespace ir
}  // namespace flockgpu
