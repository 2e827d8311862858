// Plan-level C ABI (include/flockgpu_plan.h): serde_json physical plan in, Arrow C Data Interface batches in/out.
// Host-side code only; the compute goes through the flockgpu_q*_ entry points (one window = everything fed).
#include <algorithm>
#include <cstdlib>
#include <memory>

#include "../../include/flockgpu_plan.h"
#include "common.hpp"

using namespace flockgpu;

namespace {

// ------------------------------------------------------------------ minimal JSON
struct JValue;
using JPtr = std::shared_ptr<JValue>;
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    int64_t inum = 0;
    bool is_int = false;
    std::string str;
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;
    const JValue *get(const char *key) const {
        for (auto &kv : obj)
            if (kv.first == key) return kv.second.get();
        return nullptr;
    }
    std::string s(const char *key) const {
        const JValue *v = get(key);
        return v && v->kind == Str ? v->str : std::string();
    }
};

struct JParser {
    const char *p, *end;
    std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool fail(const char *m) { if (err.empty()) err = m; return false; }
    bool parse_string(std::string &out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (end - p < 5) return fail("bad \\u escape");
                        unsigned v = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                        if (v < 0x80) out += (char)v;
                        else if (v < 0x800) { out += (char)(0xC0 | (v >> 6)); out += (char)(0x80 | (v & 0x3F)); }
                        else { out += (char)(0xE0 | (v >> 12)); out += (char)(0x80 | ((v >> 6) & 0x3F)); out += (char)(0x80 | (v & 0x3F)); }
                        p += 4;
                        break;
                    }
                    default: out += *p;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    bool parse(JPtr &out, int depth = 0) {
        if (depth > 200) return fail("plan nested too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        out = std::make_shared<JValue>();
        if (*p == '{') {
            out->kind = JValue::Obj;
            ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!parse_string(key)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                ++p;
                JPtr v;
                if (!parse(v, depth + 1)) return false;
                out->obj.emplace_back(std::move(key), v);
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            out->kind = JValue::Arr;
            ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                JPtr v;
                if (!parse(v, depth + 1)) return false;
                out->arr.push_back(v);
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') { out->kind = JValue::Str; return parse_string(out->str); }
        if (!strncmp(p, "true", std::min<size_t>(4, end - p)) && end - p >= 4) { out->kind = JValue::Bool; out->b = true; p += 4; return true; }
        if (!strncmp(p, "false", std::min<size_t>(5, end - p)) && end - p >= 5) { out->kind = JValue::Bool; p += 5; return true; }
        if (!strncmp(p, "null", std::min<size_t>(4, end - p)) && end - p >= 4) { p += 4; return true; }
        const char *q = p;
        bool is_int = true;
        if (q < end && (*q == '-' || *q == '+')) ++q;
        while (q < end && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '-' || *q == '+')) {
            if (*q == '.' || *q == 'e' || *q == 'E') is_int = false;
            ++q;
        }
        if (q == p) return fail("unexpected character");
        std::string tok(p, q);
        out->kind = JValue::Num;
        out->num = strtod(tok.c_str(), nullptr);
        out->is_int = is_int;
        if (is_int) out->inum = strtoll(tok.c_str(), nullptr, 10);
        p = q;
        return true;
    }
};

// ------------------------------------------------------------------ plan recognition
const std::string &tag(const JValue *n) {
    static const std::string empty;
    const JValue *t = n ? n->get("execution_plan") : nullptr;
    return t && t->kind == JValue::Str ? t->str : empty;
}
const std::string &etag(const JValue *e) {
    static const std::string empty;
    const JValue *t = e ? e->get("physical_expr") : nullptr;
    return t && t->kind == JValue::Str ? t->str : empty;
}
// cast_expr / try_cast_expr are value-preserving for the widening casts the planner inserts (planner.rs:90,122,155)
const JValue *uncast(const JValue *e) {
    while (e && (etag(e) == "cast_expr" || etag(e) == "try_cast_expr")) e = e->get("expr");
    return e;
}
bool is_column(const JValue *e, std::string *name) {
    e = uncast(e);
    if (etag(e) != "column") return false;
    if (name) *name = e->s("name");
    return true;
}
// literal{"value": {"Int64": 123}} (also accepts a bare number / string)
bool literal_i64(const JValue *e, int64_t *v) {
    e = uncast(e);
    if (etag(e) != "literal") return false;
    const JValue *val = e->get("value");
    if (!val) return false;
    if (val->kind == JValue::Obj && val->obj.size() == 1) val = val->obj[0].second.get();
    if (val->kind != JValue::Num || !val->is_int) return false;
    *v = val->inum;
    return true;
}
bool literal_f64(const JValue *e, double *v) {
    e = uncast(e);
    if (etag(e) != "literal") return false;
    const JValue *val = e->get("value");
    if (!val) return false;
    if (val->kind == JValue::Obj && val->obj.size() == 1) val = val->obj[0].second.get();
    if (val->kind != JValue::Num) return false;
    *v = val->num;
    return true;
}
bool literal_utf8(const JValue *e, std::string *v) {
    e = uncast(e);
    if (etag(e) != "literal") return false;
    const JValue *val = e->get("value");
    if (!val) return false;
    if (val->kind == JValue::Obj && val->obj.size() == 1) val = val->obj[0].second.get();
    if (val->kind != JValue::Str) return false;
    *v = val->str;
    return true;
}
bool is_binary(const JValue *e, const char *op, const JValue **l, const JValue **r) {
    if (etag(e) != "binary_expr" || e->s("op") != op) return false;
    *l = e->get("left");
    *r = e->get("right");
    return *l && *r;
}
bool projection_is_passthrough(const JValue *n) {
    const JValue *ex = n->get("expr");
    if (!ex || ex->kind != JValue::Arr) return false;
    for (auto &pair : ex->arr) {
        if (pair->kind != JValue::Arr || pair->arr.empty()) return false;
        if (etag(pair->arr[0].get()) != "column") return false;
    }
    return true;
}
// Skips nodes that do not change the row multiset.
const JValue *strip(const JValue *n) {
    for (;;) {
        const std::string &t = tag(n);
        if (t == "coalesce_batches_exec" || t == "repartition_exec" || t == "merge_exec" ||
            t == "coalesce_partitions_exec" || (t == "projection_exec" && projection_is_passthrough(n))) {
            n = n->get("input");
            continue;
        }
        return n;
    }
}
std::vector<std::string> leaf_columns(const JValue *mem) {
    std::vector<std::string> all, out;
    const JValue *schema = mem->get("schema");
    const JValue *fields = schema ? schema->get("fields") : nullptr;
    if (fields && fields->kind == JValue::Arr)
        for (auto &f : fields->arr) all.push_back(f->s("name"));
    const JValue *proj = mem->get("projection");
    if (proj && proj->kind == JValue::Arr && !proj->arr.empty()) {
        for (auto &i : proj->arr)
            if (i->kind == JValue::Num && i->inum >= 0 && (size_t)i->inum < all.size()) out.push_back(all[i->inum]);
        // fixtures of older fork revisions list only the projected fields: indices then exceed the list
        if (out.size() != proj->arr.size()) out = all;
    } else {
        out = all;
    }
    return out;
}
// Logical aggregate = Final/FinalPartitioned over (transparent nodes over) Partial with the same grouping, or a
// single-stage aggregate.  Returns the aggregate's input (below the Partial stage).
struct Agg {
    std::vector<std::string> group;
    std::vector<std::string> kinds;  // "count", "max", ...
    const JValue *input = nullptr;
};
bool match_agg(const JValue *n, Agg *a) {
    n = strip(n);
    if (tag(n) != "hash_aggregate_exec") return false;
    auto groups = [](const JValue *x) {
        std::vector<std::string> g;
        const JValue *ge = x->get("group_expr");
        if (ge && ge->kind == JValue::Arr)
            for (auto &pair : ge->arr) {
                std::string name;
                if (pair->kind == JValue::Arr && !pair->arr.empty() && is_column(pair->arr[0].get(), &name)) g.push_back(name);
                else g.push_back("?");
            }
        return g;
    };
    a->group = groups(n);
    a->kinds.clear();
    const JValue *ae = n->get("aggr_expr");
    if (ae && ae->kind == JValue::Arr)
        for (auto &x : ae->arr) a->kinds.push_back(x->s("aggregate_expr"));
    const std::string mode = n->s("mode");
    const JValue *in = strip(n->get("input"));
    if ((mode == "Final" || mode == "FinalPartitioned") && tag(in) == "hash_aggregate_exec" && in->s("mode") == "Partial") {
        if (groups(in).size() != a->group.size()) return false;
        in = strip(in->get("input"));
    }
    a->input = in;
    return true;
}
const JValue *match_leaf(const JValue *n, const char *need_a, const char *need_b = nullptr) {
    n = strip(n);
    if (tag(n) != "memory_exec") return nullptr;
    auto cols = leaf_columns(n);
    auto has = [&](const char *c) { return !c || std::find(cols.begin(), cols.end(), c) != cols.end(); };
    return has(need_a) && has(need_b) ? n : nullptr;
}

struct HostCol {  // concatenated host copy (pass-through columns of q1)
    std::vector<uint8_t> bytes;
};
struct DevCol {
    std::string name;
    std::string format;       // Arrow format string expected: "i", "l"/"tsm:", "u"
    bool keep_host = false;   // q1 pass-through
    // device copies, filled by feed
    void *values = nullptr;   // fixed width values or Utf8 bytes
    int32_t *offsets = nullptr;
    int64_t bytes = 0;        // Utf8 bytes so far
    HostCol host;
};
struct Leaf {
    std::string relation;
    std::vector<DevCol> cols;
    int64_t rows = 0;
};

}  // namespace

struct flockgpu_plan {
    flockgpu_ctx *ctx = nullptr;
    int query = 0;
    std::vector<Leaf> leaves;
    // parameters lifted from the plan
    double q1_factor = 0.908;
    std::string q1_out_name = "price";
    int64_t q2_modulus = 123;
    int64_t q3_category = 10;
    std::vector<std::string> q3_states;
};

namespace {

DevCol col(const char *name, const char *fmt, bool keep_host = false) {
    DevCol c;
    c.name = name;
    c.format = fmt;
    c.keep_host = keep_host;
    return c;
}

// Recognises the five NEXMark stage shapes.  Returns false (UNSUPPORTED) for everything else.
bool recognise(const JValue *root, flockgpu_plan *pl, std::string *why) {
    const JValue *top = root;
    // the root projection carries the computed / selected columns; keep it unless it is pure pass-through
    const JValue *n = strip(top);
    const std::string &t = tag(n);

    // ---- q1: Projection [auction, bidder, 0.908 * CAST(price AS Float64) AS price, b_date_time] over bid
    if (tag(top) == "projection_exec" && !projection_is_passthrough(top)) {
        const JValue *in = strip(top->get("input"));
        const JValue *ex = top->get("expr");
        if (tag(in) == "memory_exec" && ex && ex->kind == JValue::Arr && ex->arr.size() == 4) {
            std::vector<std::string> names;
            bool ok = true;
            int computed = -1;
            for (size_t i = 0; i < 4 && ok; ++i) {
                const JValue *pair = ex->arr[i].get();
                if (pair->kind != JValue::Arr || pair->arr.size() < 2) { ok = false; break; }
                const JValue *e = pair->arr[0].get();
                std::string cname;
                const JValue *l, *r;
                if (etag(e) == "column") {
                    names.push_back(e->s("name"));
                } else if (is_binary(e, "Multiply", &l, &r)) {
                    double f;
                    if (literal_f64(l, &f) && is_column(r, &cname)) { pl->q1_factor = f; }
                    else if (literal_f64(r, &f) && is_column(l, &cname)) { pl->q1_factor = f; }
                    else ok = false;
                    names.push_back(cname);
                    computed = (int)i;
                    pl->q1_out_name = pair->arr[1]->str;
                } else ok = false;
            }
            if (ok && computed == 2 && names[0] == "auction" && names[1] == "bidder" && names[2] == "price" &&
                names[3] == "b_date_time") {
                pl->query = 1;
                Leaf lf;
                lf.relation = "bid";
                lf.cols = {col("auction", "i", true), col("bidder", "i", true), col("price", "i"), col("b_date_time", "tsm:", true)};
                pl->leaves = {lf};
                return true;
            }
        }
    }
    // ---- q2: Projection [auction, price] <- Filter CAST(auction AS Int64) % m = 0 <- bid
    if (t == "filter_exec") {
        const JValue *pred = n->get("predicate"), *l, *r, *ml, *mr;
        const JValue *leaf = match_leaf(n->get("input"), "auction", "price");
        int64_t m, rem;
        std::string cname;
        if (leaf && is_binary(pred, "Eq", &l, &r) && is_binary(uncast(l), "Modulo", &ml, &mr) && is_column(ml, &cname) &&
            cname == "auction" && literal_i64(mr, &m) && literal_i64(r, &rem) && rem == 0) {
            pl->query = 2;
            pl->q2_modulus = m;
            Leaf lf;
            lf.relation = "bid";
            lf.cols = {col("auction", "i"), col("price", "i")};
            pl->leaves = {lf};
            return true;
        }
    }
    if (t == "hash_join_exec") {
        const JValue *on = n->get("on");
        std::string lk, rk;
        if (on && on->kind == JValue::Arr && on->arr.size() == 1 && on->arr[0]->kind == JValue::Arr && on->arr[0]->arr.size() == 2) {
            auto keyname = [](const JValue *k) {
                if (k->kind == JValue::Str) return k->str;  // older fork revision: bare names
                return k->s("name");
            };
            lk = keyname(on->arr[0]->arr[0].get());
            rk = keyname(on->arr[0]->arr[1].get());
        }
        if (n->s("join_type") != "Inner") { *why = "only Inner joins"; return false; }
        const JValue *L = strip(n->get("left")), *R = strip(n->get("right"));
        // ---- q3: Filter(category = 10)(auction) JOIN Filter(state = .. OR ..)(person) ON seller = p_id
        if (lk == "seller" && rk == "p_id" && tag(L) == "filter_exec" && tag(R) == "filter_exec") {
            const JValue *al = match_leaf(L->get("input"), "seller", "category");
            const JValue *pr = match_leaf(R->get("input"), "p_id", "state");
            const JValue *l, *r;
            std::string cname;
            int64_t cat;
            bool ok = al && pr && is_binary(L->get("predicate"), "Eq", &l, &r) && is_column(l, &cname) && cname == "category" &&
                      literal_i64(r, &cat);
            std::vector<std::string> states;
            if (ok) {
                // flatten the OR chain of `state = literal`
                std::vector<const JValue *> stack{R->get("predicate")};
                while (!stack.empty() && ok) {
                    const JValue *e = stack.back();
                    stack.pop_back();
                    const JValue *a, *b;
                    std::string lit;
                    if (is_binary(e, "Or", &a, &b)) { stack.push_back(b); stack.push_back(a); }
                    else if (is_binary(e, "Eq", &a, &b) && is_column(a, &cname) && cname == "state" && literal_utf8(b, &lit)) states.push_back(lit);
                    else ok = false;
                }
            }
            if (ok && !states.empty()) {
                pl->query = 3;
                pl->q3_category = cat;
                pl->q3_states = states;
                Leaf a, p;
                a.relation = "auction";
                a.cols = {col("a_id", "i"), col("seller", "i"), col("category", "i")};
                p.relation = "person";
                p.cols = {col("p_id", "i"), col("name", "u"), col("city", "u"), col("state", "u")};
                pl->leaves = {a, p};
                return true;
            }
        }
        // ---- q8: DISTINCT(p_id, name)(person) JOIN DISTINCT(seller)(auction) ON p_id = seller
        if (lk == "p_id" && rk == "seller") {
            Agg la, ra;
            if (match_agg(L, &la) && match_agg(R, &ra) && la.kinds.empty() && ra.kinds.empty() &&
                la.group == std::vector<std::string>{"p_id", "name"} && ra.group == std::vector<std::string>{"seller"} &&
                match_leaf(la.input, "p_id", "name") && match_leaf(ra.input, "seller")) {
                pl->query = 8;
                Leaf p, a;
                p.relation = "person";
                p.cols = {col("p_id", "i"), col("name", "u")};
                a.relation = "auction";
                a.cols = {col("seller", "i")};
                pl->leaves = {p, a};
                return true;
            }
        }
        // ---- q5: (COUNT(*) GROUP BY auction) JOIN (MAX(num) over the same counts) ON num = maxn
        if (lk == "num" && rk == "maxn") {
            Agg cnt, mx, cnt2;
            if (match_agg(L, &cnt) && cnt.group == std::vector<std::string>{"auction"} && cnt.kinds == std::vector<std::string>{"count"} &&
                match_leaf(cnt.input, "auction") && match_agg(R, &mx) && mx.group.empty() && mx.kinds == std::vector<std::string>{"max"} &&
                match_agg(mx.input, &cnt2) && cnt2.group == std::vector<std::string>{"auction"} &&
                cnt2.kinds == std::vector<std::string>{"count"} && match_leaf(cnt2.input, "auction")) {
                pl->query = 5;
                Leaf b;
                b.relation = "bid";
                b.cols = {col("auction", "i")};
                // the SQL scans `bid` twice (no CSE in the reference): both leaves read the same relation
                pl->leaves = {b, b};
                return true;
            }
        }
        // ---- q13 ("next" query): bid JOIN side_input ON auction = key  (q13.sql; the side input is a bounded table)
        if (lk == "auction" && rk == "key") {
            if (match_leaf(L, "auction", "price") && match_leaf(L, "bidder", "b_date_time") && match_leaf(R, "key", "value")) {
                pl->query = 13;
                Leaf b, sd;
                b.relation = "bid";
                b.cols = {col("auction", "i"), col("bidder", "i"), col("price", "i"), col("b_date_time", "tsm:")};
                sd.relation = "side_input";
                sd.cols = {col("key", "i"), col("value", "i")};
                pl->leaves = {b, sd};
                return true;
            }
        }
        // ---- q7 ("next" query): bid JOIN (MAX(price) AS maxprice over bid) ON price = maxprice
        if (lk == "price" && rk == "maxprice") {
            Agg mx;
            if (match_leaf(L, "auction", "price") && match_leaf(L, "bidder", "b_date_time") && match_agg(R, &mx) && mx.group.empty() &&
                mx.kinds == std::vector<std::string>{"max"} && match_leaf(mx.input, "price")) {
                pl->query = 7;
                Leaf b;
                b.relation = "bid";
                b.cols = {col("auction", "i"), col("bidder", "i"), col("price", "i"), col("b_date_time", "tsm:")};
                pl->leaves = {b, b};  // the SQL scans `bid` twice; whichever leaf is fed holds the relation
                return true;
            }
        }
    }
    *why = "plan shape is not NEXMark q1/q2/q3/q5/q7/q8/q13";
    return false;
}

// ------------------------------------------------------------------ feeding
int find_child(const ArrowSchema *schema, const std::string &name) {
    for (int64_t i = 0; i < schema->n_children; ++i)
        if (schema->children[i] && schema->children[i]->name && name == schema->children[i]->name) return (int)i;
    return -1;
}
bool format_ok(const std::string &want, const char *got) {
    if (!got) return false;
    if (want == "tsm:") return !strncmp(got, "tsm:", 4) || !strcmp(got, "l");
    return want == got;
}
size_t width_of(const std::string &fmt) { return fmt == "i" ? 4 : 8; }

struct Grow {  // device buffer that keeps its contents when it grows (append-only feeding)
    static int ensure(flockgpu_ctx *ctx, const std::string &key, size_t keep_bytes, size_t want_bytes, void **ptr) {
        DeviceBuf &b = ctx->arena[key];
        if (b.cap >= want_bytes && b.ptr) { *ptr = b.ptr; return FLOCKGPU_OK; }
        size_t cap = std::max<size_t>(want_bytes + want_bytes / 2, 1024);
        cap = (cap + 255) & ~size_t(255);
        void *np = nullptr;
        hipError_t e = hipMalloc(&np, cap);
        if (e != hipSuccess) return fail(ctx, FLOCKGPU_ERR_OOM, "plan feed: hipMalloc(%zu): %s", cap, hipGetErrorString(e));
        if (b.ptr) {
            if (keep_bytes) {
                e = hipMemcpyAsync(np, b.ptr, keep_bytes, hipMemcpyDeviceToDevice, ctx->stream);
                if (e != hipSuccess) { (void)hipFree(np); return fail(ctx, FLOCKGPU_ERR_HIP, "plan feed: grow copy: %s", hipGetErrorString(e)); }
            }
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(b.ptr);
        }
        b.ptr = np;
        b.cap = cap;
        *ptr = np;
        return FLOCKGPU_OK;
    }
};

std::string key_of(const flockgpu_plan *pl, int leaf, const DevCol &c, const char *what) {
    char buf[160];
    snprintf(buf, sizeof buf, "plan%p.%d.%s.%s", (const void *)pl, leaf, c.name.c_str(), what);
    return buf;
}

// ------------------------------------------------------------------ Arrow export
struct ExportPriv {
    std::vector<void *> owned;                    // malloc'd buffers
    std::vector<const void *> buffers;            // this node's buffer pointers
    std::vector<ArrowArray *> child_ptrs;
    std::vector<std::unique_ptr<ArrowArray>> children;
};
void release_array(ArrowArray *a) {
    if (!a || !a->release) return;
    ExportPriv *p = static_cast<ExportPriv *>(a->private_data);
    if (p) {
        for (auto &c : p->children)
            if (c && c->release) c->release(c.get());
        for (void *o : p->owned) free(o);
        delete p;
    }
    a->release = nullptr;
}
struct SchemaPriv {
    std::string format, name;
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
void make_struct_array(ArrowArray *a, int64_t length) {
    ExportPriv *p = new ExportPriv();
    std::memset(a, 0, sizeof *a);
    a->length = length;
    p->buffers = {nullptr};  // validity
    a->n_buffers = 1;
    a->buffers = p->buffers.data();
    a->release = release_array;
    a->private_data = p;
}
// takes ownership of the malloc'd buffers
void add_array_child(ArrowArray *parent, int64_t length, void *values, void *offsets /*nullable*/) {
    ExportPriv *pp = static_cast<ExportPriv *>(parent->private_data);
    pp->children.emplace_back(new ArrowArray());
    ArrowArray *c = pp->children.back().get();
    ExportPriv *p = new ExportPriv();
    std::memset(c, 0, sizeof *c);
    c->length = length;
    if (offsets) {
        p->buffers = {nullptr, offsets, values};
        p->owned = {offsets, values};
    } else {
        p->buffers = {nullptr, values};
        p->owned = {values};
    }
    c->n_buffers = (int64_t)p->buffers.size();
    c->buffers = p->buffers.data();
    c->release = release_array;
    c->private_data = p;
    pp->child_ptrs.push_back(c);
    parent->children = pp->child_ptrs.data();
    parent->n_children = (int64_t)pp->child_ptrs.size();
}

int d2h_alloc(flockgpu_ctx *ctx, const void *dev, size_t bytes, void **out) {
    void *h = malloc(bytes ? bytes : 8);
    if (!h) return fail(ctx, FLOCKGPU_ERR_OOM, "plan execute: malloc(%zu)", bytes);
    if (bytes) {
        hipError_t e = hipMemcpyAsync(h, dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { free(h); return fail(ctx, FLOCKGPU_ERR_HIP, "plan execute: D2H: %s", hipGetErrorString(e)); }
    }
    *out = h;
    return FLOCKGPU_OK;
}
int host_dup(flockgpu_ctx *ctx, const std::vector<uint8_t> &src, void **out) {
    void *h = malloc(src.size() ? src.size() : 8);
    if (!h) return fail(ctx, FLOCKGPU_ERR_OOM, "plan execute: malloc(%zu)", src.size());
    if (!src.empty()) std::memcpy(h, src.data(), src.size());
    *out = h;
    return FLOCKGPU_OK;
}

flockgpu_utf8 dev_utf8(const DevCol &c) { return flockgpu_utf8{c.offsets, static_cast<const uint8_t *>(c.values)}; }

}  // namespace

extern "C" {

int flockgpu_plan_create(flockgpu_ctx *ctx, const char *plan_json, size_t len, flockgpu_plan **out) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!plan_json || !out) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_create: null argument");
    *out = nullptr;
    JParser jp{plan_json, plan_json + len, {}};
    JPtr root;
    if (!jp.parse(root) || root->kind != JValue::Obj)
        return fail(ctx, FLOCKGPU_ERR_PLAN, "plan_create: JSON error: %s", jp.err.empty() ? "not an object" : jp.err.c_str());
    std::unique_ptr<flockgpu_plan> pl(new flockgpu_plan());
    pl->ctx = ctx;
    std::string why;
    if (!recognise(root.get(), pl.get(), &why)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_create: %s", why.c_str());
    *out = pl.release();
    return FLOCKGPU_OK;
}

int flockgpu_plan_recognise(const char *plan_json, size_t len, int *query) {
    if (!plan_json || !query) return FLOCKGPU_ERR_INVALID;
    JParser jp{plan_json, plan_json + len, {}};
    JPtr root;
    if (!jp.parse(root) || root->kind != JValue::Obj) return FLOCKGPU_ERR_PLAN;
    flockgpu_plan pl;
    std::string why;
    if (!recognise(root.get(), &pl, &why)) return FLOCKGPU_ERR_UNSUPPORTED;
    *query = pl.query;
    return FLOCKGPU_OK;
}

void flockgpu_plan_destroy(flockgpu_plan *plan) {
    if (!plan) return;
    flockgpu_ctx *ctx = plan->ctx;
    (void)hipStreamSynchronize(ctx->stream);
    char prefix[64];
    snprintf(prefix, sizeof prefix, "plan%p.", (const void *)plan);
    for (auto it = ctx->arena.begin(); it != ctx->arena.end();) {
        if (it->first.compare(0, strlen(prefix), prefix) == 0) {
            if (it->second.ptr) (void)hipFree(it->second.ptr);
            it = ctx->arena.erase(it);
        } else {
            ++it;
        }
    }
    delete plan;
}

int flockgpu_plan_query(const flockgpu_plan *plan) { return plan ? plan->query : 0; }
int flockgpu_plan_num_inputs(const flockgpu_plan *plan) { return plan ? (int)plan->leaves.size() : 0; }
const char *flockgpu_plan_input_name(const flockgpu_plan *plan, int input) {
    if (!plan || input < 0 || input >= (int)plan->leaves.size()) return nullptr;
    return plan->leaves[input].relation.c_str();
}
int flockgpu_plan_input_matches(const flockgpu_plan *plan, int input, const struct ArrowSchema *schema) {
    if (!plan || !schema || input < 0 || input >= (int)plan->leaves.size()) return 0;
    for (auto &c : plan->leaves[input].cols)
        if (find_child(schema, c.name) < 0) return 0;
    return 1;
}

int flockgpu_plan_feed(flockgpu_plan *plan, int input, const struct ArrowSchema *schema,
                       const struct ArrowArray *const *batches, int n_batches) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    flockgpu_ctx *ctx = plan->ctx;
    if (!schema || input < 0 || input >= (int)plan->leaves.size() || n_batches < 0 || (n_batches && !batches))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: bad argument");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    Leaf &lf = plan->leaves[input];
    std::vector<int> child(lf.cols.size());
    for (size_t c = 0; c < lf.cols.size(); ++c) {
        child[c] = find_child(schema, lf.cols[c].name);
        if (child[c] < 0) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: column '%s' missing from the fed schema", lf.cols[c].name.c_str());
        if (!format_ok(lf.cols[c].format, schema->children[child[c]]->format))
            return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed: column '%s' has Arrow format '%s', expected '%s'",
                        lf.cols[c].name.c_str(), schema->children[child[c]]->format, lf.cols[c].format.c_str());
    }
    std::vector<int32_t> rebased;
    for (int b = 0; b < n_batches; ++b) {
        const ArrowArray *rb = batches[b];
        if (!rb || rb->n_children < schema->n_children) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: batch %d does not match the schema", b);
        const int64_t n = rb->length;
        if (n == 0) continue;
        if (lf.rows + n >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed: more than 2^31 rows per relation");
        for (size_t c = 0; c < lf.cols.size(); ++c) {
            DevCol &dc = lf.cols[c];
            const ArrowArray *a = rb->children[child[c]];
            if (!a || a->length != n) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: column '%s' length mismatch", dc.name.c_str());
            if (a->null_count > 0) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed: column '%s' holds NULLs (NEXMark fields are non-nullable)", dc.name.c_str());
            const int64_t off = a->offset + rb->offset;
            if (dc.format == "u") {
                if (a->n_buffers < 3) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: Utf8 column '%s' without 3 buffers", dc.name.c_str());
                const int32_t *src_off = static_cast<const int32_t *>(a->buffers[1]) + off;
                const uint8_t *src = static_cast<const uint8_t *>(a->buffers[2]);
                const int64_t b0 = src_off[0], nbytes = (int64_t)src_off[n] - b0;
                if (dc.bytes + nbytes >= (int64_t(1) << 31)) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_feed: Utf8 column exceeds 2^31 bytes");
                void *p = nullptr;
                FG_TRY(Grow::ensure(ctx, key_of(plan, input, dc, "off"), (size_t)(lf.rows + 1) * 4, (size_t)(lf.rows + n + 1) * 4, &p));
                dc.offsets = static_cast<int32_t *>(p);
                FG_TRY(Grow::ensure(ctx, key_of(plan, input, dc, "bytes"), (size_t)dc.bytes, (size_t)(dc.bytes + nbytes) + 16, &p));
                dc.values = p;
                rebased.resize((size_t)n + 1);
                for (int64_t i = 0; i <= n; ++i) rebased[i] = (int32_t)(src_off[i] - b0 + dc.bytes);
                // offsets[rows .. rows + n] (the shared boundary entry is rewritten with the same value)
                FG_HIP(ctx, hipMemcpyAsync(dc.offsets + lf.rows, rebased.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
                if (nbytes) FG_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(dc.values) + dc.bytes, src + b0, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream));
                FG_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `rebased` is reused for the next column
                dc.bytes += nbytes;
            } else {
                const size_t w = width_of(dc.format);
                if (a->n_buffers < 2) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_feed: column '%s' without a data buffer", dc.name.c_str());
                const uint8_t *src = static_cast<const uint8_t *>(a->buffers[1]) + (size_t)off * w;
                void *p = nullptr;
                FG_TRY(Grow::ensure(ctx, key_of(plan, input, dc, "val"), (size_t)lf.rows * w, (size_t)(lf.rows + n) * w + 16, &p));
                dc.values = p;
                FG_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(dc.values) + (size_t)lf.rows * w, src, (size_t)n * w, hipMemcpyHostToDevice, ctx->stream));
                if (dc.keep_host) dc.host.bytes.insert(dc.host.bytes.end(), src, src + (size_t)n * w);
            }
        }
        lf.rows += n;
    }
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FLOCKGPU_OK;
}

int flockgpu_plan_reset(flockgpu_plan *plan) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    for (auto &lf : plan->leaves) {
        lf.rows = 0;
        for (auto &c : lf.cols) {
            c.bytes = 0;
            c.host.bytes.clear();
        }
    }
    return FLOCKGPU_OK;
}

int flockgpu_plan_execute(flockgpu_plan *plan, struct ArrowSchema *out_schema, struct ArrowArray *out_batch) {
    if (!plan) return FLOCKGPU_ERR_INVALID;
    flockgpu_ctx *ctx = plan->ctx;
    if (!out_schema || !out_batch) return fail(ctx, FLOCKGPU_ERR_INVALID, "plan_execute: null output");
    FG_HIP(ctx, hipSetDevice(ctx->device));
    auto whole = [](int64_t rows, int64_t (&off)[2], int32_t (&lo)[1], int32_t (&hi)[1]) {
        off[0] = 0; off[1] = rows; lo[0] = 0; hi[0] = 1;
        return flockgpu_windows{off, 1, lo, hi, 1};
    };
    auto dummy_utf8 = [&](DevCol &c) -> int {  // an unfed Utf8 column still needs a one-entry offsets array
        if (c.offsets) return FLOCKGPU_OK;
        void *p = nullptr;
        FG_TRY(Grow::ensure(ctx, key_of(plan, 0, c, "empty"), 0, 64, &p));
        FG_HIP(ctx, hipMemsetAsync(p, 0, 64, ctx->stream));
        c.offsets = static_cast<int32_t *>(p);
        c.values = static_cast<uint8_t *>(p) + 16;
        return FLOCKGPU_OK;
    };
    int64_t off_a[2], off_b[2];
    int32_t lo_a[1], hi_a[1], lo_b[1], hi_b[1];

    if (plan->query == 1) {
        Leaf &b = plan->leaves[0];
        double *d_out = nullptr;
        FG_TRY(arena_get_t(ctx, "plan.q1.out", (size_t)b.rows + 2, &d_out));
        flockgpu_bid_cols bc{nullptr, nullptr, static_cast<const int32_t *>(b.cols[2].values), nullptr, b.rows};
        FG_TRY(flockgpu_q1_project(ctx, &bc, plan->q1_factor, d_out));
        void *h_price, *h_a, *h_b, *h_t;
        FG_TRY(d2h_alloc(ctx, d_out, (size_t)b.rows * 8, &h_price));
        FG_TRY(host_dup(ctx, b.cols[0].host.bytes, &h_a));
        FG_TRY(host_dup(ctx, b.cols[1].host.bytes, &h_b));
        FG_TRY(host_dup(ctx, b.cols[3].host.bytes, &h_t));
        make_schema(out_schema, "+s", "", false);
        add_schema_child(out_schema, "i", "auction", false);
        add_schema_child(out_schema, "i", "bidder", false);
        add_schema_child(out_schema, "g", plan->q1_out_name.c_str(), false);
        add_schema_child(out_schema, "tsm:", "b_date_time", false);
        make_struct_array(out_batch, b.rows);
        add_array_child(out_batch, b.rows, h_a, nullptr);
        add_array_child(out_batch, b.rows, h_b, nullptr);
        add_array_child(out_batch, b.rows, h_price, nullptr);
        add_array_child(out_batch, b.rows, h_t, nullptr);
        return FLOCKGPU_OK;
    }
    if (plan->query == 2) {
        Leaf &b = plan->leaves[0];
        flockgpu_bid_cols bc{static_cast<const int32_t *>(b.cols[0].values), nullptr, static_cast<const int32_t *>(b.cols[1].values), nullptr, b.rows};
        flockgpu_windows w = whole(b.rows, off_a, lo_a, hi_a);
        flockgpu_q2_result r{};
        FG_TRY(flockgpu_q2_filter(ctx, &bc, &w, plan->q2_modulus, &r));
        void *h_a, *h_p;
        FG_TRY(d2h_alloc(ctx, r.auction, (size_t)r.rows * 4, &h_a));
        FG_TRY(d2h_alloc(ctx, r.price, (size_t)r.rows * 4, &h_p));
        make_schema(out_schema, "+s", "", false);
        add_schema_child(out_schema, "i", "auction", false);
        add_schema_child(out_schema, "i", "price", false);
        make_struct_array(out_batch, r.rows);
        add_array_child(out_batch, r.rows, h_a, nullptr);
        add_array_child(out_batch, r.rows, h_p, nullptr);
        return FLOCKGPU_OK;
    }
    if (plan->query == 3) {
        Leaf &a = plan->leaves[0], &p = plan->leaves[1];
        for (int c = 1; c < 4; ++c) FG_TRY(dummy_utf8(p.cols[c]));
        flockgpu_auction_cols ac{static_cast<const int32_t *>(a.cols[0].values), static_cast<const int32_t *>(a.cols[1].values),
                                 static_cast<const int32_t *>(a.cols[2].values), a.rows};
        flockgpu_person_cols pc{static_cast<const int32_t *>(p.cols[0].values), dev_utf8(p.cols[1]), dev_utf8(p.cols[2]), dev_utf8(p.cols[3]), p.rows};
        flockgpu_windows aw = whole(a.rows, off_a, lo_a, hi_a), pw = whole(p.rows, off_b, lo_b, hi_b);
        std::vector<const char *> lits;
        for (auto &s : plan->q3_states) lits.push_back(s.c_str());
        flockgpu_q3_result r{};
        FG_TRY(flockgpu_q3_join(ctx, &ac, &aw, &pc, &pw, plan->q3_category, lits.data(), (int)lits.size(), &r));
        void *h[7];
        FG_TRY(d2h_alloc(ctx, r.name.offsets, (size_t)(r.rows + 1) * 4, &h[0]));
        FG_TRY(d2h_alloc(ctx, r.name.data, (size_t)r.name_bytes, &h[1]));
        FG_TRY(d2h_alloc(ctx, r.city.offsets, (size_t)(r.rows + 1) * 4, &h[2]));
        FG_TRY(d2h_alloc(ctx, r.city.data, (size_t)r.city_bytes, &h[3]));
        FG_TRY(d2h_alloc(ctx, r.state.offsets, (size_t)(r.rows + 1) * 4, &h[4]));
        FG_TRY(d2h_alloc(ctx, r.state.data, (size_t)r.state_bytes, &h[5]));
        FG_TRY(d2h_alloc(ctx, r.a_id, (size_t)r.rows * 4, &h[6]));
        make_schema(out_schema, "+s", "", false);
        add_schema_child(out_schema, "u", "name", false);
        add_schema_child(out_schema, "u", "city", false);
        add_schema_child(out_schema, "u", "state", false);
        add_schema_child(out_schema, "i", "a_id", false);
        make_struct_array(out_batch, r.rows);
        add_array_child(out_batch, r.rows, h[1], h[0]);
        add_array_child(out_batch, r.rows, h[3], h[2]);
        add_array_child(out_batch, r.rows, h[5], h[4]);
        add_array_child(out_batch, r.rows, h[6], nullptr);
        return FLOCKGPU_OK;
    }
    if (plan->query == 5) {
        // both leaves scan `bid`; whichever was fed holds the relation (feed_data_sources gives it to the first match)
        Leaf &b = plan->leaves[0].rows ? plan->leaves[0] : plan->leaves[1];
        flockgpu_bid_cols bc{static_cast<const int32_t *>(b.cols[0].values), nullptr, nullptr, nullptr, b.rows};
        flockgpu_windows w = whole(b.rows, off_a, lo_a, hi_a);
        flockgpu_q5_result r{};
        FG_TRY(flockgpu_q5_hot_items(ctx, &bc, &w, &r));
        void *h_a, *h_n;
        FG_TRY(d2h_alloc(ctx, r.auction, (size_t)r.rows * 4, &h_a));
        FG_TRY(d2h_alloc(ctx, r.num, (size_t)r.rows * 8, &h_n));
        make_schema(out_schema, "+s", "", false);
        add_schema_child(out_schema, "i", "auction", false);
        add_schema_child(out_schema, "L", "num", true);  // COUNT(*) -> UInt64, nullable in the schema (q5_plan.fmt:1)
        make_struct_array(out_batch, r.rows);
        add_array_child(out_batch, r.rows, h_a, nullptr);
        add_array_child(out_batch, r.rows, h_n, nullptr);
        return FLOCKGPU_OK;
    }
    if (plan->query == 7) {
        Leaf &b = plan->leaves[0].rows ? plan->leaves[0] : plan->leaves[1];
        flockgpu_bid_cols bc{static_cast<const int32_t *>(b.cols[0].values), static_cast<const int32_t *>(b.cols[1].values),
                             static_cast<const int32_t *>(b.cols[2].values), static_cast<const int64_t *>(b.cols[3].values), b.rows};
        flockgpu_windows w = whole(b.rows, off_a, lo_a, hi_a);
        flockgpu_q7_result r{};
        FG_TRY(flockgpu_q7_highest_bid(ctx, &bc, &w, &r));
        void *h_a, *h_p, *h_b, *h_t;
        FG_TRY(d2h_alloc(ctx, r.auction, (size_t)r.rows * 4, &h_a));
        FG_TRY(d2h_alloc(ctx, r.price, (size_t)r.rows * 4, &h_p));
        FG_TRY(d2h_alloc(ctx, r.bidder, (size_t)r.rows * 4, &h_b));
        FG_TRY(d2h_alloc(ctx, r.b_date_time, (size_t)r.rows * 8, &h_t));
        make_schema(out_schema, "+s", "", false);
        add_schema_child(out_schema, "i", "auction", false);   // q7_plan.fmt:1
        add_schema_child(out_schema, "i", "price", false);
        add_schema_child(out_schema, "i", "bidder", false);
        add_schema_child(out_schema, "tsm:", "b_date_time", false);
        make_struct_array(out_batch, r.rows);
        add_array_child(out_batch, r.rows, h_a, nullptr);
        add_array_child(out_batch, r.rows, h_p, nullptr);
        add_array_child(out_batch, r.rows, h_b, nullptr);
        add_array_child(out_batch, r.rows, h_t, nullptr);
        return FLOCKGPU_OK;
    }
    if (plan->query == 13) {
        Leaf &b = plan->leaves[0], &sd = plan->leaves[1];
        flockgpu_bid_cols bc{static_cast<const int32_t *>(b.cols[0].values), static_cast<const int32_t *>(b.cols[1].values),
                             static_cast<const int32_t *>(b.cols[2].values), static_cast<const int64_t *>(b.cols[3].values), b.rows};
        flockgpu_windows w = whole(b.rows, off_a, lo_a, hi_a);
        flockgpu_q13_result r{};
        FG_TRY(flockgpu_q13_side_join(ctx, &bc, &w, static_cast<const int32_t *>(sd.cols[0].values),
                                      static_cast<const int32_t *>(sd.cols[1].values), sd.rows, &r));
        void *h_a, *h_b, *h_p, *h_t, *h_v;
        FG_TRY(d2h_alloc(ctx, r.auction, (size_t)r.rows * 4, &h_a));
        FG_TRY(d2h_alloc(ctx, r.bidder, (size_t)r.rows * 4, &h_b));
        FG_TRY(d2h_alloc(ctx, r.price, (size_t)r.rows * 4, &h_p));
        FG_TRY(d2h_alloc(ctx, r.b_date_time, (size_t)r.rows * 8, &h_t));
        FG_TRY(d2h_alloc(ctx, r.value, (size_t)r.rows * 4, &h_v));
        make_schema(out_schema, "+s", "", false);
        add_schema_child(out_schema, "i", "auction", false);   // q13_plan.fmt:1
        add_schema_child(out_schema, "i", "bidder", false);
        add_schema_child(out_schema, "i", "price", false);
        add_schema_child(out_schema, "tsm:", "b_date_time", false);
        add_schema_child(out_schema, "i", "value", false);
        make_struct_array(out_batch, r.rows);
        add_array_child(out_batch, r.rows, h_a, nullptr);
        add_array_child(out_batch, r.rows, h_b, nullptr);
        add_array_child(out_batch, r.rows, h_p, nullptr);
        add_array_child(out_batch, r.rows, h_t, nullptr);
        add_array_child(out_batch, r.rows, h_v, nullptr);
        return FLOCKGPU_OK;
    }
    if (plan->query == 8) {
        Leaf &p = plan->leaves[0], &a = plan->leaves[1];
        FG_TRY(dummy_utf8(p.cols[1]));
        flockgpu_person_cols pc{static_cast<const int32_t *>(p.cols[0].values), dev_utf8(p.cols[1]), {nullptr, nullptr}, {nullptr, nullptr}, p.rows};
        flockgpu_auction_cols ac{nullptr, static_cast<const int32_t *>(a.cols[0].values), nullptr, a.rows};
        flockgpu_windows pw = whole(p.rows, off_a, lo_a, hi_a), aw = whole(a.rows, off_b, lo_b, hi_b);
        flockgpu_q8_result r{};
        FG_TRY(flockgpu_q8_join(ctx, &pc, &pw, &ac, &aw, &r));
        void *h_id, *h_off, *h_bytes;
        FG_TRY(d2h_alloc(ctx, r.p_id, (size_t)r.rows * 4, &h_id));
        FG_TRY(d2h_alloc(ctx, r.name.offsets, (size_t)(r.rows + 1) * 4, &h_off));
        FG_TRY(d2h_alloc(ctx, r.name.data, (size_t)r.name_bytes, &h_bytes));
        make_schema(out_schema, "+s", "", false);
        add_schema_child(out_schema, "i", "p_id", false);
        add_schema_child(out_schema, "u", "name", false);
        make_struct_array(out_batch, r.rows);
        add_array_child(out_batch, r.rows, h_id, nullptr);
        add_array_child(out_batch, r.rows, h_bytes, h_off);
        return FLOCKGPU_OK;
    }
    return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "plan_execute: unknown query %d", plan->query);
}

}  // extern "C"
