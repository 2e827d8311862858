"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by flock_amd).

Pure-Python, row-at-a-time restatement of the DataFusion physical operators that
Flock's hot path executes (FilterExec, ProjectionExec, HashAggregateExec,
HashJoinExec(Inner), SortExec, GlobalLimitExec, CoalesceBatchesExec,
RepartitionExec).  Small cases only.  Its job is to be *pinned* against every
operator-level golden vector the reference's tests hold at the
`ExecutionContext::execute` boundary, and then to pin the scalar C oracle
(oracle/nexmark_ops.c) and the HIP kernels on NEXMark windows:

    reference goldens  ->  generic_ops.py  ->  nexmark_ops.c  ->  HIP kernels
    (context.rs:493-503, 579-589; launcher/local.rs:223-231; transmute.rs:298-393)

A table is ``dict[str, list]`` (column name -> python values), a batch list is a
list of tables.  The operator arithmetic itself is upstream DataFusion ~6.x
(un-vendored dependency, flock/Cargo.toml:21); semantics restated per
SURVEY.md appendix D.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, List, Sequence, Tuple

Table = Dict[str, list]


def num_rows(t: Table) -> int:
    return len(next(iter(t.values()))) if t else 0


def rows(t: Table) -> List[tuple]:
    return list(zip(*t.values())) if t else []


def concat(batches: Sequence[Table]) -> Table:
    out: Table = {k: [] for k in batches[0]}
    for b in batches:
        for k in out:
            out[k].extend(b[k])
    return out


# -- FilterExec: keep rows whose predicate is True (NULL -> dropped), order preserved -------------
def filter_exec(t: Table, pred: Callable[[dict], bool]) -> Table:
    names = list(t)
    keep = [i for i, r in enumerate(rows(t)) if pred(dict(zip(names, r))) is True]
    return {k: [v[i] for i in keep] for k, v in t.items()}


# -- physical expressions of the plan-JSON dialect, SQL three-valued logic ------------------------------------
def _trunc_mod(a, m):
    """Rust / Arrow `%` on integers: the remainder of the TRUNCATED quotient (sign of the dividend)."""
    q = abs(a) // abs(m)
    q = q if (a >= 0) == (m >= 0) else -q
    return a - q * m


def eval_physical_expr(e: dict, row: dict):
    """Value of one serialised PhysicalExpr for one row (a dict column name -> python value, None = NULL): `column`, `literal`,
    `cast_expr` / `try_cast_expr`, `binary_expr` (Eq NotEq Lt LtEq Gt GtEq And Or Modulo Multiply Plus Minus), `not_expr`,
    `is_null_expr`, `is_not_null_expr`, `negative_expr`, `in_list_expr`.  Semantics: upstream DataFusion ~6:
    a comparison / arithmetic with NULL is NULL; AND is FALSE as soon as one side is, OR is TRUE as soon as one side is (Kleene);
    NOT NULL is NULL; x IN (..) is TRUE on a match, NULL when x is NULL, else FALSE; IS NULL never yields NULL."""
    t = e["physical_expr"]
    if t == "column":
        return row[e["name"]]
    if t == "literal":
        v = e["value"]
        return next(iter(v.values())) if isinstance(v, dict) else v
    if t in ("cast_expr", "try_cast_expr"):
        v = eval_physical_expr(e["expr"], row)
        if v is None:
            return None
        ty = e["cast_type"]
        if ty in ("Float64", "Float32"):
            return float(v)
        if isinstance(ty, str) and ty.startswith(("Int", "UInt")):
            return int(v)
        return v
    if t == "not_expr":
        v = eval_physical_expr(e.get("arg", e.get("expr")), row)
        return None if v is None else (not v)
    if t == "is_null_expr":
        return eval_physical_expr(e.get("arg", e.get("expr")), row) is None
    if t == "is_not_null_expr":
        return eval_physical_expr(e.get("arg", e.get("expr")), row) is not None
    if t == "negative_expr":
        v = eval_physical_expr(e.get("arg", e.get("expr")), row)
        return None if v is None else -v
    if t == "in_list_expr":
        v = eval_physical_expr(e["expr"], row)
        if v is None:
            return None
        hit = any(v == eval_physical_expr(x, row) for x in e["list"])
        return (not hit) if e.get("negated") else hit
    if t == "binary_expr":
        op = e["op"]
        a, b = eval_physical_expr(e["left"], row), eval_physical_expr(e["right"], row)
        if op == "And":
            return False if (a is False or b is False) else (None if (a is None or b is None) else True)
        if op == "Or":
            return True if (a is True or b is True) else (None if (a is None or b is None) else False)
        if a is None or b is None:
            return None
        if op == "Eq":
            return a == b
        if op == "NotEq":
            return a != b
        if op == "Lt":
            return a < b
        if op == "LtEq":
            return a <= b
        if op == "Gt":
            return a > b
        if op == "GtEq":
            return a >= b
        if op == "Modulo":
            return _trunc_mod(a, b)
        if op == "Multiply":
            return a * b
        if op == "Plus":
            return a + b
        if op == "Minus":
            return a - b
        raise ValueError("binary operator " + op)
    raise ValueError("physical_expr " + t)


# -- the same dialect WITH its types: arithmetic that wraps at the operand type's width, checked casts, division, CASE ---------------------
# (the general expression evaluator of the HIP path, flock_amd/csrc/valprog.hpp, states the assumptions; this is their twin: upstream
# DataFusion ~6 / arrow-rs 6 semantics, not pinned by reference-held vectors)
class ExprError(Exception):
    """An expression that fails the whole call: integer division by zero (ArrowError::DivideByZero), a CAST that does not fit."""


_INT_RANGE = {"Int32": (-2**31, 2**31 - 1), "Int64": (-2**63, 2**63 - 1), "UInt64": (0, 2**64 - 1)}
_ARITH = ("Plus", "Minus", "Multiply", "Divide", "Modulo")


def _type_name(t):
    return "Int64" if isinstance(t, dict) and "Timestamp" in t else t


def static_type(e: dict, types: dict):
    """Type of an expression: 'Int32' | 'Int64' | 'UInt64' | 'Float64' | 'Utf8' | 'Boolean', None for a literal that takes the type of what
    it meets (a bare NULL, a literal of a width the boundary has no column for)."""
    t = e["physical_expr"]
    if t == "column":
        return _type_name(types[e["name"]])
    if t == "literal":
        v = e["value"]
        kind = next(iter(v)) if isinstance(v, dict) else None
        val = next(iter(v.values())) if isinstance(v, dict) else v
        if val is None:
            return None
        if kind == "Boolean" or isinstance(val, bool):
            return "Boolean"
        if kind == "Utf8" or isinstance(val, str):
            return "Utf8"
        if kind in ("Float64", "Float32") or isinstance(val, float):
            return "Float64"
        return kind if kind in _INT_RANGE else None
    if t in ("cast_expr", "try_cast_expr"):
        return _type_name(e["cast_type"])
    if t == "negative_expr":
        return static_type(e.get("arg", e.get("expr")), types)
    if t in ("not_expr", "is_null_expr", "is_not_null_expr", "in_list_expr"):
        return "Boolean"
    if t == "binary_expr":
        if e["op"] not in _ARITH:
            return "Boolean"
        a, b = static_type(e["left"], types), static_type(e["right"], types)
        return a if a is not None else b
    if t == "case_expr":
        for _, th in e["when_then_expr"]:
            ty = static_type(th, types)
            if ty is not None:
                return ty
        return static_type(e["else_expr"], types) if e.get("else_expr") else None
    raise ValueError("physical_expr " + t)


def _wrap(v, ty):
    if ty in _INT_RANGE:
        lo, hi = _INT_RANGE[ty]
        return (v - lo) % (hi - lo + 1) + lo
    return v


def _cast(v, ty, safe):
    """arrow's numeric cast: a value that does not fit is an error (CAST, safe = false) or NULL (TRY_CAST)."""
    import math
    if ty == "Float64":
        return float(v)
    if ty in _INT_RANGE:
        lo, hi = _INT_RANGE[ty]
        if isinstance(v, float):
            if math.isnan(v) or math.isinf(v):
                v = None
            else:
                v = int(v)   # truncates towards zero
        if v is None or v < lo or v > hi:
            if safe:
                return None
            raise ExprError("a value does not fit the type it is cast to")
        return v
    return v


def eval_typed(e: dict, row: dict, types: dict, want=None):
    """eval_physical_expr with the column types at hand: + - * and unary - wrap at the operand type's width, integer / and % truncate
    towards zero; a zero divisor in a row whose operands are not NULL fails the call FOR EVERY TYPE -- Float64 0.0 / -0.0 included: arrow-rs's
    `divide` / `modulus` (math_checked_divide_op, divide_scalar, the simd feature's simd_checked_divide) test is_zero() on every native type and
    return ArrowError::DivideByZero (Arrow C++ / pyarrow answer +-inf / NaN there: the two Arrows differ, the fork runs arrow-rs); INT_MIN / -1
    and INT_MIN % -1 fail the call, too (Rust's `/` / `%` panic and the fork's release profile aborts on panic).  Other Float64 arithmetic is
    numpy's (IEEE), CAST is checked, CASE picks the first WHEN that is TRUE (every branch is evaluated for every row, as the fork's CaseExpr
    evaluates them over the whole batch).  `want`: the type an untyped literal takes.  (Assumptions A-V1..7 of flock_amd/csrc/valprog.hpp.)"""
    import numpy as np
    t = e["physical_expr"]
    if t in ("column", "literal"):
        v = eval_physical_expr(e, row)
        if t == "literal" and isinstance(v, int) and not isinstance(v, bool) and (static_type(e, types) or want) == "Float64":
            return float(v)
        return v
    if t in ("cast_expr", "try_cast_expr"):
        v = eval_typed(e["expr"], row, types)
        return None if v is None else _cast(v, _type_name(e["cast_type"]), t == "try_cast_expr")
    if t == "negative_expr":
        a = e.get("arg", e.get("expr"))
        v = eval_typed(a, row, types, want)
        return None if v is None else _wrap(-v, static_type(a, types) or want)
    if t == "not_expr":
        v = eval_typed(e.get("arg", e.get("expr")), row, types, "Boolean")
        return None if v is None else (not v)
    if t == "is_null_expr":
        return eval_typed(e.get("arg", e.get("expr")), row, types) is None
    if t == "is_not_null_expr":
        return eval_typed(e.get("arg", e.get("expr")), row, types) is not None
    if t == "in_list_expr":
        ty = static_type(e["expr"], types)
        v = eval_typed(e["expr"], row, types)
        if v is None:
            return None
        hit = any(v == eval_typed(x, row, types, ty) for x in e["list"])
        return (not hit) if e.get("negated") else hit
    if t == "case_expr":
        ty = static_type(e, types) or want
        out = eval_typed(e["else_expr"], row, types, ty) if e.get("else_expr") else None
        for w, th in reversed(e["when_then_expr"]):
            then = eval_typed(th, row, types, ty)
            if e.get("expr"):
                tb = static_type(e["expr"], types) or static_type(w, types)
                a, b = eval_typed(e["expr"], row, types, tb), eval_typed(w, row, types, tb)
                cond = None if a is None or b is None else a == b
            else:
                cond = eval_typed(w, row, types, "Boolean")
            if cond is True:
                out = then
        return out
    if t == "binary_expr":
        op = e["op"]
        if op in ("And", "Or"):
            a, b = eval_typed(e["left"], row, types, "Boolean"), eval_typed(e["right"], row, types, "Boolean")
            if op == "And":
                return False if (a is False or b is False) else (None if (a is None or b is None) else True)
            return True if (a is True or b is True) else (None if (a is None or b is None) else False)
        ty = static_type(e["left"], types) or static_type(e["right"], types) or (want if op in _ARITH else None)
        a, b = eval_typed(e["left"], row, types, ty), eval_typed(e["right"], row, types, ty)
        if a is None or b is None:
            return None
        if op not in _ARITH:
            return {"Eq": a == b, "NotEq": a != b, "Lt": a < b, "LtEq": a <= b, "Gt": a > b, "GtEq": a >= b}[op]
        if ty == "Float64" or isinstance(a, float) or isinstance(b, float):
            x, y = np.float64(a), np.float64(b)
            if op in ("Divide", "Modulo") and y == 0:
                raise ExprError("division by zero")
            with np.errstate(all="ignore"):
                return float({"Plus": x + y, "Minus": x - y, "Multiply": x * y, "Divide": x / y, "Modulo": np.fmod(x, y)}[op])
        if op in ("Divide", "Modulo"):
            if b == 0:
                raise ExprError("division by zero")
            if b == -1 and ty in _INT_RANGE and ty != "UInt64" and a == _INT_RANGE[ty][0]:
                raise ExprError("INT_MIN / -1 overflows")
            if op == "Modulo":
                return _wrap(_trunc_mod(a, b), ty)
            q = abs(a) // abs(b)
            return _wrap(q if (a >= 0) == (b >= 0) else -q, ty)
        return _wrap({"Plus": a + b, "Minus": a - b, "Multiply": a * b}[op], ty)
    raise ValueError("physical_expr " + t)


def filter_by_typed_expr(t: Table, pred: dict, types: dict) -> Table:
    """FilterExec through eval_typed."""
    return filter_exec(t, lambda r: eval_typed(pred, r, types, "Boolean"))


def project_typed(t: Table, exprs: Sequence[Tuple[dict, str]], types: dict) -> Table:
    """ProjectionExec over serialised expressions: [(expr, output name)]."""
    names = list(t)
    rs = [dict(zip(names, r)) for r in rows(t)]
    return {out: [eval_typed(e, r, types, static_type(e, types)) for r in rs] for e, out in exprs}


def filter_by_expr(t: Table, pred: dict) -> Table:
    """FilterExec with a serialised predicate: the rows for which it is TRUE."""
    return filter_exec(t, lambda r: eval_physical_expr(pred, r))


# -- ProjectionExec ---------------------------------------------------------------------------------
def projection_exec(t: Table, exprs: Sequence[Tuple[str, Callable[[dict], object]]]) -> Table:
    names = list(t)
    rs = [dict(zip(names, r)) for r in rows(t)]
    return {out: [f(r) for r in rs] for out, f in exprs}


# -- HashAggregateExec (Partial+Final collapsed: the split is result-neutral) ------------------------
def _agg_init(kind):
    return {"count": 0, "max": None, "min": None, "sum": None, "avg": (0, 0.0)}[kind]


def _agg_update(kind, st, v):
    if v is None:
        return st
    if kind == "count":
        return st + 1
    if kind == "max":
        return v if st is None or v > st else st
    if kind == "min":
        return v if st is None or v < st else st
    if kind == "sum":
        return v if st is None else st + v
    if kind == "avg":  # state = (UInt64 count, Float64 sum)
        return (st[0] + 1, st[1] + float(v))
    raise ValueError(kind)


def _agg_final(kind, st):
    if kind == "avg":
        return None if st[0] == 0 else st[1] / st[0]
    return st


def hash_aggregate_exec(t: Table, group_by: Sequence[str],
                        aggs: Sequence[Tuple[str, str, str | None]]) -> Table:
    """aggs = [(out_name, kind, input_col or None for COUNT(*))].  Groups come out in
    first-appearance order; an ungrouped aggregate over empty input yields one row."""
    names = list(t)
    groups: "OrderedDict[tuple, list]" = OrderedDict()
    if not group_by:
        groups[()] = [_agg_init(k) for _, k, _ in aggs]
    for r in rows(t):
        d = dict(zip(names, r))
        key = tuple(d[g] for g in group_by)
        st = groups.get(key)
        if st is None:
            st = groups[key] = [_agg_init(k) for _, k, _ in aggs]
        for i, (_, kind, col) in enumerate(aggs):
            st[i] = _agg_update(kind, st[i], 1 if col is None else d[col])
    out: Table = {g: [] for g in group_by}
    for name, _, _ in aggs:
        out[name] = []
    for key, st in groups.items():
        for g, v in zip(group_by, key):
            out[g].append(v)
        for (name, kind, _), s in zip(aggs, st):
            out[name].append(_agg_final(kind, s))
    return out


# -- HashJoinExec(Inner, Partitioned): build LEFT, probe RIGHT in row order, left cols ++ right cols --
def hash_join_inner(left: Table, right: Table, on: Sequence[Tuple[str, str]]) -> Table:
    ln, rn = list(left), list(right)
    build: Dict[tuple, list] = {}
    lrows = rows(left)
    for i, r in enumerate(lrows):
        d = dict(zip(ln, r))
        key = tuple(d[l] for l, _ in on)
        if any(k is None for k in key):
            continue
        build.setdefault(key, []).append(i)
    out: Table = {k: [] for k in ln + rn}
    for r in rows(right):
        d = dict(zip(rn, r))
        key = tuple(d[rc] for _, rc in on)
        for i in build.get(key, ()):
            for k, v in zip(ln, lrows[i]):
                out[k].append(v)
            for k, v in zip(rn, r):
                out[k].append(v)
    return out


# -- SortExec / GlobalLimitExec ------------------------------------------------------------------
def sort_exec(t: Table, by) -> Table:
    """by = [(column, descending[, nulls_first])], first key most significant; stable.  NULLs go where `nulls_first` says (arrow-rs
    SortOptions; DESC does not move them) and tie among themselves."""
    idx = list(range(num_rows(t)))
    for spec in reversed(list(by)):
        col, desc = spec[0], spec[1]
        nulls_first = spec[2] if len(spec) > 2 else False
        vals = [i for i in idx if t[col][i] is not None]
        nulls = [i for i in idx if t[col][i] is None]
        vals.sort(key=lambda i: t[col][i], reverse=desc)
        idx = nulls + vals if nulls_first else vals + nulls
    return {k: [v[i] for i in idx] for k, v in t.items()}


def window_row_number(t: Table, partition_by: Sequence[str], name: str = "ROW_NUMBER()") -> Table:
    """WindowAggExec with ROW_NUMBER() over an input that arrives sorted by (PARTITION BY, ORDER BY) -- the physical planner puts the SortExec
    underneath: the rows of every RUN of equal partition keys are numbered 1, 2, ... in arrival order (NULL keys equal each other); the window
    column comes FIRST (benchmarks/src/nexmark/query/q6_plan.fmt: the WindowAggr schemas)."""
    out, prev, k = [], object(), 0
    for r in range(num_rows(t)):
        key = tuple(t[c][r] for c in partition_by)
        k = k + 1 if (r > 0 and key == prev) else 1
        prev = key
        out.append(k)
    res: Table = {name: out}
    res.update(t)
    return res


def limit_exec(t: Table, n: int) -> Table:
    return {k: v[:n] for k, v in t.items()}


# -- CoalesceBatchesExec(target): buffer until >= target rows, then concat (transmute.rs:38-71) ------
def coalesce_batches(batches: Sequence[Table], target: int) -> List[Table]:
    out, buf, n = [], [], 0
    for b in batches:
        if num_rows(b) == 0:
            continue
        buf.append(b)
        n += num_rows(b)
        if n >= target:
            out.append(concat(buf))
            buf, n = [], 0
    if buf:
        out.append(concat(buf))
    return out


# -- RepartitionExec --------------------------------------------------------------------------------
def repartition_round_robin(partitions: Sequence[Sequence[Table]], n: int) -> List[List[Table]]:
    """RoundRobinBatch(n): every input partition deals its batches i % n (transmute.rs:74-109)."""
    out: List[List[Table]] = [[] for _ in range(n)]
    for part in partitions:
        for i, b in enumerate(part):
            out[i % n].append(b)
    return out


def repartition_hash(partitions: Sequence[Sequence[Table]], key: str, n: int,
                     hash_fn: Callable[[object], int] = hash) -> List[List[Table]]:
    """Hash(exprs, n): row -> hash(key) % n.  The destination of a key is an
    implementation detail (ahash-version dependent) and unobservable in query results;
    only the row multiset is a contract (transmute.rs:381-390)."""
    out: List[List[Table]] = [[] for _ in range(n)]
    for part in partitions:
        for b in part:
            dest = [hash_fn(v) % n for v in b[key]]
            for p in range(n):
                sel = [i for i, d in enumerate(dest) if d == p]
                if sel:
                    out[p].append({k: [v[i] for i in sel] for k, v in b.items()})
    return out


def repartition_hash_diff(t: Table, key: str, n: int) -> List[Table]:
    """HashDiff(exprs, n) -- the fork's own Partitioning variant (flock-function/src/aws/window/session.rs:236-253, global.rs:234; datasource/nexmark/
    queries/q6.rs:128, q11.rs:168): n is COUNT(DISTINCT key), counted by the host just before, and "each partition has a unique key after repartition
    execution".  The implementation lives in the absent DataFusion fork; what the call sites rely on -- and all this restates -- is: every distinct key
    a partition of its own, rows in input order inside it, which partition a key gets unobservable (the session code keys its map by the partition's
    first row, session.rs:262-280).  Here: partitions in ascending key order, empty ones behind them when n exceeds the distinct count."""
    keys = sorted(set(t[key]))
    if len(keys) > n:
        raise ValueError("HashDiff: more distinct keys than partitions")
    out: List[Table] = []
    for k in keys:
        sel = [i for i, v in enumerate(t[key]) if v == k]
        out.append({c: [v[i] for i in sel] for c, v in t.items()})
    return out + [{c: [] for c in t} for _ in range(n - len(keys))]


# -- the five NEXMark plans on top of the generic operators ------------------------------------------
def nexmark_q1(bid: Table) -> Table:
    return projection_exec(bid, [("auction", lambda r: r["auction"]), ("bidder", lambda r: r["bidder"]),
                                 ("price", lambda r: 0.908 * float(r["price"])),
                                 ("b_date_time", lambda r: r["b_date_time"])])


def nexmark_q2(bid: Table) -> Table:
    def trunc_mod(a, m):  # Rust / arrow `%` on i64: truncated remainder
        q = abs(a) // abs(m)
        q = q if (a >= 0) == (m >= 0) else -q
        return a - q * m
    f = filter_exec({"auction": bid["auction"], "price": bid["price"]}, lambda r: trunc_mod(r["auction"], 123) == 0)
    return projection_exec(f, [("auction", lambda r: r["auction"]), ("price", lambda r: r["price"])])


def nexmark_q3(auction: Table, person: Table) -> Table:
    a = filter_exec({k: auction[k] for k in ("a_id", "seller", "category")}, lambda r: r["category"] == 10)
    p = filter_exec({k: person[k] for k in ("p_id", "name", "city", "state")},
                    lambda r: r["state"] == "or" or r["state"] == "id" or r["state"] == "ca")
    j = hash_join_inner(a, p, [("seller", "p_id")])
    return {k: j[k] for k in ("name", "city", "state", "a_id")}


def nexmark_q5(bid: Table) -> Table:
    counts = hash_aggregate_exec({"auction": bid["auction"]}, ["auction"], [("num", "count", None)])
    maxn = hash_aggregate_exec({"num": counts["num"]}, [], [("maxn", "max", "num")])
    j = hash_join_inner(counts, maxn, [("num", "maxn")])
    return {"auction": j["auction"], "num": j["num"]}


def nexmark_q8(person: Table, auction: Table) -> Table:
    p = hash_aggregate_exec({k: person[k] for k in ("p_id", "name")}, ["p_id", "name"], [])
    a = hash_aggregate_exec({"seller": auction["seller"]}, ["seller"], [])
    j = hash_join_inner(p, a, [("p_id", "seller")])
    return {"p_id": j["p_id"], "name": j["name"]}


def nexmark_q6(auction: Table, bid: Table, last: int = 10) -> Table:
    """benchmarks/src/nexmark/query/q6.sql (q6_plan.fmt) operator by operator: the join with its BETWEEN, ROW_NUMBER by price per auction (= 1: the
    winning bid; ties in input order -- the reference's sort is not stable, so which of two equal top bids wins there is unspecified), ROW_NUMBER by
    b_date_time DESC per seller (<= `last`), AVG(price) per seller.  Rows in first-appearance order of the sellers."""
    j = hash_join_inner(auction, bid, [("a_id", "auction")])
    j = filter_exec(j, lambda r: r["a_date_time"] <= r["b_date_time"] <= r["expires"])
    w = window_row_number(sort_exec(j, [("a_id", False), ("price", True, True)]), ["a_id"], "price_rank")
    q = filter_exec(w, lambda r: r["price_rank"] == 1)
    q = sort_exec(q, [("a_id", False), ("price", True, True)])
    w2 = window_row_number(sort_exec(q, [("seller", False), ("b_date_time", True, True)]), ["seller"], "time_rank")
    r = filter_exec(w2, lambda x: x["time_rank"] <= last)
    return hash_aggregate_exec({"seller": r["seller"], "price": r["price"]}, ["seller"], [("AVG(R.price)", "avg", "price")])
