"""Computed expressions in general (flock_amd/csrc/valprog.hpp, round 5): ProjectionExec columns and FilterExec predicates built from
arithmetic (+ - * / %, unary -), CAST / TRY_CAST between the numeric types, comparisons of computed values, IN, IS [NOT] NULL and
CASE -- random expression trees over nullable Int32 / Int64 / Float64 columns, row for row against the oracle's typed evaluator
(oracle/generic_ops.py: eval_typed, the twin of the assumptions valprog.hpp states: wrapping integer arithmetic, truncating division,
a zero divisor or a CAST that does not fit fails the call, TRY_CAST yields NULL, Kleene logic, CASE picks the first TRUE WHEN)."""
import json
import math

import numpy as np
import pyarrow as pa
import pytest

from oracle import generic_ops as g
from test_plan_round5 import F, NAMES, batches, binary, cast, col, lit, pyrows, scan, table, unary

TYPES = {f["name"]: f["data_type"] for f in F}
COLS = {"Int32": ["i", "j"], "Int64": ["l"], "Float64": ["f"]}


@pytest.fixture(scope="module")
def gpu():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


def try_cast(e, t):
    return {"physical_expr": "try_cast_expr", "expr": e, "cast_type": t}


def case(whens, els=None, base=None):
    return {"physical_expr": "case_expr", "expr": base, "when_then_expr": [[w, t] for w, t in whens], "else_expr": els}


def rand_lit(r, ty, nonzero=False):
    if ty == "Float64":
        v = float(r.choice([0.5, -2.0, 3.25, 10.0, -0.1, 1e9]))
        return lit("Float64", v)
    v = int(r.choice([1, -1, 2, 3, 7, -13, 100, 65_537, 2**31 - 1] if ty == "Int32" else [1, -1, 2, 5, -9, 1000, 2**33 + 1, 2**62]))
    if not nonzero and r.random() < 0.15:
        v = 0
    return lit(ty, v)


def rand_value(r, ty, depth):
    """An expression of type `ty`; divisors are non-zero literals or columns wrapped in CASE WHEN x = 0 THEN 1 ELSE x END."""
    if depth == 0 or r.random() < 0.2:
        return col(str(r.choice(COLS[ty]))) if r.random() < 0.7 else rand_lit(r, ty)
    k = r.random()
    if k < 0.45:
        op = str(r.choice(["Plus", "Minus", "Multiply"]))
        return binary(rand_value(r, ty, depth - 1), op, rand_value(r, ty, depth - 1))
    if k < 0.6:
        op = str(r.choice(["Divide", "Modulo"]))
        if ty == "Float64" or r.random() < 0.5:
            d = rand_lit(r, ty, nonzero=True)
        else:
            c = col(str(r.choice(COLS[ty])))
            d = case([(binary(c, "Eq", lit(ty, 0)), lit(ty, 1))], c)
        return binary(rand_value(r, ty, depth - 1), op, d)
    if k < 0.7:
        return unary("negative_expr", rand_value(r, ty, depth - 1))
    if k < 0.85:   # a cast from another type: widening ones checked, narrowing ones TRY_CAST (NULL where the value does not fit)
        src = str(r.choice([t for t in COLS if t != ty]))
        inner = rand_value(r, src, depth - 1)
        widening = (src, ty) in (("Int32", "Int64"), ("Int32", "Float64"), ("Int64", "Float64"))
        return cast(inner, ty) if widening else try_cast(inner, ty)
    whens = [(rand_bool(r, depth - 1), rand_value(r, ty, depth - 1)) for _ in range(int(r.integers(1, 3)))]
    return case(whens, rand_value(r, ty, depth - 1) if r.random() < 0.6 else None)


def rand_bool(r, depth):
    k = r.random()
    if depth == 0 or k < 0.5:
        ty = str(r.choice(list(COLS)))
        op = str(r.choice(["Eq", "NotEq", "Lt", "LtEq", "Gt", "GtEq"]))
        return binary(rand_value(r, ty, max(depth - 1, 0)), op, rand_value(r, ty, max(depth - 1, 0)))
    if k < 0.6:
        return unary("is_null_expr" if r.random() < 0.5 else "is_not_null_expr", rand_value(r, str(r.choice(list(COLS))), depth - 1))
    if k < 0.7:
        ty = str(r.choice(["Int32", "Int64"]))
        return {"physical_expr": "in_list_expr", "expr": rand_value(r, ty, depth - 1), "negated": bool(r.random() < 0.4),
                "list": [rand_lit(r, ty) for _ in range(int(r.integers(1, 4)))]}
    if k < 0.8:
        return unary("not_expr", rand_bool(r, depth - 1))
    return binary(rand_bool(r, depth - 1), "And" if k < 0.9 else "Or", rand_bool(r, depth - 1))


def norm(rows):
    """NaN compares unequal to itself: rows with their NaNs named."""
    return [tuple("nan" if isinstance(v, float) and math.isnan(v) else v for v in row) for row in rows]


def out_field(name, ty):
    return {"data_type": ty, "dict_id": 0, "dict_is_ordered": False, "name": name, "nullable": True}


def projection(exprs):
    fields = [out_field(n, g.static_type(e, TYPES) or "Int64") for e, n in exprs]
    return {"execution_plan": "projection_exec", "expr": [[e, n] for e, n in exprs], "input": scan(), "schema": {"fields": fields, "metadata": {}}}


# ------------------------------------------------------------------ CPU: the oracle's typed evaluator on hand-worked rows
def test_oracle_typed_arithmetic_by_hand():
    row = {"i": 2**31 - 1, "j": -2**31, "l": -7, "f": 2.5, "s": None, "u": 2**64 - 1}
    ev = lambda e: g.eval_typed(e, row, TYPES)
    assert ev(binary(col("i"), "Plus", lit("Int32", 1))) == -2**31                      # wraps at the operand's width
    with pytest.raises(g.ExprError):
        ev(binary(col("j"), "Divide", lit("Int32", -1)))                                    # INT_MIN / -1: the reference's arithmetic panics (valprog.hpp A-V4)
    assert ev(binary(cast(col("i"), "Int64"), "Plus", lit("Int64", 1))) == 2**31        # ... and does not after the widening cast
    assert ev(binary(col("l"), "Divide", lit("Int64", 2))) == -3 and ev(binary(col("l"), "Modulo", lit("Int64", 2))) == -1   # truncation
    assert ev(binary(col("u"), "Plus", lit("UInt64", 2))) == 1
    for zero in (0.0, -0.0):   # a Float64 zero divisor is the same error (valprog.hpp A-V3: arrow-rs tests is_zero() for every native type)
        with pytest.raises(g.ExprError):
            ev(binary(col("f"), "Divide", lit("Float64", zero)))
    assert ev(binary(col("f"), "Divide", lit("Float64", math.inf))) == 0.0 and math.isnan(ev(binary(lit("Float64", math.inf), "Divide", lit("Float64", math.inf))))
    assert ev(binary(col("f"), "Modulo", lit("Float64", -2.0))) == 0.5
    with pytest.raises(g.ExprError):
        ev(binary(col("l"), "Divide", lit("Int64", 0)))
    assert g.eval_typed(binary(col("l"), "Divide", lit("Int64", 0)), dict(row, l=None), TYPES) is None   # a NULL row never looks at the divisor
    with pytest.raises(g.ExprError):
        ev(cast(binary(cast(col("i"), "Int64"), "Plus", lit("Int64", 1)), "Int32"))
    assert ev(try_cast(binary(cast(col("i"), "Int64"), "Plus", lit("Int64", 1)), "Int32")) is None
    assert ev(cast(binary(col("f"), "Multiply", lit("Float64", -1.0)), "Int64")) == -2 and ev(try_cast(lit("Float64", -0.9), "UInt64")) == 0
    c = case([(binary(col("l"), "Lt", lit("Int64", -10)), lit("Int64", 1)), (binary(col("l"), "Lt", lit("Int64", 0)), lit("Int64", 2))])
    assert ev(c) == 2 and g.eval_typed(c, dict(row, l=5), TYPES) is None and g.eval_typed(c, dict(row, l=None), TYPES) is None
    assert ev(case([(lit("Int64", -7), lit("Int32", 1))], lit("Int32", 0), base=col("l"))) == 1        # CASE l WHEN -7 THEN 1 ELSE 0
    assert g.static_type(binary(lit("Int64", 1), "Plus", col("l")), TYPES) == "Int64" and g.static_type(c, TYPES) == "Int64"


def test_expression_tags_of_the_general_evaluator_parse():
    from flock_amd import FlockGpuError
    from flock_amd.runtime import explain
    e1 = binary(binary(col("i"), "Plus", col("j")), "Divide", lit("Int32", 3))
    e2 = case([(binary(col("l"), "Gt", lit("Int64", 0)), cast(col("i"), "Float64"))], col("f"))
    txt = explain(projection([(col("j"), "j"), (e1, "x"), (e2, "y"), (try_cast(col("f"), "Int64"), "z")]))
    assert "Project" in txt
    with pytest.raises(FlockGpuError) as e:   # no Boolean columns at this boundary
        explain(projection([(binary(col("i"), "Lt", col("j")), "b")]))
    assert "Boolean" in str(e.value)
    with pytest.raises(FlockGpuError) as e:
        explain(projection([(binary(col("i"), "Plus", col("l")), "x")]))        # Int32 + Int64: the planner would have cast
    assert "numeric type" in str(e.value)


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_projections_and_filters_over_random_expressions(gpu, seed):
    from flock_amd import FlockGpuError, _ffi
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(7000 + seed)
    n = [1, 300, 4097, 20_000][seed % 4]
    t = table(n, r, null_p=[0.0, 0.2, 0.5][seed % 3])
    ran = 0

    def run(plan, chunk):   # None: the tree outgrew one program (96 operators, stack of 8) -- refused by name, never truncated
        ctx = ExecutionContext([plan], gpu=gpu)
        try:
            return collect(ctx, [[batches(t, chunk)]])[0][0]
        except FlockGpuError as e:
            assert e.code == _ffi.ERR_UNSUPPORTED and "expression too large" in str(e), str(e)
            return None
        finally:
            ctx.close()
    for trial in range(3):
        exprs = [(col("j"), "j")] + [(rand_value(r, str(r.choice(list(COLS))), 3), "x%d" % k) for k in range(int(r.integers(1, 4)))]
        rb = run(projection(exprs), max(1, n // 2))
        if rb is not None:
            ran += 1
            want = g.project_typed(t, exprs, TYPES)
            assert norm(pyrows(rb)) == norm(g.rows(want)), (seed, trial, json.dumps(exprs))
            assert [str(f.type) for f in rb.schema] == [{"Int32": "int32", "Int64": "int64", "Float64": "double"}[g.static_type(e, TYPES)] for e, _ in exprs]
        pred = rand_bool(r, 3)
        rb = run({"execution_plan": "filter_exec", "predicate": pred, "input": scan()}, max(1, n // 3))
        if rb is not None:
            ran += 1
            assert norm(pyrows(rb)) == norm(g.rows(g.filter_by_typed_expr(t, pred, TYPES))), (seed, trial, json.dumps(pred))
    assert ran >= 4, ran


@pytest.mark.gpu
def test_a_zero_divisor_or_an_unfit_cast_fails_the_call_and_try_cast_does_not(gpu):
    from flock_amd import FlockGpuError, _ffi
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(3)
    t = table(5000, r, null_p=0.3)
    t["i"][1234] = 0
    t["l"][77] = 2**35

    def run(plan):
        ctx = ExecutionContext([plan], gpu=gpu)
        try:
            return collect(ctx, [[batches(t, 2500)]])[0][0]
        finally:
            ctx.close()
    for bad, what in ((binary(col("j"), "Divide", col("i")), "division by zero"), (binary(col("l"), "Modulo", lit("Int64", 0)), "division by zero"),
                      (cast(col("l"), "Int32"), "does not fit")):
        with pytest.raises(FlockGpuError) as e:
            run(projection([(bad, "x")]))
        assert e.value.code == _ffi.ERR_INVALID and what in str(e.value), str(e.value)
    # the same divisor with its zero (and only its zero) made NULL: the row is NULL, the call stands
    t["i"][1234] = None
    e = binary(col("j"), "Divide", col("i"))
    exprs = [(e, "q"), (try_cast(col("l"), "Int32"), "n"), (binary(col("f"), "Divide", lit("Float64", 4.0)), "quarter")]
    if all(v != 0 for v in t["i"] if v is not None):
        assert norm(pyrows(run(projection(exprs)))) == norm(g.rows(g.project_typed(t, exprs, TYPES)))
    else:   # the generator drew another zero: the error it is
        with pytest.raises(FlockGpuError):
            run(projection(exprs))
    # Float64: a zero divisor -- 0.0 or -0.0 -- is the same error (valprog.hpp A-V3: arrow-rs tests is_zero() for floats, too); and INT_MIN / -1
    # fails the call (A-V4: the reference's arithmetic panics there)
    for bad, what in ((binary(col("f"), "Divide", lit("Float64", 0.0)), "division by zero"), (binary(col("f"), "Modulo", lit("Float64", -0.0)), "division by zero"),
                      (binary(cast(col("i"), "Float64"), "Divide", col("f")), "division by zero")):
        if bad["right"]["physical_expr"] == "column":
            t["f"][10], t["i"][10] = 0.0, 5
        with pytest.raises(FlockGpuError) as e2:
            run(projection([(bad, "x")]))
        assert e2.value.code == _ffi.ERR_INVALID and what in str(e2.value), str(e2.value)
    t["f"][10] = 1.0
    t["j"][3], t["l"][4] = -2**31, -2**63
    t["i"] = [-1 if v is not None else None for v in t["i"]]            # a column of -1 (and NULLs) as the divisor of the column / column case
    for bad in (binary(col("j"), "Divide", lit("Int32", -1)), binary(col("l"), "Modulo", lit("Int64", -1)), binary(col("j"), "Divide", col("i")),
                binary(col("l"), "Modulo", cast(col("i"), "Int64"))):
        t["i"][3] = t["i"][4] = -1
        with pytest.raises(FlockGpuError) as e3:
            run(projection([(bad, "x")]))
        assert e3.value.code == _ffi.ERR_INVALID and "overflows" in str(e3.value), str(e3.value)
    t["j"][3], t["l"][4] = -2**31 + 1, -2**63 + 1                         # one above the minimum: -x, no error
    exprs = [(binary(col("j"), "Divide", lit("Int32", -1)), "a"), (binary(col("l"), "Divide", cast(col("i"), "Int64")), "b"), (binary(col("l"), "Modulo", lit("Int64", -1)), "c")]
    assert norm(pyrows(run(projection(exprs)))) == norm(g.rows(g.project_typed(t, exprs, TYPES)))
    t = table(5000, r, null_p=0.3)
    # in a filter: rows an error sits in fail the call even when another conjunct would have dropped them (the whole batch is evaluated)
    with pytest.raises(FlockGpuError):
        run({"execution_plan": "filter_exec", "predicate": binary(binary(col("l"), "Divide", lit("Int64", 0)), "Gt", lit("Int64", 1)), "input": scan()})


@pytest.mark.gpu
def test_q1_conversion_keeps_its_own_kernel_and_other_products_take_the_evaluator(gpu):
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(5)
    t = table(3000, r, null_p=0.1)
    q1 = binary(lit("Float64", 0.908), "Multiply", cast(col("j"), "Float64"))
    other = binary(lit("Float64", 0.908), "Multiply", cast(col("l"), "Float64"))
    for e, kernel in ((q1, "q1_project_kernel"), (other, "valprog_kernel")):
        ctx = ExecutionContext([projection([(e, "x")])], gpu=gpu)
        gpu.profile_reset()
        gpu.profile(True)
        try:
            rb = collect(ctx, [[batches(t, 3000)]])[0][0]
            ran = gpu.profile_read()
        finally:
            gpu.profile(False)
            ctx.close()
        assert kernel in ran and ("valprog_kernel" in ran) == (kernel == "valprog_kernel"), sorted(ran)
        assert norm(pyrows(rb)) == norm(g.rows(g.project_typed(t, [(e, "x")], TYPES)))


@pytest.mark.gpu
def test_casts_that_change_values_are_not_looked_through_in_predicates(gpu):
    """The one-pass predicate program compares a column through casts that change nothing (Int32 -> Int64 / Float64); a cast that truncates
    (Float64 -> Int64), may not fit (TRY_CAST Int64 -> Int32: NULL) or changes signedness goes to the general evaluator with its value."""
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(11)
    t = table(6000, r, null_p=0.2)
    preds = [binary(cast(col("f"), "Int64"), "Eq", lit("Int64", 3)),                          # 3.0 .. 3.9 all match
             binary(try_cast(col("l"), "Int32"), "Lt", lit("Int32", 5)),                      # |l| >= 2^31: NULL, dropped
             unary("is_null_expr", try_cast(col("l"), "Int32")),                              # ... and IS NULL sees exactly those (and the NULLs)
             binary(try_cast(col("u"), "Int64"), "GtEq", lit("Int64", 0)),                    # UInt64 above 2^63 - 1: NULL
             binary(cast(col("i"), "Int64"), "Lt", lit("Int64", 7)),                          # (value-preserving: the one-pass program's)
             binary(binary(cast(col("i"), "Float64"), "Divide", lit("Float64", 4.0)), "Gt", lit("Float64", 10.1))]
    for pred in preds:
        ctx = ExecutionContext([{"execution_plan": "filter_exec", "predicate": pred, "input": scan()}], gpu=gpu)
        gpu.profile_reset()
        gpu.profile(True)
        try:
            rb = collect(ctx, [[batches(t, 2000)]])[0][0]
            ran = gpu.profile_read()
        finally:
            gpu.profile(False)
            ctx.close()
        want = g.rows(g.filter_by_typed_expr(t, pred, TYPES))
        assert norm(pyrows(rb)) == norm(want) and 0 < len(want) < 6000, json.dumps(pred)
        assert ("pred_flag_kernel" in ran) == (pred is preds[4]) and ("valprog_kernel" in ran) == (pred is not preds[4]), (json.dumps(pred), sorted(ran))


def _agg_over_expressions(key_expr, aggs):
    """GROUP BY <expression> with aggregates over expressions: Partial -> Hash -> FinalPartitioned, as the reference's planner lays it out
    (the Final stage reads the Partial stage's columns by position)."""
    def ae(fn, arg, dt, k):
        return {"aggregate_expr": fn, "name": "%s(#%d)" % (fn.upper(), k), "data_type": dt, "nullable": True, "expr": arg if arg is not None else lit("UInt8", 1)}
    exprs = [ae(fn, arg, dt, k) for k, (fn, arg, dt) in enumerate(aggs)]
    part = {"execution_plan": "hash_aggregate_exec", "mode": "Partial", "group_expr": [[key_expr, "k"]], "aggr_expr": exprs, "input": scan(),
            "input_schema": {"fields": F, "metadata": {}}, "schema": {"fields": [], "metadata": {}}}
    kc = {"physical_expr": "column", "name": "k", "index": 0}
    rep = {"execution_plan": "repartition_exec", "input": part, "partitioning": {"Hash": [[kc], 4]}}
    return {"execution_plan": "hash_aggregate_exec", "mode": "FinalPartitioned", "group_expr": [[kc, "k"]], "aggr_expr": exprs, "input": rep,
            "input_schema": {"fields": F, "metadata": {}}, "schema": {"fields": [], "metadata": {}}}


def test_aggregates_over_expressions_parse_with_a_projection_underneath():
    from flock_amd.runtime import explain
    txt = explain(_agg_over_expressions(binary(col("j"), "Modulo", lit("Int32", 10)), [("sum", binary(cast(col("i"), "Int64"), "Multiply", lit("Int64", 2)), "Int64")]))
    assert txt.count("Aggregate") == 2 and "Project" in txt, txt


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_group_by_and_aggregates_over_expressions(gpu, seed):
    """GROUP BY j % 10 / CASE ... with COUNT(*), COUNT(expr), SUM / MIN / MAX(expr): the expressions become columns of a projection under
    the Partial stage; groups equal the oracle's GROUP BY over its own projected columns (NULL keys form one group, NULL arguments are
    skipped)."""
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(400 + seed)
    n = [500, 9000, 40_000, 9000][seed]
    t = table(n, r, null_p=[0.0, 0.2, 0.4, 0.2][seed])
    key = [binary(col("j"), "Modulo", lit("Int32", 10)),
           binary(cast(col("i"), "Int64"), "Divide", lit("Int64", 7)),                                         # NULL where i is
           case([(binary(col("l"), "Lt", lit("Int64", 0)), lit("Int32", -1)), (binary(col("f"), "Gt", lit("Float64", 10.0)), lit("Int32", 1))], lit("Int32", 0)),
           binary(binary(col("j"), "Modulo", lit("Int32", 1000)), "Multiply", lit("Int32", 1_000_003))][seed]   # keys spread wide: the hash table
    aggs = [("count", None, "UInt64"), ("count", binary(col("i"), "Plus", lit("Int32", 1)), "UInt64"),
            ("sum", binary(cast(col("i"), "Int64"), "Multiply", lit("Int64", 3)), "Int64"),
            [("max", binary(col("l"), "Minus", cast(col("j"), "Int64")), "Int64"), ("min", unary("negative_expr", col("i")), "Int32")][seed % 2]]   # (four accumulators per GROUP BY)
    ctx = ExecutionContext([_agg_over_expressions(key, aggs)], gpu=gpu)
    rb = collect(ctx, [[batches(t, max(1, n // 2))]])[0][0]
    ctx.close()
    cols = g.project_typed(t, [(key, "k")] + [(a if a is not None else lit("Int32", 1), "a%d" % k) for k, (_, a, _) in enumerate(aggs)], TYPES)
    want = g.hash_aggregate_exec(cols, ["k"], [("o%d" % k, fn, None if a is None else "a%d" % k) for k, (fn, a, _) in enumerate(aggs)])
    assert sorted(norm(pyrows(rb)), key=repr) == sorted(norm(g.rows(want)), key=repr), (seed, len(rb), len(want["k"]))


@pytest.mark.gpu
def test_a_subquery_named_twice_runs_once(gpu):
    """q5's SQL names its COUNT(*) GROUP BY auction subquery in both join inputs (q5_plan.fmt:6,13), and only one of the plan's two `bid`
    leaves is ever fed: on the generic operators the aggregate sub-tree runs ONCE per execute (a memo keyed on the sub-tree's signature and
    the leaves its scans resolve to) and the result is the fused pipeline's."""
    import os
    from flock_amd import NEXMarkSource, Window
    from flock_amd.runtime import ExecutionContext, collect
    plan = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plans", "q5.json")))
    b = NEXMarkSource(4, 50_000, Window.element_wise(), seed=3).generate_data(gpu).bids
    rel = pa.record_batch([pa.array(b.auction.cpu().numpy()), pa.array(b.bidder.cpu().numpy()), pa.array(b.price.cpu().numpy()),
                           pa.array(b.b_date_time.cpu().numpy()).cast(pa.timestamp("ms"))], names=["auction", "bidder", "price", "b_date_time"])
    rows = {}
    for mode in ("generic", "fused"):
        ctx = ExecutionContext([plan], gpu=gpu, generic_only=(mode == "generic"))
        gpu.profile_reset()
        gpu.profile(True)
        try:
            rows[mode] = sorted(pyrows(collect(ctx, [[[rel]]])[0][0]))
            ran = gpu.profile_read()
        finally:
            gpu.profile(False)
            ctx.close()
        if mode == "generic":
            assert ran["dense_group_kernel"]["launches"] == 1, ran
    assert rows["generic"] == rows["fused"] and len(rows["fused"]) >= 1


def _sort_plan(keys):
    return {"execution_plan": "sort_exec", "input": scan(), "expr": [{"expr": e, "options": {"descending": bool(d), "nulls_first": bool(nf)}} for e, d, nf in keys]}


def test_order_by_an_expression_parses_with_projections_around_the_sort():
    from flock_amd.runtime import explain
    txt = explain(_sort_plan([(binary(cast(col("i"), "Int64"), "Plus", col("l")), True, True), (col("j"), False, False)]))
    lines = [l.strip() for l in txt.splitlines()]
    assert lines[0].startswith("Project") and lines[1].startswith("Sort(#") and lines[2].startswith("Project") and lines[0].count(":") == len(F), txt


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_order_by_expressions(gpu, seed):
    """ORDER BY <expression> [DESC], <column>: rows IN ORDER equal the oracle's stable sort over the expression's values (NULLs where the options put
    them); the output carries the input's columns only."""
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(90 + seed)
    t = table([40, 3000, 9000][seed], r, null_p=0.25)
    e1 = [binary(cast(col("i"), "Int64"), "Plus", col("l")), binary(col("f"), "Multiply", lit("Float64", -1.5)),
          case([(binary(col("i"), "Lt", lit("Int32", 100)), lit("Int32", 0))], binary(col("i"), "Modulo", lit("Int32", 7)))][seed]
    keys = [(e1, seed != 1, seed == 0), (col("j"), False, False)]
    ctx = ExecutionContext([_sort_plan(keys)], gpu=gpu)
    rb = collect(ctx, [[batches(t, 2000)]])[0][0]
    ctx.close()
    assert rb.schema.names == NAMES
    tmp = dict(t)
    tmp["#k"] = g.project_typed(t, [(e1, "#k")], TYPES)["#k"]
    want = g.sort_exec(tmp, [("#k", keys[0][1], keys[0][2]), ("j", False, False)])
    del want["#k"]
    assert norm(pyrows(rb)) == norm(g.rows(want))


@pytest.mark.gpu
def test_the_new_operators_on_an_empty_relation(gpu):
    """No rows in: computed projections, a predicate of the general evaluator, GROUP BY / aggregates over expressions, ORDER BY an expression and
    ROW_NUMBER() all hand back an empty batch of the right schema (an ungrouped aggregate is not among them: MAX over nothing is one NULL row)."""
    from flock_amd.runtime import ExecutionContext, collect
    t = {n: [] for n in NAMES}
    e = binary(cast(col("i"), "Int64"), "Plus", col("l"))
    srt = _sort_plan([(e, True, True)])
    plans = {
        "projection": (projection([(col("j"), "j"), (e, "x"), (case([(binary(col("f"), "Gt", lit("Float64", 0.0)), col("f"))]), "y")]), ["j", "x", "y"]),
        "filter": ({"execution_plan": "filter_exec", "predicate": binary(binary(col("i"), "Divide", lit("Int32", 3)), "Gt", lit("Int32", 1)), "input": scan()}, NAMES),
        "aggregate": (_agg_over_expressions(binary(col("j"), "Modulo", lit("Int32", 10)), [("count", None, "UInt64"), ("sum", e, "Int64")]), None),
        "sort": (srt, NAMES),
        "window": ({"execution_plan": "window_agg_exec", "input": srt, "window_expr": [{"fun": "RowNumber", "name": "rn", "partition_by": [col("i")], "order_by": []}]}, ["rn"] + NAMES),
    }
    for name, (plan, names) in plans.items():
        ctx = ExecutionContext([plan], gpu=gpu)
        for feed in ([[batches(t, 1)]], [[batches(t, 1)]]):     # twice: the second execute finds the first one's (empty) arenas
            out = collect(ctx, feed)[0]
            assert sum(b.num_rows for b in out) == 0, name
            if names is not None:
                assert out[0].schema.names == names, (name, out[0].schema.names)
        ctx.close()
