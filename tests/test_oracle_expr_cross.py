"""The oracle's typed expression evaluator (oracle/generic_ops.py: eval_typed -- the twin of the HIP path's general evaluator, valprog.hpp) against
Arrow C++ (pyarrow.compute), expression by expression on random nullable columns: wrapping + - * and unary -, truncating integer division and its
zero-divisor error, IEEE Float64 arithmetic, comparisons, Kleene AND / OR / NOT, IS [NOT] NULL, CASE (case_when), the value-preserving casts.  Arrow
C++ is not the reference's arrow-rs, but the two implement one specification; the assumptions valprog.hpp lists are upstream-Arrow semantics and this
is where they meet an Arrow implementation.  (Left out because Arrow C++ differs or lacks the kernel: integer `%`, Float64 -> integer casts -- C++
refuses to truncate where arrow-rs truncates --, TRY_CAST.)"""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from oracle import generic_ops as g

TYPES = {"i": "Int32", "j": "Int32", "l": "Int64", "m": "Int64", "f": "Float64", "h": "Float64"}
PA = {"Int32": pa.int32(), "Int64": pa.int64(), "Float64": pa.float64()}
COLS = {"Int32": ["i", "j"], "Int64": ["l", "m"], "Float64": ["f", "h"]}


def col(n):
    return {"physical_expr": "column", "name": n}


def lit(k, v):
    return {"physical_expr": "literal", "value": {k: v}}


def binary(a, op, b):
    return {"physical_expr": "binary_expr", "left": a, "op": op, "right": b}


def table(n, r):
    nul = lambda xs: [None if r.random() < 0.2 else x for x in xs]
    return {"i": nul([int(x) for x in r.integers(-2**31, 2**31 - 1, n)]), "j": nul([int(x) for x in r.integers(-50, 50, n)]),
            "l": nul([int(x) for x in r.integers(-2**63, 2**63 - 1, n)]), "m": nul([int(x) for x in r.integers(-9, 9, n)]),
            "f": nul([float(x) for x in r.normal(0, 1e3, n)]), "h": nul([float(x) for x in r.choice([0.25, -0.5, 1.5, -2.25, 1e300, np.inf], n)])}   # (no zeroes: a Float64 zero divisor is an error, below)


def rand_value(r, ty, depth):
    if depth == 0 or r.random() < 0.25:
        if r.random() < 0.75:
            return col(str(r.choice(COLS[ty])))
        return lit(ty, float(r.choice([0.5, -3.0, 1e10])) if ty == "Float64" else int(r.choice([0, 1, -1, 7, 2**31 - 1] if ty == "Int32" else [0, 1, -1, 9, 2**62])))
    k = r.random()
    if k < 0.5:
        return binary(rand_value(r, ty, depth - 1), str(r.choice(["Plus", "Minus", "Multiply"])), rand_value(r, ty, depth - 1))
    if k < 0.62:
        return binary(rand_value(r, ty, depth - 1), "Divide", rand_value(r, ty, depth - 1))
    if k < 0.72:
        return {"physical_expr": "negative_expr", "arg": rand_value(r, ty, depth - 1)}
    if k < 0.82 and ty != "Int32":   # widening casts only (Int32 -> Int64 / Float64, Int64 -> Float64)
        src = "Int32" if ty == "Int64" else str(r.choice(["Int32", "Int64"]))
        return {"physical_expr": "cast_expr", "expr": rand_value(r, src, depth - 1), "cast_type": ty}
    whens = [[rand_bool(r, depth - 1), rand_value(r, ty, depth - 1)] for _ in range(int(r.integers(1, 3)))]
    return {"physical_expr": "case_expr", "expr": None, "when_then_expr": whens, "else_expr": rand_value(r, ty, depth - 1) if r.random() < 0.6 else None}


def rand_bool(r, depth):
    k = r.random()
    if depth == 0 or k < 0.5:
        ty = str(r.choice(list(COLS)))
        return binary(rand_value(r, ty, max(depth - 1, 0)), str(r.choice(["Eq", "NotEq", "Lt", "LtEq", "Gt", "GtEq"])), rand_value(r, ty, max(depth - 1, 0)))
    if k < 0.65:
        return {"physical_expr": "is_null_expr" if r.random() < 0.5 else "is_not_null_expr", "arg": rand_value(r, str(r.choice(list(COLS))), depth - 1)}
    if k < 0.75:
        return {"physical_expr": "not_expr", "arg": rand_bool(r, depth - 1)}
    return binary(rand_bool(r, depth - 1), "And" if k < 0.88 else "Or", rand_bool(r, depth - 1))


class DivideByZero(Exception):
    pass


def to_arrow(e, t):
    """The expression over pyarrow arrays (types as eval_typed infers them)."""
    k = e["physical_expr"]
    if k == "column":
        return t[e["name"]]
    if k == "literal":
        (kind, v), = e["value"].items()
        return pa.scalar(v, PA[kind])
    if k == "cast_expr":
        return pc.cast(to_arrow(e["expr"], t), PA[e["cast_type"]], safe=False)   # (widening only: Int64 -> Float64 rounds to nearest, which the safe cast refuses)
    if k == "negative_expr":
        return pc.negate(to_arrow(e["arg"], t))
    if k == "not_expr":
        return pc.invert(to_arrow(e["arg"], t))
    if k == "is_null_expr":
        return pc.is_null(to_arrow(e["arg"], t))
    if k == "is_not_null_expr":
        return pc.is_valid(to_arrow(e["arg"], t))
    if k == "case_expr":
        conds = [to_arrow(w, t) for w, _ in e["when_then_expr"]]
        thens = [to_arrow(th, t) for _, th in e["when_then_expr"]]
        n = len(next(iter(t.values())))
        full = lambda x: x if isinstance(x, (pa.Array, pa.ChunkedArray)) else pa.array([x.as_py()] * n, x.type)
        ty = next(x.type for x in thens)
        args = [full(x) for x in thens] + [full(to_arrow(e["else_expr"], t)) if e.get("else_expr") else pa.nulls(n, ty)]
        return pc.case_when(pc.make_struct(*[pc.fill_null(full(c), False) for c in conds]), *args)     # a NULL WHEN does not match
    a, b = to_arrow(e["left"], t), to_arrow(e["right"], t)
    op = e["op"]
    if op == "Divide":
        n = len(next(iter(t.values())))
        bb = b if isinstance(b, (pa.Array, pa.ChunkedArray)) else pa.array([b.as_py()] * n, b.type)
        aa = a if isinstance(a, (pa.Array, pa.ChunkedArray)) else pa.array([a.as_py()] * n, a.type)
        both = pc.and_(pc.is_valid(aa), pc.is_valid(bb))
        # a zero divisor in a valid row fails the call for EVERY type (arrow-rs: math_checked_divide_op; Arrow C++ would answer +-inf / NaN for
        # floats -- the one place the two Arrows differ here), and so does INT_MIN / -1 (a panic in arrow-rs, an abort of the fork's release build)
        if pc.any(pc.and_(both, pc.equal(pc.fill_null(bb, 1), 0))).as_py():
            raise DivideByZero()
        if pa.types.is_integer(a.type):
            lo = pa.scalar(-2**31 if a.type == pa.int32() else -2**63, a.type)
            if pc.any(pc.and_(both, pc.and_(pc.equal(pc.fill_null(aa, 0), lo), pc.equal(pc.fill_null(bb, 1), -1)))).as_py():
                raise DivideByZero()
        return pc.divide(aa, bb)
    fn = {"Plus": pc.add, "Minus": pc.subtract, "Multiply": pc.multiply, "Divide": pc.divide, "Eq": pc.equal, "NotEq": pc.not_equal, "Lt": pc.less,
          "LtEq": pc.less_equal, "Gt": pc.greater, "GtEq": pc.greater_equal, "And": pc.and_kleene, "Or": pc.or_kleene}[op]
    return fn(a, b)


def same(x, y):
    if x is None or y is None:
        return x is None and y is None
    if isinstance(x, float) and isinstance(y, float):
        return (np.isnan(x) and np.isnan(y)) or (x == y and np.signbit(x) == np.signbit(y))
    return x == y and type(x) is type(y)


@pytest.mark.parametrize("seed", range(12))
def test_eval_typed_equals_arrow_compute(seed):
    r = np.random.default_rng(500 + seed)
    t = table(400, r)
    at = {k: pa.array(v, PA[TYPES[k]]) for k, v in t.items()}
    names = list(t)
    rows = [dict(zip(names, vals)) for vals in zip(*t.values())]
    checked = 0
    for trial in range(25):
        e = rand_bool(r, 3) if trial % 3 == 0 else rand_value(r, str(r.choice(list(COLS))), 3)
        try:
            want = to_arrow(e, at)
        except DivideByZero:
            with pytest.raises(g.ExprError):
                [g.eval_typed(e, row, TYPES) for row in rows]
            continue
        want = want.to_pylist() if isinstance(want, (pa.Array, pa.ChunkedArray)) else [want.as_py()] * len(rows)
        got = [g.eval_typed(e, row, TYPES, g.static_type(e, TYPES)) for row in rows]
        bad = [i for i in range(len(rows)) if not same(got[i], want[i])]
        assert not bad, (seed, trial, e, rows[bad[0]], got[bad[0]], want[bad[0]])
        checked += 1
    assert checked >= 15


def test_a_float64_zero_divisor_fails_the_call_where_arrow_cpp_answers_infinity():
    """Assumption A-V3 of valprog.hpp, stated as a test: arrow-rs's divide / modulus check is_zero() for floats as well (ArrowError::DivideByZero),
    Arrow C++ follows IEEE.  The oracle (and the HIP evaluator, tests/test_plan_round5b.py) take arrow-rs's side; a NULL divisor or dividend
    makes the row NULL before the divisor is looked at."""
    e = binary(col("f"), "Divide", col("h"))
    for zero in (0.0, -0.0):
        with pytest.raises(g.ExprError):
            g.eval_typed(e, {"f": 1.5, "h": zero}, TYPES)
        with pytest.raises(g.ExprError):
            g.eval_typed(binary(col("f"), "Modulo", col("h")), {"f": 1.5, "h": zero}, TYPES)
        assert g.eval_typed(e, {"f": None, "h": zero}, TYPES) is None
        assert pc.divide(pa.array([1.5]), pa.array([zero])).to_pylist()[0] in (np.inf, -np.inf)
    assert g.eval_typed(e, {"f": 3.0, "h": None}, TYPES) is None
    with pytest.raises(g.ExprError):
        g.eval_typed(binary(col("i"), "Divide", col("j")), {"i": -2**31, "j": -1}, TYPES)
    with pytest.raises(g.ExprError):
        g.eval_typed(binary(col("l"), "Modulo", col("m")), {"l": -2**63, "m": -1}, TYPES)
    assert g.eval_typed(binary(col("i"), "Divide", col("j")), {"i": -2**31 + 1, "j": -1}, TYPES) == 2**31 - 1
