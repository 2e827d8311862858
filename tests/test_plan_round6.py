"""Round-6 additions to the plan path (include/flockgpu_plan.h), each against the oracle:
  * comparisons whose right side is not a literal (`a % 3 = b`, `a = b % 3`, `a % 2 = b % 2`): the one-pass predicate program has no leaf
    for them and must hand them to the general evaluator instead of comparing against a literal that is not there;
  * the dense GROUP BY with four non-COUNT accumulators (72 KB of LDS per workgroup: above the 64 KB a launch gets without asking)."""
import json

import numpy as np
import pytest

from oracle import generic_ops as g
from test_plan_round5 import F, NAMES, _agg_plan, batches, binary, cast, col, lit, pyrows, scan, table
from test_plan_round5b import TYPES, norm


@pytest.fixture(scope="module")
def gpu():
    from flock_amd import GpuContext
    c = GpuContext(0)
    yield c
    c.close()


def _mod(e, m, ty="Int64"):
    return binary(e, "Modulo", lit(ty, m))


@pytest.mark.gpu
def test_a_remainder_or_a_column_against_a_non_literal_goes_to_the_general_evaluator(gpu):
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(606)
    t = table(9000, r, null_p=0.1)
    i64, j64 = cast(col("i"), "Int64"), cast(col("j"), "Int64")
    preds = [binary(_mod(j64, 7), "Eq", i64),                                   # a % 7 = b   (i is in [-40, 400))
             binary(i64, "Eq", _mod(j64, 401)),                                 # a = b % 401   (i is in [-40, 400))
             binary(_mod(i64, 2), "Eq", _mod(j64, 2)),                          # a % 2 = b % 2
             binary(_mod(col("i"), 5, "Int32"), "Lt", col("i")),                # a % 5 < a
             binary(col("l"), "GtEq", _mod(col("l"), 1000)),                    # l >= l % 1000
             binary(binary(_mod(i64, 3), "Eq", _mod(j64, 3)), "And", binary(col("i"), "Gt", lit("Int32", 10)))]
    for pred in preds:
        ctx = ExecutionContext([{"execution_plan": "filter_exec", "predicate": pred, "input": scan()}], gpu=gpu)
        gpu.profile_reset()
        gpu.profile(True)
        try:
            rb = collect(ctx, [[batches(t, 3000)]])[0][0]
            ran = gpu.profile_read()
        finally:
            gpu.profile(False)
            ctx.close()
        want = g.rows(g.filter_by_typed_expr(t, pred, TYPES))
        assert norm(pyrows(rb)) == norm(want) and 0 < len(want) < 9000, json.dumps(pred)
        assert "valprog_kernel" in ran and "pred_flag_kernel" not in ran, (json.dumps(pred), sorted(ran))


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["j", "l"])
def test_dense_group_by_with_four_accumulators(gpu, key):
    """SUM / MIN / MAX / SUM over a dense key: four 8-byte accumulator planes + the row counts = 72 KB of LDS per workgroup."""
    from flock_amd.runtime import ExecutionContext, collect
    r = np.random.default_rng(77)
    n = 60_000
    t = table(n, r, null_p=0.0)
    t[key] = [int(x) for x in (1000 + np.sort(r.integers(0, 5000, n)))]
    other = "l" if key == "j" else "j"
    aggs = [("sum", "i", "Int64"), ("min", other, "Int64" if other == "l" else "Int32"), ("max", "i", "Int32"), ("sum", other, "Int64")]   # (kMaxGroupAggs = 4 accumulators per GROUP BY)
    plan = _agg_plan(key, aggs)
    ctx = ExecutionContext([plan], gpu=gpu)
    gpu.profile_reset()
    gpu.profile(True)
    try:
        rb = collect(ctx, [[batches(t, 20_000)]])[0][0]
        ran = gpu.profile_read()
    finally:
        gpu.profile(False)
        ctx.close()
    assert "dense_group_kernel" in ran, sorted(ran)
    want = g.hash_aggregate_exec(t, [key], [("%s(%s)" % (fn.upper(), c or "UInt8(1)"), fn, c) for fn, c, _ in aggs])
    assert sorted(pyrows(rb)) == sorted(g.rows(want))
