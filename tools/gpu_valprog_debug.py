#!/usr/bin/env python3
"""gpurun helper: a failing random predicate of tests/test_plan_round5b.py re-run with its two sides as projections, against the oracle."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_plan_round5b import *
from test_plan_round5 import table
from flock_amd import GpuContext
from flock_amd.runtime import ExecutionContext, collect
seed, bad_trial = int(sys.argv[1]), int(sys.argv[2])
gpu = GpuContext(0)
r = np.random.default_rng(7000 + seed)
n = [1, 300, 4097, 20_000][seed % 4]
t = table(n, r, null_p=[0.0, 0.2, 0.5][seed % 3])
for trial in range(3):
    exprs = [(col("j"), "j")] + [(rand_value(r, str(r.choice(list(COLS))), 3), "x%d" % k) for k in range(int(r.integers(1, 4)))]
    pred = rand_bool(r, 3)
    if trial != bad_trial:
        continue
    for chunk in (n, max(1, n // 3)):
        ctx = ExecutionContext([{"execution_plan": "filter_exec", "predicate": pred, "input": scan()}], gpu=gpu)
        rb = collect(ctx, [[batches(t, chunk)]])[0][0]
        ctx.close()
        got = norm(pyrows(rb)); want = norm(g.rows(g.filter_by_typed_expr(t, pred, TYPES)))
        print("filter chunk", chunk, len(got), len(want), got == want)
    L, R = pred["left"], pred["right"]
    variants = {"i != R": binary(col("i"), "NotEq", R), "L != j": binary(L, "NotEq", col("j")), "100 != R": binary(lit("Int32", 100), "NotEq", R),
                "R != 100": binary(R, "NotEq", lit("Int32", 100)), "R != L": binary(R, "NotEq", L),
                "L != (i-i)": binary(L, "NotEq", binary(col("i"), "Minus", col("i")))}
    for name, v in list(variants.items())[:2]:
        e = case([(v, lit("Int32", 1))], lit("Int32", 0))
        ctx = ExecutionContext([projection([(e, "x")])], gpu=gpu)
        rb = collect(ctx, [[batches(t, n)]])[0][0]
        ctx.close()
        got = norm(pyrows(rb)); want = norm(g.rows(g.project_typed(t, [(e, "x")], TYPES)))
        print("as a projection:", name, sum(x[0] for x in got), sum(x[0] for x in want), got == want)
        ctx = ExecutionContext([{"execution_plan": "filter_exec", "predicate": v, "input": {"execution_plan": "projection_exec", "expr": [[col(c), c] for c in ("i", "j", "l", "f")], "input": scan()}}], gpu=gpu)
        rb = collect(ctx, [[batches(t, n)]])[0][0]
        ctx.close()
        print("filter over a 4-column projection:", name, rb.num_rows)
    for name, v in variants.items():
        ctx = ExecutionContext([{"execution_plan": "filter_exec", "predicate": v, "input": scan()}], gpu=gpu)
        rb = collect(ctx, [[batches(t, n)]])[0][0]
        ctx.close()
        got = norm(pyrows(rb)); want = norm(g.rows(g.filter_by_typed_expr(t, v, TYPES)))
        print(name, len(got), len(want), got == want)
    if pred["physical_expr"] == "binary_expr":
        for side in ():
            e = pred[side]
            if g.static_type(e, TYPES) in ("Int32", "Int64", "Float64"):
                for chunk in (n, max(1, n // 3)):
                    ctx = ExecutionContext([projection([(e, "x")])], gpu=gpu)
                    rb = collect(ctx, [[batches(t, chunk)]])[0][0]
                    ctx.close()
                    got = norm(pyrows(rb)); want = norm(g.rows(g.project_typed(t, [(e, "x")], TYPES)))
                    bad = [k for k in range(len(want)) if got[k] != want[k]]
                    print(side, "chunk", chunk, "ok" if not bad else ("%d differ, first row %d: got %s want %s (i=%s l=%s f=%s)" % (len(bad), bad[0], got[bad[0]], want[bad[0]], t["i"][bad[0]], t["l"][bad[0]], t["f"][bad[0]])))
