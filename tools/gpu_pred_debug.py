"""Debug aid: re-runs failing seeds of tests/test_plan_round5.py::test_predicates_at_random and prints where the kept rows differ."""
import json, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import test_plan_round5 as T
from oracle import generic_ops as g
from flock_amd import GpuContext
from flock_amd.runtime import ExecutionContext, collect

gpu = GpuContext(0)
for seed in [int(x) for x in sys.argv[1:]] or range(24):
    r = np.random.default_rng(9000 + seed)
    n = [1, 700, 8192, 8193, 30_000, 16_384][seed % 6]
    t = T.table(n, r, null_p=[0.0, 0.15, 0.5][seed % 3])
    for trial in range(3):
        pred = T.random_pred(r, 3)
        plan = {"execution_plan": "filter_exec", "predicate": pred, "input": T.scan()}
        ctx = ExecutionContext([plan], gpu=gpu)
        rb = collect(ctx, [[T.batches(t, max(1, n // 3))]])[0][0]
        ctx.close()
        got, want = T.pyrows(rb), g.rows(g.filter_by_expr(t, pred))
        if got == want:
            continue
        print("SEED", seed, "trial", trial, "n", n, "got", len(got), "want", len(want))
        print(json.dumps(pred))
        allrows = g.rows(t)
        gs, ws = set(got), set(want)
        extra = [i for i, row in enumerate(allrows) if row in gs and row not in ws][:5]
        missing = [i for i, row in enumerate(allrows) if row in ws and row not in gs][:5]
        print(" extra rows", [(i, allrows[i]) for i in extra])
        print(" missing rows", [(i, allrows[i]) for i in missing])
        # leaf by leaf
        def leaves(e):
            if e["physical_expr"] == "binary_expr" and e["op"] in ("And", "Or"):
                return leaves(e["left"]) + leaves(e["right"])
            if e["physical_expr"] == "not_expr":
                return leaves(e["arg"])
            return [e]
        for lf in leaves(pred):
            c2 = ExecutionContext([{"execution_plan": "filter_exec", "predicate": lf, "input": T.scan()}], gpu=gpu)
            rb2 = collect(c2, [[T.batches(t, max(1, n // 3))]])[0][0]
            c2.close()
            ok = T.pyrows(rb2) == g.rows(g.filter_by_expr(t, lf))
            print("  leaf", "OK " if ok else "BAD", json.dumps(lf), rb2.num_rows, len(g.rows(g.filter_by_expr(t, lf))))
