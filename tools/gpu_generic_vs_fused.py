#!/usr/bin/env python3
"""gpurun helper: whole-plan q3 / q5 / q8 through the plan ABI, fused pipeline vs the generic operators (FLOCKGPU_PLAN_GENERIC_ONLY),
one window at `eps` events/s, with the generic run's per-kernel times: what an arbitrary plan of the same shape costs."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pyarrow as pa
from flock_amd import GpuContext, NEXMarkSource, Window
from flock_amd.runtime import ExecutionContext, collect

eps = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = 10
gpu = GpuContext(0)
for q, seconds in ((3, 1), (5, 10), (8, 10)):
    plan = json.load(open(os.path.join(ROOT, "tests", "golden", "plans", f"q{q}.json")))
    g = NEXMarkSource(seconds, eps, Window.element_wise(), seed=11).generate_data(gpu)

    def utf8(u, n):
        off = u.offsets.cpu().numpy()[: n + 1]
        return pa.StringArray.from_buffers(n, pa.py_buffer(off.tobytes()), pa.py_buffer(u.data.cpu().numpy()[: int(off[-1])].tobytes()))
    if q == 5:
        b = g.bids
        rel = [pa.record_batch([pa.array(b.auction.cpu().numpy()), pa.array(b.bidder.cpu().numpy()), pa.array(b.price.cpu().numpy()),
                                pa.array(b.b_date_time.cpu().numpy()).cast(pa.timestamp("ms"))], names=["auction", "bidder", "price", "b_date_time"])]
    else:
        a, p = g.auctions, g.persons
        ra = pa.record_batch([pa.array(a.a_id.cpu().numpy()), pa.array(a.seller.cpu().numpy()), pa.array(a.category.cpu().numpy())], names=["a_id", "seller", "category"])
        rp = pa.record_batch([pa.array(p.p_id.cpu().numpy()), utf8(p.name, p.rows), utf8(p.city, p.rows), utf8(p.state, p.rows)], names=["p_id", "name", "city", "state"])
        rel = [ra, rp] if q == 3 else [rp, ra]
    rows = sum(r.num_rows for r in rel)
    src = [[[rb]] for rb in rel]
    line = {"q": q, "rows": rows}
    for mode in ("fused", "generic"):
        ctx = ExecutionContext([plan], gpu=gpu, generic_only=(mode == "generic"))
        n = sum(b.num_rows for b in collect(ctx, src)[0])
        collect(ctx, src)
        gpu.profile_reset(); gpu.profile(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            collect(ctx, src)
        dt = (time.perf_counter() - t0) / steps
        gpu.profile(False)
        st = gpu.profile_read()
        top = sorted(((k, v["total_ms"] / steps) for k, v in st.items()), key=lambda kv: -kv[1])[:8]
        line[mode] = {"ms": round(dt * 1e3, 3), "result_rows": n, "kernel_ms": round(sum(v["total_ms"] for v in st.values()) / steps, 3),
                      "top": {k: round(v, 3) for k, v in top}}
        ctx.close()
    print(json.dumps(line))
