#!/usr/bin/env python3
"""gpurun helper: q3's stage plans run with the stage boundary in HBM (StagedRun on_device=True, as bench.py's plan_stages), `steps` runs --
under `rocprofv3 --hip-trace --kernel-trace --stats`: launches, copies and host waits per run of the staged query (argv: q steps mode)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyarrow as pa
from flock_amd import GpuContext, NEXMarkSource, Window
from flock_amd.runtime import ExecutionContext, collect
from flock_amd.stages import StagedRun, build_query_dag

q = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mode = sys.argv[3] if len(sys.argv) > 3 else "on_device"
gpu = GpuContext(0)
plan = json.load(open(os.path.join(ROOT, "tests", "golden", "plans", f"q{q}.json")))
g = NEXMarkSource(1 if q == 3 else 10, 1_000_000, Window.element_wise(), seed=11).generate_data(gpu)


def utf8(u, n):
    off = u.offsets.cpu().numpy()[: n + 1]
    return pa.StringArray.from_buffers(n, pa.py_buffer(off.tobytes()), pa.py_buffer(u.data.cpu().numpy()[: int(off[-1])].tobytes()))


if q == 5:
    b = g.bids
    rel = {"bid": pa.record_batch([pa.array(b.auction.cpu().numpy()), pa.array(b.bidder.cpu().numpy()), pa.array(b.price.cpu().numpy()),
                                   pa.array(b.b_date_time.cpu().numpy()).cast(pa.timestamp("ms"))], names=["auction", "bidder", "price", "b_date_time"])}
else:
    a, p = g.auctions, g.persons
    rel = {"auction": pa.record_batch([pa.array(a.a_id.cpu().numpy()), pa.array(a.seller.cpu().numpy()), pa.array(a.category.cpu().numpy())], names=["a_id", "seller", "category"]),
           "person": pa.record_batch([pa.array(p.p_id.cpu().numpy()), utf8(p.name, p.rows), utf8(p.city, p.rows), utf8(p.state, p.rows)], names=["p_id", "name", "city", "state"])}
    if q == 8:
        rel = {"person": rel["person"], "auction": rel["auction"]}
if mode == "whole":
    ctx = ExecutionContext([plan], gpu=gpu)
    src = [[[rb]] for rb in rel.values()]
    run = lambda: collect(ctx, src)
else:
    dev = StagedRun(gpu, build_query_dag(plan), share_sources=True, on_device=True)
    run = lambda: dev.run(rel)
run(); run()
t0 = time.perf_counter()
for _ in range(steps):
    run()
print(json.dumps({"q": q, "mode": mode, "ms_per_run": round((time.perf_counter() - t0) / steps * 1e3, 4)}))
