#!/usr/bin/env python3
"""Per `collect` of a one-instance staged run: wall time and the kernels it launched (tools; GPU box)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, pyarrow as pa
from flock_amd import GpuContext, NEXMarkSource, Window
from flock_amd import runtime as R
from flock_amd.stages import StagedRun, build_query_dag

gpu = GpuContext(0)
ON_DEVICE = os.environ.get("TRACE_ON_DEVICE", "1") == "1"
orig_collect = R.collect
log = []
def traced(ctx, src):
    gpu.profile_reset(); gpu.profile(True)
    t0 = time.perf_counter()
    out = orig_collect(ctx, src)
    dt = time.perf_counter() - t0
    st = gpu.profile_read(); gpu.profile(False)
    log.append((ctx.name if hasattr(ctx, "name") else "?", dt, st))
    return out
for q, seconds in ((3, 1), (8, 10), (5, 10)):
    plan = json.load(open(os.path.join(ROOT, "tests", "golden", "plans", f"q{q}.json")))
    g = NEXMarkSource(seconds, 1_000_000, Window.element_wise(), seed=11).generate_data(gpu)
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
    st = StagedRun(gpu, build_query_dag(plan), instances=1, share_sources=True, on_device=ON_DEVICE)
    for _ in range(3):
        st.run(rel)
    import flock_amd.stages as S
    R.collect = traced
    log.clear()
    if ON_DEVICE:   # one bracket per stage: execute_retain / execute of every stage context
        for ctx in st.ctxs:
            for name in ("execute_retain", "execute"):
                f = getattr(ctx, name)
                def g(f=f, ctx=ctx, name=name):
                    gpu.profile_reset(); gpu.profile(True)
                    t0 = time.perf_counter()
                    out = f()
                    dt = time.perf_counter() - t0
                    stt = gpu.profile_read(); gpu.profile(False)
                    log.append((ctx.name + "." + name, dt, stt))
                    return out
                setattr(ctx, name, g)
    st.run(rel)
    R.collect = orig_collect
    print(f"== q{q}")
    for name, dt, stats in log:
        n = sum(v["launches"] for v in stats.values())
        ms = sum(v["total_ms"] for v in stats.values())
        print(f"  {name}: {dt * 1e3:.3f} ms wall (profiled), {n} launches, {ms:.3f} ms of kernels")
        print("     " + ", ".join(f"{k.replace('_kernel', '')} x{v['launches']}" for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])))
    st.close()
