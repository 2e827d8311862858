#!/usr/bin/env python3
"""Where a stage plan's `collect` spends its time: Python / Arrow C-interface glue vs the library calls (tools; GPU box)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, pyarrow as pa
from flock_amd import GpuContext, NEXMarkSource, Window
from flock_amd import runtime as R
from flock_amd.stages import StagedRun, build_query_dag

gpu = GpuContext(0)
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[key] = T.get(key, 0.0) + time.perf_counter() - t0
    setattr(obj, name, g)
lib = R._ffi.load()
for sym in ("flockgpu_plan_feed", "flockgpu_plan_execute", "flockgpu_plan_execute_partitioned", "flockgpu_plan_reset", "flockgpu_plan_input_matches"):
    wrap(lib, sym, "C:" + sym)
for q, seconds in ((3, 1), (8, 10)):
    plan = json.load(open(os.path.join(ROOT, "tests", "golden", "plans", f"q{q}.json")))
    g = NEXMarkSource(seconds, 1_000_000, Window.element_wise(), seed=11).generate_data(gpu)
    def utf8(u, n):
        off = u.offsets.cpu().numpy()[: n + 1]
        return pa.StringArray.from_buffers(n, pa.py_buffer(off.tobytes()), pa.py_buffer(u.data.cpu().numpy()[: int(off[-1])].tobytes()))
    a, p = g.auctions, g.persons
    rel = {"auction": pa.record_batch([pa.array(a.a_id.cpu().numpy()), pa.array(a.seller.cpu().numpy()), pa.array(a.category.cpu().numpy())], names=["a_id", "seller", "category"]),
           "person": pa.record_batch([pa.array(p.p_id.cpu().numpy()), utf8(p.name, p.rows), utf8(p.city, p.rows), utf8(p.state, p.rows)], names=["p_id", "name", "city", "state"])}
    if q == 8:
        rel = {"person": rel["person"], "auction": rel["auction"]}
    for inst in (0, 1):
      st = StagedRun(gpu, build_query_dag(plan), instances=inst)
      for _ in range(3):
          st.run(rel)
      T.clear()
      n = 10
      t0 = time.perf_counter()
      for _ in range(n):
          st.run(rel)
      total = time.perf_counter() - t0
      c = sum(v for k, v in T.items() if k.startswith("C:"))
      print(f"q{q} instances={inst}: {total / n * 1e3:.3f} ms per staged run; inside the library {c / n * 1e3:.3f} ms, Python / Arrow glue {(total - c) / n * 1e3:.3f} ms")
      for k, v in sorted(T.items()):
          print(f"   {k}: {v / n * 1e3:.3f} ms per run")
      st.close()
