#!/usr/bin/env python3
"""gpurun helper: one whole-query plan (q = argv[1]) on the GENERIC operators, fed once and executed `steps` times with the result left in
HBM -- run under `rocprofv3 --hip-trace --kernel-trace --stats` to see how many launches, copies and host waits one execute of an
arbitrary plan of that shape costs (tools/gpu_generic_api_counts.sh)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyarrow as pa
from flock_amd import GpuContext, NEXMarkSource, Window
from flock_amd.runtime import ExecutionContext

q = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mode = sys.argv[3] if len(sys.argv) > 3 else "generic"
gpu = GpuContext(0)
plan = json.load(open(os.path.join(ROOT, "tests", "golden", "plans", f"q{q}.json")))
g = NEXMarkSource(1 if q == 3 else 10, 1_000_000, Window.element_wise(), seed=11).generate_data(gpu)


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
ctx = ExecutionContext([plan], gpu=gpu, generic_only=(mode == "generic"))
ctx.feed_data_sources([[[rb]] for rb in rel])
ctx.plans[0].execute_retain()
gpu.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ctx.plans[0].execute_retain()
    gpu.synchronize()
print(json.dumps({"q": q, "mode": mode, "steps": steps, "execute_only_ms": round((time.perf_counter() - t0) / steps * 1e3, 4)}))
