#!/bin/bash
# gpurun helper: the plan path's PCIe-inclusive collect (bench.py: plan_collect_pcie) on its own and after the other side entries of the
# default bench (it measured 1.96 ms alone and 4.8 ms at the end of the default run: which predecessor changes the process?)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
python - <<'PY'
import os, json, time
import bench
from flock_amd import GpuContext
ctx = GpuContext(0)
def m(tag):
    e = bench.plan_collect_pcie(ctx, 1_000_000, 10)
    print(tag, e["ms_per_step"], e["roofline"]["achieved"], "GB/s", flush=True)
m("fresh")
for name, fn in (("pcie_inclusive_q5", lambda: bench.pcie_inclusive_q5(ctx, 1_000_000)), ("ysb", lambda: bench.ysb_side(ctx, 1_000_000, 3, True, None)),
                 ("payload", lambda: bench.payload_side(ctx, 3, True)), ("json", lambda: bench.json_side(ctx, 3, True)),
                 ("q11", lambda: bench.q11_side(ctx, 1_000_000, 3, True)),
                 ("q5 entry + cpu baseline", lambda: bench.entry_for(ctx, 5, 100, 1_000_000, 3, 1, False, None))):
    t = time.time(); fn(); print("   ran", name, round(time.time() - t, 1), "s", flush=True)
    m("after " + name)
PY
