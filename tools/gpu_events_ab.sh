#!/bin/bash
# gpurun helper: the default bench's side rows (q2 / q8 / q7 after the q5 headline, 20 steps each) with LaunchScope's events bound to the dispatch
# (shipped) and recorded either side of the launch (FLOCKGPU_AB_RECORDED_EVENTS=1, experimental build), alternating on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
cp flock_amd/libflockgpu.so /tmp/shipped.so
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
for round in 1 2; do for mode in bound recorded; do
  if [ $mode = recorded ]; then export FLOCKGPU_AB_RECORDED_EVENTS=1; else unset FLOCKGPU_AB_RECORDED_EVENTS; fi
  python - <<PY
import sys, json
sys.path.insert(0, ".")
import bench
from flock_amd import GpuContext
ctx = GpuContext(0)
out = {}
for q, secs in ((5, bench.DEFAULT_SECONDS[5]), (2, bench.DEFAULT_SECONDS[2]), (8, bench.DEFAULT_SECONDS[8]), (7, bench.DEFAULT_SECONDS[7])):
    e = bench.entry_for(ctx, q, secs, 1_000_000, 20, 2, True, 1)
    r = e["roofline"]
    out[q] = (e["ms_per_step"], r["frac"], r.get("avg_launch_ms"))
print("$mode", json.dumps(out))
PY
done; done
cp /tmp/shipped.so flock_amd/libflockgpu.so
