#!/bin/bash
# q5_count_kernel as num_cus x N persistent workgroups walking the tiles (A/B build -DFLOCKGPU_AB_Q5_PERSIST) against one workgroup per tile
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp flock_amd/libflockgpu.so /tmp/shipped.so
show='import json,sys
d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}
print(sys.argv[1], "ms/step", d["ms_per_step"], "count", r.get("avg_launch_ms"), "frac", r.get("frac"), "rows", d["config"].get("result_rows"))'
for round in 1 2; do
  cp /tmp/shipped.so flock_amd/libflockgpu.so
  python bench.py --query 5 --no-also --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "$show" shipped
  cp flock_amd/libflockgpu_experimental_persist.so flock_amd/libflockgpu.so
  for per in 4 6 8 9 16; do
    FLOCKGPU_Q5_PERSIST_PER_CU=$per python bench.py --query 5 --no-also --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "$show" persist_$per
  done
done
cp /tmp/shipped.so flock_amd/libflockgpu.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "q5" -x -p no:cacheprovider 2>&1 | tail -2
