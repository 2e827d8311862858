#!/bin/bash
# valprog_kernel: rows per pass A/B (experimental build, FLOCKGPU_VALPROG_GROUPS = 1 | 2 | 4)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp flock_amd/libflockgpu.so /tmp/shipped.so
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
for round in 1 2; do for g in 1 2 4; do
  FLOCKGPU_VALPROG_GROUPS=$g python bench.py --only-side expr --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('expr_project','expr_filter'):
    e=d.get(k,{}); print('groups $g', k, e.get('ms_per_step'), (e.get('roofline') or {}).get('avg_launch_ms'), (e.get('roofline') or {}).get('frac'), e.get('result_rows'), e.get('error'))"
done; done
cp /tmp/shipped.so flock_amd/libflockgpu.so
