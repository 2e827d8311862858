#!/bin/bash
# gpurun helper: the plan path's PCIe rows (bench.py --only-side plan_collect) with 2 / 4 / 8 staging lanes (experimental build's FLOCKGPU_STAGE_LANES)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
cp flock_amd/libflockgpu.so /tmp/shipped.so
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
for round in 1 2; do for lanes in ${LANES:-4 8 2}; do
  FLOCKGPU_STAGE_LANES=$lanes python bench.py --only-side plan_collect --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lanes $lanes', 'one instance ms', d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), {k:(v.get('ms_per_window'), v.get('pcie_frac')) for k,v in d.items() if isinstance(v,dict) and 'ms_per_window' in v})"
done; done
cp /tmp/shipped.so flock_amd/libflockgpu.so
