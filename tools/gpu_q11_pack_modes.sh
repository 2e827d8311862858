#!/bin/bash
# where q11_pack_kernel's time goes: cut-outs with the experimental library (results are wrong under the modes; times only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp flock_amd/libflockgpu.so /tmp/shipped.so
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
for mode in 0 1 2 4 7; do
FLOCKGPU_Q11_PACK_MODE=$mode timeout 300 python bench.py --only-side q11 --steps 3 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('mode $mode', d.get('ms_per_step'), (r.get('kernels_ms_per_step') or {}).get('q11_pack_kernel'))"
done
cp /tmp/shipped.so flock_amd/libflockgpu.so
