#!/bin/bash
# phase cut-outs of q3's window kernel (experimental library, FLOCKGPU_Q3W_MODE)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp flock_amd/libflockgpu.so /tmp/shipped.so
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
for round in 1 2; do for mode in 0 1 2 3; do
  FLOCKGPU_Q3W_MODE=$mode python bench.py --only-general q3_hash 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('mode $mode', d['ms_per_step'], r.get('kernels_ms'))"
done; done
cp /tmp/shipped.so flock_amd/libflockgpu.so
