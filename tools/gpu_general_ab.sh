#!/bin/bash
# gpurun helper: one of bench.py's general-path rows (GENERAL=q5_uniform) under several library variants, twice each, alternating
# (LIBS="libflockgpu.so libflockgpu_experimental_x.so ..."; experimental variants: FLOCKGPU_BUILD_EXPERIMENTAL=1 FLOCKGPU_BUILD_TAG=..
# FLOCKGPU_BUILD_DEFINES=.. python -m flock_amd.build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
cp flock_amd/libflockgpu.so /tmp/shipped.so
for round in 1 2; do for lib in $LIBS; do
  src=flock_amd/$lib; [ "$lib" = "libflockgpu.so" ] && src=/tmp/shipped.so
  cp $src flock_amd/libflockgpu.so
  timeout 300 python bench.py --only-general ${GENERAL:-q5_uniform} --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$lib', 'ms/step', d['ms_per_step'], 'rows', d.get('result_rows'), r.get('kernels_ms'))"
done; done
cp /tmp/shipped.so flock_amd/libflockgpu.so
