#!/bin/bash
# gpurun helper: A/B of library variants on one box, alternating: LIBS="libflockgpu.so libflockgpu_experimental_q5v1.so ..." ARGS="--query 5"
# (experimental variants: FLOCKGPU_BUILD_EXPERIMENTAL=1 FLOCKGPU_BUILD_TAG=.. FLOCKGPU_BUILD_DEFINES=.. python -m flock_amd.build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
cp flock_amd/libflockgpu.so /tmp/shipped.so
for round in 1 2; do for lib in $LIBS; do
  src=flock_amd/$lib; [ "$lib" = "libflockgpu.so" ] && src=/tmp/shipped.so
  cp $src flock_amd/libflockgpu.so
  python bench.py ${ARGS:---query 5} --no-also --no-cpu --steps ${STEPS:-20} --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$lib', 'ms/step', d['ms_per_step'], 'kernel', r.get('kernel'), r.get('avg_launch_ms'), 'frac', r.get('frac'), 'rows', d['config'].get('result_rows'))"
done; done
cp /tmp/shipped.so flock_amd/libflockgpu.so
