#!/bin/bash
# gpurun helper: A/B of (library variant, environment) pairs on one box, alternating.  CASES="name|lib|ENV=1 ENV2=2;name2|lib2|" ARGS="--query 3"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
cp flock_amd/libflockgpu.so /tmp/shipped.so
IFS=';' read -ra cases <<< "$CASES"
for round in 1 2; do for cs in "${cases[@]}"; do
  IFS='|' read -r name lib envs <<< "$cs"
  src=flock_amd/$lib; [ "$lib" = "libflockgpu.so" ] && src=/tmp/shipped.so
  cp $src flock_amd/libflockgpu.so
  for args in "${ARGS:---query 5}" ${ARGS2:+"$ARGS2"}; do
  env $envs python bench.py $args --no-also --no-cpu --steps ${STEPS:-20} --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$name', '[$args]', 'ms/step', d['ms_per_step'], 'kernel', r.get('kernel'), r.get('avg_launch_ms'), 'frac', r.get('frac'), 'rows', d['config'].get('result_rows'), r.get('kernels_ms'))"
  done
done; done
cp /tmp/shipped.so flock_amd/libflockgpu.so
