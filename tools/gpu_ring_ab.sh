#!/bin/bash
# gpurun helper: the plan_collect_pcie row (whole-window feed vs pane ring) with the experimental library and its knobs:
#   FLOCKGPU_RING_ROWS=1 (q5 keeps the panes' rows, not their groups), FLOCKGPU_PLAN_TIMES=1 (wall time per entry point on stderr)
cd "$GRAFT_REPO_ROOT"
cp flock_amd/libflockgpu.so /tmp/shipped.so
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
for e in "FLOCKGPU_PLAN_TIMES=1" "FLOCKGPU_PLAN_TIMES=1 FLOCKGPU_RING_ROWS=1"; do
env $e python bench.py --only-side plan_collect --steps 20 2>/tmp/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('also',d).get('plan_collect_pcie', d)
print('[$e]', 'whole', e.get('ms_per_step'), 'ring', (e.get('ring_one_instance_pageable') or {}).get('ms_per_window'), 'whole_reg', (e.get('one_instance_registered') or {}).get('ms_per_window'), 'ring_reg', (e.get('ring_one_instance_registered') or {}).get('ms_per_window'), e.get('ring_error'))"
grep 'plan times' /tmp/err.txt
done
cp /tmp/shipped.so flock_amd/libflockgpu.so
