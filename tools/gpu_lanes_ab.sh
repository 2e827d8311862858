#!/bin/bash
# gpurun helper: the plan path's staging lanes on the plan_collect_pcie row (FLOCKGPU_STAGE_LANES is read by the EXPERIMENTAL library only: build it and copy it over libflockgpu.so first)
cd "$GRAFT_REPO_ROOT"
for lanes in 2 4 6 8 4; do
FLOCKGPU_STAGE_LANES=$lanes python bench.py --only-side plan_collect --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('also',d).get('plan_collect_pcie', d)
print('lanes $lanes', 'whole', e.get('ms_per_step'), 'ring', (e.get('ring_one_instance_pageable') or {}).get('ms_per_window'), 'two', (e.get('two_instances_pageable') or {}).get('ms_per_window'), 'whole_reg', (e.get('one_instance_registered') or {}).get('ms_per_window'), 'ring_reg', (e.get('ring_one_instance_registered') or {}).get('ms_per_window'))"
done
