#!/bin/bash
# gpurun helper: the plan_collect_pcie row -- whole-window feed vs pane ring vs pane ring with the next pane prefetched, pageable and registered
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
python bench.py --only-side plan_collect --steps 20 2>/tmp/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('also',d).get('plan_collect_pcie', d)
g=lambda k: (e.get(k) or {}).get('ms_per_window')
print('pageable: whole', e.get('ms_per_step'), 'ring', g('ring_one_instance_pageable'), 'ring+prefetch', g('ring_prefetch_pageable'), '| registered: whole', g('one_instance_registered'), 'ring', g('ring_one_instance_registered'), 'ring+prefetch', g('ring_prefetch_registered'), e.get('ring_error'), e.get('variants_error'))"
done
