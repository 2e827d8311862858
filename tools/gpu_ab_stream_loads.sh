#!/bin/bash
# gpurun helper: A/B of non-temporal vs plain 16-byte loads in the shared flag-tile loader (scan.hpp: load_flag_tile), same box, alternating.
# B = a second library built with -DFLOCKGPU_AB_PLAIN_TILE_LOADS (flock_amd/libflockgpu_plain_tile_loads.so), swapped in for its runs.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
# build B first: FLOCKGPU_BUILD_EXPERIMENTAL=1 FLOCKGPU_BUILD_DEFINES=-DFLOCKGPU_AB_PLAIN_TILE_LOADS python -m flock_amd.build
cp flock_amd/libflockgpu.so /tmp/A.so; cp flock_amd/libflockgpu_experimental.so /tmp/B.so
for round in 1 2; do for v in A B; do
  cp /tmp/$v.so flock_amd/libflockgpu.so
  python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=['$v']
def f(e): r=e.get('roofline') or {}; return '%s/%s' % (e.get('ms_per_step'), r.get('avg_launch_ms'))
o.append('q5 '+f(d)); o.append('q3 '+f(d['q3']))
for k in ('q2','q8','q3_1e9_events','q2_1e9_bids','q8_4e9_events','q7_next','q9_next','q4_next','q13_next','ysb_next','json_ingest_next','q11_next'): o.append(k.replace('_next','')+' '+f(d['also'][k]))
for k,v in d['also']['exchange_1rank'].items(): o.append('x'+k+' '+str(v.get('ms_per_step')))
print(' | '.join(o))"
done; done
cp /tmp/A.so flock_amd/libflockgpu.so
