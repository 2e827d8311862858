#!/bin/bash
# the count pass's stand-alone twin (tools/micro/stream_read: loads / + compute / + flush) and the library's own step, on ONE box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in 1 2; do
  tools/micro/stream_read | grep -E "count_like load|front \(|  \+ both" | head -4
  python bench.py --query 5 --no-also --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('library step', d['ms_per_step'], 'count', r['avg_launch_ms'], r.get('launch_ms_spread'))"
done
