#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_next_goldens.py -m gpu -x -q > $O/next_goldens.log 2>&1; echo "next goldens rc=$?"; tail -3 $O/next_goldens.log
for g in q3_general q8_general q5_uniform; do
  timeout 400 python bench.py --only-general $g --steps 5 > $O/$g.out 2> $O/$g.err; echo "$g rc=$?"
  tail -1 $O/$g.out | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline'] and {k:d['roofline'][k] for k in ('kernel','frac','kernels_ms')})"
done
