#!/bin/bash
# gpurun helper: q4 / q9 parity + their bench lines
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_plan_q4_q9_ysb.py -m gpu -q -x -k "q4 or q9" 2>&1 | grep -E "passed|failed" | tail -2)
for q in 9 4; do
  python bench.py --query $q --no-also --no-cpu --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('q$q', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_ms'])"
done
