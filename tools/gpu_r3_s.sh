#!/bin/bash
mkdir -p gpurun_out/r3s
timeout 900 python -m pytest tests/test_stage_plans.py tests/test_plan_boundary.py tests/test_plan_q4_q9_ysb.py -x -q -m gpu > gpurun_out/r3s/tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r3s/tests.log
timeout 300 python bench.py --only-side plan_stages --steps 20 > gpurun_out/r3s/ps.out 2> gpurun_out/r3s/ps.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3s/ps.out').read().strip().splitlines()[-1])
for q in ('q3','q5','q8'): print(q, {k:d[q][k] for k in ('whole_plan_ms','staged_ms','one_instance_ms','one_instance_over_whole','on_device_ms','on_device_over_whole')})
PY
tail -3 gpurun_out/r3s/ps.err
