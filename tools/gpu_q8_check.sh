#!/bin/bash
# gpurun helper: q8 parity (batched ABI, goldens, plan path, exchange) + the q8 bench line
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_nexmark_goldens.py tests/test_gpu_comm.py tests/test_stage_plans.py tests/test_plan_boundary.py -m gpu -q -x -k "q8 or Q8 or frozen or unfrozen or exchange or staged or generic" 2>&1 | tail -12 | tee gpurun_out/q8_tests.log)
timeout 300 python bench.py --query 8 --no-also --no-cpu --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/q8.json
python -c "
import json; d=json.loads(open('gpurun_out/q8.json').read()); print('q8', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernels_ms'])"
