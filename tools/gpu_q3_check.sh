#!/bin/bash
# gpurun helper: q3 parity (batched ABI, goldens, plan path, exchange) + the q3 bench lines at 1e8 and 1e9 events
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_nexmark_goldens.py tests/test_gpu_comm.py -m gpu -q -x -k "q3 or Q3 or frozen or unfrozen or exchange" 2>&1 | tail -6 | tee gpurun_out/q3_tests.log)
for secs in 100 1000; do
  timeout 300 python bench.py --query 3 --seconds $secs --no-also --no-cpu --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/q3_$secs.json
  python -c "
import json; d=json.loads(open('gpurun_out/q3_$secs.json').read()); print('q3 $secs s', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['kernels_ms'])"
done
