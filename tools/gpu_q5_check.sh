#!/bin/bash
# gpurun helper: everything that touches q5 (parity, goldens, exchange, plans), then the q5 step with its kernels
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x -k "q5 or golden or frozen or unfrozen or exchange or comm or stage or plan or baseline or full_size" 2>&1 | grep -E "passed|failed" | tail -3)
for i in 1 2; do
FLOCK_BENCH_STEP_TIMES=1 timeout 300 python bench.py --query 5 --no-also --no-cpu --steps 10 --warmup 3 2> gpurun_out/q5.err | tail -1 > gpurun_out/q5.json
grep 'step wall' gpurun_out/q5.err
python -c "
import json; d=json.loads(open('gpurun_out/q5.json').read()); r=d['roofline']; print('q5', d['ms_per_step'], 'frac', r['frac'], r['kernels_ms'])"
done
timeout 300 python bench.py --mode exchange --query 5 --no-also --no-cpu --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('exchange q5', d['ms_per_step'], d['value'], d.get('kernels_ms_rank0'))"
