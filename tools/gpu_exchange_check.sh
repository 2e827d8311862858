#!/bin/bash
# gpurun helper: exchange tests, then the one-rank exchange step of q5 / q8 / q3 with its kernels and HIP API calls.
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_comm.py tests/test_gpu_exchange.py -m gpu -q -x 2>&1 | tail -5)
for q in 5 8 3; do
  extra=""; [ "$q" = "3" ] && extra="--seconds 1000"
  timeout 300 python bench.py --mode exchange --query $q $extra --no-also --no-cpu --steps 5 2>/dev/null | tail -1 > gpurun_out/bench_exchange_q$q.json
  python -c "
import json; d=json.loads(open('gpurun_out/bench_exchange_q$q.json').read()); print('exchange q$q', d['ms_per_step'], d['value'], d.get('kernels_ms_rank0'))"
done
