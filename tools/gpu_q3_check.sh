#!/bin/bash
# gpurun helper: q3 parity + timings at 1e8 and 1e9 events, and the exchange variants
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x -k "q3 or q8 or exchange or comm or golden or frozen or stage or plan" 2>&1 | tail -5)
for secs in 100 1000; do
  timeout 300 python bench.py --query 3 --seconds $secs --no-also --no-cpu --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/q3_$secs.json
  python -c "
import json; d=json.loads(open('gpurun_out/q3_$secs.json').read()); r=d['roofline']; print('q3 $secs s', d['ms_per_step'], 'probe frac', r['frac'], r['kernels_ms'])"
done
for q in 8 3; do
  extra=""; [ "$q" = "3" ] && extra="--seconds 1000"
  timeout 300 python bench.py --mode exchange --query $q $extra --no-also --no-cpu --steps 5 2>/dev/null | tail -1 > gpurun_out/bench_exchange_q$q.json
  python -c "
import json; d=json.loads(open('gpurun_out/bench_exchange_q$q.json').read()); print('exchange q$q', d['ms_per_step'], d['value'], d.get('kernels_ms_rank0'))"
done
