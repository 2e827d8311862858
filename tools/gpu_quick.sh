#!/bin/bash
# gpurun helper: parity tests, then per-query kernel timings from bench.py (no CPU leg).
mkdir -p gpurun_out
(timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/quick_tests.log
cat gpurun_out/quick_tests.log
for q in ${QUERIES:-2 3 8 5 7 9 4 13}; do
  timeout 200 python bench.py --query $q --no-also --no-cpu --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/quick_q$q.json
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/quick_q$q.json").read())
    r=d["roofline"]
    print("q$q", "ms/step", d["ms_per_step"], "frac", r["frac"], r["kernels_ms"])
except Exception as e:
    print("q$q failed", e, open("gpurun_out/quick_q$q.json").read()[-2000:])
PY
done
