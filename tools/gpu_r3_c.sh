#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/all_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/all_tests.log
for s in 100 1000; do
  timeout 300 python bench.py --query 3 --seconds $s --steps 30 --warmup 5 --no-also --no-cpu > $O/q3_$s.out 2> $O/q3_$s.err; echo "q3 $s rc=$?"
  cp gpurun_out/bench_also.json $O/q3_${s}_full.json
done
FLOCKGPU_Q3_NO_FAST=1 timeout 300 python bench.py --query 3 --seconds 100 --steps 30 --warmup 5 --no-also --no-cpu > $O/q3_100_nofast.out 2>&1
cp gpurun_out/bench_also.json $O/q3_100_nofast_full.json
python - <<'P'
import json
for s in ("100", "1000", "100_nofast"):
    try:
        d = json.load(open(f"gpurun_out/r3c/q3_{s}_full.json"))
        print(s, d["ms_per_step"], d["roofline"]["frac"], {k: x for k, x in d["roofline"]["kernels_ms"].items()})
    except Exception as e:
        print(s, "ERR", e)
P
