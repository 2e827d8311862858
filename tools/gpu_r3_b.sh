#!/bin/bash
# round 3, second GPU call: comm failure semantics, q3 five-launch sequence (parity + timing)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_comm.py -x -q > $O/comm_tests.log 2>&1; echo "comm rc=$?"; tail -3 $O/comm_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "q3 or q8" > $O/q3_tests.log 2>&1; echo "q3 rc=$?"; tail -3 $O/q3_tests.log
timeout 900 python -m pytest tests/test_nexmark_goldens.py -x -q -k "q3 or q8" > $O/golden_tests.log 2>&1; echo "golden rc=$?"; tail -3 $O/golden_tests.log
for s in 100 1000; do
  timeout 300 python bench.py --query 3 --seconds $s --steps 30 --warmup 5 --no-also --no-cpu > $O/q3_$s.out 2> $O/q3_$s.err; echo "q3 $s rc=$?"
  cp gpurun_out/bench_also.json $O/q3_${s}_full.json
  FLOCKGPU_Q3_NO_FAST=1 timeout 300 python bench.py --query 3 --seconds $s --steps 30 --warmup 5 --no-also --no-cpu > $O/q3_${s}_nofast.out 2> $O/q3_${s}_nofast.err
  cp gpurun_out/bench_also.json $O/q3_${s}_nofast_full.json
done
python - <<'P'
import json
for s in (100, 1000):
    for v in ("", "_nofast"):
        try:
            d = json.load(open(f"gpurun_out/r3b/q3_{s}{v}_full.json"))
            print(s, v or "fast", d["ms_per_step"], d["roofline"]["frac"], {k: x for k, x in d["roofline"]["kernels_ms"].items()})
        except Exception as e:
            print(s, v, "ERR", e)
P
