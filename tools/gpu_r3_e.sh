#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_nexmark_goldens.py tests/test_gpu_comm.py tests/test_gpu_exchange.py -m gpu -x -q -k "q5 or properties or exchange" > $O/q5_tests.log 2>&1; echo "q5 tests rc=$?"; tail -3 $O/q5_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-also --no-cpu > $O/q5.out 2> $O/q5.err; echo "q5 rc=$?"; cp gpurun_out/bench_also.json $O/q5_full.json
FLOCKGPU_Q5_WINDOW_SCAN=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-also --no-cpu > $O/q5_winscan.out 2> $O/q5_winscan.err; cp gpurun_out/bench_also.json $O/q5_winscan_full.json
timeout 400 python bench.py --only-general q5_uniform --steps 5 > $O/q5_uniform.out 2> $O/q5_uniform.err; echo "q5_uniform rc=$?"; tail -3 $O/q5_uniform.err
python - <<'P'
import json
for s in ("q5", "q5_winscan"):
    d = json.load(open(f"gpurun_out/r3e/{s}_full.json"))
    print(s, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernels_ms"])
try:
    d = json.loads(open("gpurun_out/r3e/q5_uniform.out").read().strip().splitlines()[-1])
    print("q5_uniform", d["value"], d["ms_per_step"], d["roofline"] and d["roofline"]["kernels_ms"])
except Exception as e:
    print("q5_uniform ERR", e)
P
