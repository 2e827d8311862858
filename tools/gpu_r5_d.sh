#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_plan_round4.py tests/test_plan_round5.py tests/test_stage_plans.py tests/test_plan_q4_q9_ysb.py tests/test_plan_boundary.py tests/test_abi_symbols.py tests/test_gpu_exchange.py -q -m gpu --maxfail=20 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r5d_tests.log
timeout 300 python bench.py --only-general q3_hash > gpurun_out/r5d_q3_hash.json 2> gpurun_out/r5d_q3_hash.err
timeout 300 python bench.py --only-side plan_stages > gpurun_out/r5d_plan_stages.json 2> gpurun_out/r5d_plan_stages.err
timeout 300 python bench.py --only-side plan_generic > gpurun_out/r5d_plan_generic.json 2> gpurun_out/r5d_plan_generic.err
tail -n 6 gpurun_out/r5d_tests.log
python - <<'PY'
import json
a=json.load(open('gpurun_out/r5d_q3_hash.json'))
print({k:a.get(k) for k in ("value","ms_per_step")}, (a.get("roofline") or {}).get("frac"), (a.get("roofline") or {}).get("kernels_ms"))
s=json.load(open('gpurun_out/r5d_plan_stages.json'))
for q in ("q3","q5","q8"):
    e=s[q]; print(q, e["whole_plan_ms"], e["staged_ms"], e["one_instance_ms"], e["on_device_ms"], e["on_device_over_whole"], e["top_kernels_ms_per_run"])
g=json.load(open('gpurun_out/r5d_plan_generic.json'))
for q in ("q3","q5","q8"):
    print(q, g[q])
PY
