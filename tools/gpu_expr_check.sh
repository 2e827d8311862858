#!/bin/bash
# gpurun helper: the general expression evaluator -- its tests, then its two bench rows
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/${TAG:-expr}
timeout 1500 python -m pytest tests/test_plan_round5b.py tests/test_plan_round6.py tests/test_plan_round5.py tests/test_plan_fuzz.py -q -m gpu -x -p no:cacheprovider > gpurun_out/${TAG:-expr}/tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/${TAG:-expr}/tests.log | tail -8
timeout 600 python bench.py --only-side expr ${BENCH_ARGS} 2>gpurun_out/${TAG:-expr}/bench.err | tail -1 > gpurun_out/${TAG:-expr}/bench.json
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG:-expr}/bench.json").read())
    for k in ("expr_project", "expr_filter"):
        e = d.get(k, {})
        print(k, e.get("ms_per_step"), "rows", e.get("result_rows"), "roofline", (e.get("roofline") or {}).get("frac"), (e.get("roofline") or {}).get("avg_launch_ms"), e.get("kernels_ms_per_execute"), "cpu", (e.get("cpu_baseline") or {}).get("value"), e.get("error"))
except Exception as ex:
    print("bench failed", ex); print(open("gpurun_out/${TAG:-expr}/bench.err").read()[-1500:])
PY
