#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_plan_round5.py -q -m gpu --maxfail=12 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r5b_tests_new.log
timeout 600 python bench.py --only-side arch > gpurun_out/r5b_arch.json 2> gpurun_out/r5b_arch.err
timeout 600 python bench.py --only-side plan_generic > gpurun_out/r5b_plan_generic.json 2> gpurun_out/r5b_plan_generic.err
timeout 600 python bench.py --only-side plan_stages > gpurun_out/r5b_plan_stages.json 2> gpurun_out/r5b_plan_stages.err
tail -n 5 gpurun_out/r5b_tests_new.log
tail -c 800 gpurun_out/r5b_arch.err
python - <<'PY'
import json
a=json.load(open('gpurun_out/r5b_arch.json'))
for k in ("filter","groupby","join","sort"):
    e=a[k]
    for m in ("fused","generic"):
        x=e.get(m,{})
        print(k,m,x.get("ms_per_execute"),x.get("first_execute_ms"),x.get("result_rows"),(x.get("roofline") or {}).get("frac"),x.get("error"),x.get("kernels_ms_per_execute"))
print(a.get("cpu_baseline"))
print(open('gpurun_out/r5b_plan_generic.json').read()[:3000])
print(open('gpurun_out/r5b_plan_stages.json').read()[:3000])
PY
