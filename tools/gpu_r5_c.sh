#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --maxfail=20 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r5c_tests_all.log
timeout 600 python bench.py --only-side arch --no-cpu > gpurun_out/r5c_arch.json 2> gpurun_out/r5c_arch.err
tail -n 8 gpurun_out/r5c_tests_all.log
python - <<'PY'
import json
a=json.load(open('gpurun_out/r5c_arch.json'))
for k in ("filter","groupby","join","sort"):
    e=a[k]
    for m in ("fused","generic"):
        x=e.get(m,{})
        print(k,m,x.get("ms_per_execute"),x.get("first_execute_ms"),x.get("result_rows"),(x.get("roofline") or {}).get("frac"),x.get("error"),x.get("kernels_ms_per_execute"))
PY
