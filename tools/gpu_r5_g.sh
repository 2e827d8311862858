#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_nexmark_goldens.py -q -m gpu --maxfail=20 -p no:cacheprovider -k "q3 or hash" 2>&1 | tail -12 > gpurun_out/r5g_tests.log
tail -n 5 gpurun_out/r5g_tests.log
timeout 300 python bench.py --only-general q3_hash > gpurun_out/r5g_q3_hash.json 2> gpurun_out/r5g_q3_hash.err
python - <<'PY'
import json
a=json.load(open('gpurun_out/r5g_q3_hash.json'))
print({k:a.get(k) for k in ("value","ms_per_step")}, (a.get("roofline") or {}).get("frac"), (a.get("roofline") or {}).get("kernels_ms"))
PY
