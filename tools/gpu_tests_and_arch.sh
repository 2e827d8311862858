cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/round3
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/round3/tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/round3/tests.log | tail -15
timeout 600 python bench.py --only-side arch --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('filter','groupby','join','sort'):
    e=d.get(k,{})
    for m in ('fused','generic'):
        x=e.get(m,{}); print(k, m, x.get('ms_per_execute'), x.get('result_rows'), (x.get('roofline') or {}).get('frac'), x.get('kernels_ms_per_execute'), x.get('error'))"
