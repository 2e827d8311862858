#!/bin/bash
# gpurun helper: the whole GPU suite, then the plan_generic side entry (a window's plan on the generic operators against its fused pipeline)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/${TAG:-generic}
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/${TAG:-generic}/tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG:-generic}/tests.log | tail -15
timeout 600 python bench.py --only-side plan_generic 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k in ('q3','q5','q8'):
    print(k, d.get(k))"
