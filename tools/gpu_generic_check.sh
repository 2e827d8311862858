#!/bin/bash
# gpurun helper: the plan tests that run whole queries on the generic operators, then bench.py --only-side plan_generic (fused vs generic per query)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${TAG:-generic}; mkdir -p $OUT
timeout 1500 python -m pytest ${TESTS:-tests/test_stage_plans.py tests/test_plan_boundary.py tests/test_plan_round4.py tests/test_plan_round5.py tests/test_plan_q4_q9_ysb.py tests/test_plan_fuzz.py} -q -m gpu -x -p no:cacheprovider > $OUT/tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $OUT/tests.log | tail -6
timeout 600 python bench.py --only-side plan_generic 2>$OUT/bench.err | tail -1 > $OUT/bench.json
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read())
pg = d.get("plan_generic", d)
for q in ("q5", "q3", "q8"):
    e = pg.get(q, {})
    print(q, {k: v for k, v in e.items() if "ms" in k or "over" in k})
PY
