#!/bin/bash
# gpurun helper: the arch/ops harness rows only (bench.py --only-side arch), per-kernel times of every operator
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${TAG:-arch}; mkdir -p $OUT
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -q -m gpu -x -p no:cacheprovider > $OUT/tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $OUT/tests.log | tail -8; fi
timeout 900 python bench.py --only-side arch ${BENCH_ARGS} 2>$OUT/bench.err | tail -1 > $OUT/bench.json
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read())
a = d.get("arch_ops", d)
for name in ${ROWS:-("filter", "groupby", "join", "join_sparse", "sort")}:
    e = a.get(name, {})
    for mode in ("fused", "generic"):
        m = e.get(mode, {})
        print(name, mode, m.get("ms_per_execute"), "rows", m.get("result_rows"), (m.get("roofline") or {}).get("frac"), m.get("kernels_ms_per_execute"), m.get("error"))
PY
tail -3 $OUT/bench.err
