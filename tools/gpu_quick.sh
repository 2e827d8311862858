#!/bin/bash
# gpurun helper: a chosen slice of the GPU tests (TESTS="tests/test_gpu_parity.py -k q3"), then chosen bench rows (ROWS="3 3@1000 8")
OUT=gpurun_out/${TAG:-quick}; mkdir -p $OUT
if [ -n "$TESTS" ]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest $TESTS -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -${TEST_TAIL:-8} $OUT/tests.log
fi
for row in ${ROWS:-3 3@1000}; do
  q=${row%@*}; secs=""; [ "$row" != "$q" ] && secs="--seconds ${row#*@}"
  timeout 300 python bench.py --query $q $secs --no-also --no-cpu --steps ${STEPS:-20} --warmup 3 2>$OUT/q$row.err | tail -1 > $OUT/q$row.json
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/q$row.json").read()); r=d["roofline"]
    print("q$row", "ms/step", d["ms_per_step"], "frac", r["frac"], r.get("kernel"), r.get("avg_launch_ms"), r.get("kernels_ms"))
except Exception as e:
    print("q$row failed", e); print(open("$OUT/q$row.err").read()[-1500:])
PY
done
for g in $GENERALS; do
  timeout 300 python bench.py --only-general $g --steps 5 2>$OUT/$g.err | tail -1 > $OUT/$g.json
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/$g.json").read()); r=d["roofline"] or {}
    print("$g", "ms/step", d["ms_per_step"], "frac", r.get("frac"), r.get("kernels_ms"))
except Exception as e:
    print("$g failed", e); print(open("$OUT/$g.err").read()[-1500:])
PY
done
