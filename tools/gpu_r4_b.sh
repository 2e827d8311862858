#!/bin/bash
# gpurun helper (round 4): GPU tests, then the q8 / q3 rows and the general-path rows on their own
OUT=gpurun_out/${TAG:-r4b}; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -25 $OUT/tests.log
for q in 8 3; do
  timeout 300 python bench.py --query $q --no-also --no-cpu --steps 20 --warmup 3 2>$OUT/q$q.err | tail -1 > $OUT/q$q.json
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/q$q.json").read()); r=d["roofline"]
    print("q$q", "ms/step", d["ms_per_step"], "frac", r["frac"], r.get("avg_launch_ms"))
except Exception as e:
    print("q$q failed", e); print(open("$OUT/q$q.err").read()[-1500:])
PY
done
timeout 300 python bench.py --query 3 --seconds 1000 --no-also --no-cpu --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('q3@1e9', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_ms'])"
for g in ${GENERALS:-q8_general q3_general q8_hash q3_hash}; do
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
