#!/bin/bash
# gpurun helper: the round-end sequence the driver runs -- the whole GPU suite, smoke, the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${TAG:-round}; mkdir -p $OUT
timeout ${TEST_TIMEOUT:-2400} python -m pytest tests/ -q -m gpu -p no:cacheprovider ${PYTEST_ARGS} > $OUT/tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" $OUT/tests.log | tail -15
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
if [ -z "$NO_BENCH" ]; then
  t0=$SECONDS; timeout 900 python bench.py ${BENCH_ARGS} 2>$OUT/bench.err | tail -1 > $OUT/bench.json; echo "bench wall $((SECONDS - t0)) s"
  python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read())
print({k: d[k] for k in ("value", "ms_per_step", "scaling", "n_gpus")}, d["roofline"], d["cpu_baseline"])
print("q3", d.get("q3"))
print("also", d.get("also"))
print(len(open("$OUT/bench.json").read()), "bytes")
PY
fi
