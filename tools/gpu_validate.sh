#!/bin/bash
# gpurun helper: the round-end checks in one call -- every GPU parity test, smoke(), the default bench line.
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/validate_tests.log
cat gpurun_out/validate_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: v for k, v in d.items() if k not in ("also", "roofline", "cpu_baseline")})
print("roofline", d["roofline"]); print("cpu", d["cpu_baseline"])
for k, v in d["also"].items():
    if isinstance(v, dict):
        print(k, v.get("ms_per_step"), v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"))
PY
