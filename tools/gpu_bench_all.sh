#!/bin/bash
# gpurun helper: the default bench line + exchange-mode lines (one rank)
mkdir -p gpurun_out
(time timeout 600 python bench.py) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
    r=d["roofline"]; print("q5 value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], r["kernels_ms"])
    print("cpu", d["cpu_baseline"])
    for k,v in d.get("also",{}).items():
        print(k, {kk: vv for kk, vv in v.items() if kk not in ("roofline",)}, (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("kernels_ms"))
except Exception as e:
    print("parse failed", e); print(open("gpurun_out/bench_default.json").read()[-3000:])
PY
for q in 8 3 5; do
  timeout 300 python bench.py --query $q --mode exchange --no-also --no-cpu --steps 3 --warmup 1 2>&1 | grep "^{" | tail -1 > gpurun_out/bench_exchange_q$q.json
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_exchange_q$q.json").read())
    print("exchange q$q", d["value"], "ms", d["ms_per_step"], d["config"]["parallelism"], d["roofline"]["kernels_ms"] if d["roofline"] else None)
except Exception as e:
    print("exchange q$q failed", e, open("gpurun_out/bench_exchange_q$q.json").read()[-1500:])
PY
done
