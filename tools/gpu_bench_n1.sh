#!/bin/bash
# gpurun helper: the default bench line (N = 1) + the exchange mode on one rank + a 2-rank torchrun on ONE GPU is not possible
# (RCCL refuses duplicate devices), so N > 1 is covered by tests/test_gpu_comm.py's local ranks.
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 800 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: v for k, v in d.items() if k not in ("also", "roofline", "cpu_baseline", "q3")})
print("roofline", d["roofline"]); print("cpu", d["cpu_baseline"])
q3 = d.get("q3", {}); print("q3", q3.get("ms_per_step"), q3.get("value"), (q3.get("roofline") or {}).get("frac"), (q3.get("roofline") or {}).get("kernels_ms"), q3.get("error"))
for k, v in d.get("also", {}).items():
    if k == "exchange_1rank":
        for kk, vv in v.items():
            print("exchange_1rank", kk, vv.get("ms_per_step") if isinstance(vv, dict) else vv, vv.get("over_window_sharded_step") if isinstance(vv, dict) else "", vv.get("kernels_ms_rank0") if isinstance(vv, dict) else "", vv.get("error") if isinstance(vv, dict) else "")
    elif isinstance(v, dict):
        print(k, v.get("ms_per_step"), v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"), ((v.get("cpu_baseline") or {}).get("acero") or {}).get("value"), v.get("error"))
PY
timeout 300 python bench.py --mode exchange --no-also --no-cpu --steps 5 2>/dev/null | tail -1 > gpurun_out/bench_exchange_q5.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_exchange_q5.json').read()); print('exchange q5 headline', d['ms_per_step'], d['value'], d['scaling'], d['config']['collective'], d.get('kernels_ms_rank0'))"
