cd "$GRAFT_REPO_ROOT"
python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/bench_nocpu.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_nocpu.json").read())
print("q5", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
q3 = d["q3"]; print("q3", q3["ms_per_step"], q3["roofline"]["frac"])
for k, v in d["also"].items():
    if k == "exchange_1rank":
        for kk, vv in v.items(): print("x1", kk, vv.get("ms_per_step"), vv.get("over_window_sharded_step"))
    elif isinstance(v, dict): print(k, v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("avg_launch_ms"))
PY
