#!/bin/bash
# gpurun helper: rocprofv3 kernel stats of one bench invocation: ARGS="--query 8" TAG=q8 bash tools/gpu_rocprof_one.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof1; mkdir -p "$OUT"
cmd="python bench.py ${ARGS:---query 8} --steps ${STEPS:-10} --warmup 2 --no-also --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- $cmd > "$OUT/${TAG}_run.log" 2>&1
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_kernel_stats.csv"
grep '^{' "$OUT/${TAG}_run.log" | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
python - "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0][:64]
    if n.startswith("gen_") or "at::" in n or "rocprim" in n: continue
    print(f"{n:64s} {r['Calls']:>4s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
