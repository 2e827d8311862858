#!/bin/bash
# gpurun helper: where does the one-rank exchange step spend its time?  HIP API + kernel + memcpy stats per query.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_exchange
rm -rf "$OUT"; mkdir -p "$OUT"
for q in ${QUERIES:-8 5 3}; do
  extra=""; [ "$q" = "3" ] && extra="--seconds 1000"
  rm -rf /tmp/px_q$q   # (a box can serve several calls: an earlier call's files must not be picked up)
  cmd="python bench.py --mode exchange --query $q $extra --steps 5 --warmup 2 --no-also --no-cpu"
  rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/px_q$q -- $cmd > "$OUT/q${q}_run.log" 2>&1
  for kind in hip_api_stats kernel_stats memory_copy_stats; do
    f=$(find /tmp/px_q$q -name "*${kind}.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/q${q}_${kind}.csv"
  done
  grep '^{' "$OUT/q${q}_run.log" | tail -1 > "$OUT/q${q}_bench.json"
  echo "== q$q"; python - "$OUT" $q <<'PY'
import csv, json, sys
out, q = sys.argv[1], sys.argv[2]
d = json.loads(open(f"{out}/q{q}_bench.json").read()); print("ms/step", d["ms_per_step"])
for kind in ("hip_api_stats", "kernel_stats", "memory_copy_stats"):
    try:
        rows = list(csv.DictReader(open(f"{out}/q{q}_{kind}.csv")))
    except Exception as e:
        print(kind, "missing", e); continue
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print(kind)
    for r in rows[:14]:
        print("   %-46s calls %6s total %9.3f ms avg %9.1f us" % (r["Name"][:46], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
done
