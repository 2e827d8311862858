#!/bin/bash
# launches / copies / host waits per execute of the whole-query plans on the generic operators (and on the fused pipelines beside them)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/api_counts; rm -rf "$OUT"; mkdir -p "$OUT"
for q in ${QS:-5 3 8}; do for mode in generic fused; do
  python tools/gpu_generic_api_counts.py $q 100 $mode 2>/dev/null | tail -1
  ( cd /tmp && rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/api_${q}_$mode -- python "$OLDPWD/tools/gpu_generic_api_counts.py" $q 100 $mode > /tmp/api_${q}_$mode.log 2>&1 )
  f=$(find /tmp/api_${q}_$mode -name "*hip_api_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/q${q}_${mode}_hip_api_stats.csv"
  f=$(find /tmp/api_${q}_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/q${q}_${mode}_kernel_stats.csv"
  python - "$OUT/q${q}_${mode}_hip_api_stats.csv" "$OUT/q${q}_${mode}_kernel_stats.csv" <<'PY'
import csv, sys
for f in sys.argv[1:]:
    try:
        rows = list(csv.DictReader(open(f)))
    except OSError:
        print("missing", f); continue
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    print(" ", f.split("/")[-1])
    for r in rows[:14]:
        print("   %-46s calls %7s  avg %9.1f us" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done; done
