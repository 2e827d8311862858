#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/api_counts; mkdir -p "$OUT"
for q in ${QS:-3}; do for mode in on_device whole; do
  python tools/gpu_staged_api_counts.py $q 100 $mode 2>/dev/null | tail -1
  ( cd /tmp && rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/sapi_${q}_$mode -- python "$OLDPWD/tools/gpu_staged_api_counts.py" $q 100 $mode > /tmp/sapi_${q}_$mode.log 2>&1 )
  for kind in hip_api kernel; do
    f=$(find /tmp/sapi_${q}_$mode -name "*${kind}_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/staged_q${q}_${mode}_${kind}_stats.csv"
  done
  python - "$OUT/staged_q${q}_${mode}_hip_api_stats.csv" "$OUT/staged_q${q}_${mode}_kernel_stats.csv" <<'PY'
import csv, sys
for f in sys.argv[1:]:
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 90]
    rows.sort(key=lambda r: -int(r["Calls"]))
    print(" ", f.split("/")[-1])
    for r in rows[:30]:
        if r["Name"].startswith("__hip"): continue
        print("   %-60s per-run %6.1f avg %8.1f us" % (r["Name"][:60], int(r["Calls"]) / 102, float(r["AverageNs"]) / 1e3))
PY
done; done
