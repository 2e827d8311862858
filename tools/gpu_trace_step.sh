#!/bin/bash
# gpurun helper: HIP API timeline of the last step(s) of a bench.py invocation (no counters: --hip-trace only), to see where a call's
# host time goes -- synchronisations, small copies, gaps between launches.  usage: gpu_trace_step.sh <tail-calls> <bench args...>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
N=$1; shift
OUT=$PWD/gpurun_out/trace; mkdir -p "$OUT"
rm -rf /tmp/trace_step
rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/trace_step -- python bench.py "$@" > "$OUT/run.log" 2>&1
grep '^{' "$OUT/run.log" | tail -1 | cut -c1-400
python - "$N" <<'PY'
import csv, glob, sys
n = int(sys.argv[1])
f = glob.glob("/tmp/trace_step/**/*hip_api_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region ends at the last hipDeviceSynchronize before the first hipFree burst; walk back from there
last = max(i for i, r in enumerate(rows) if r["Function"] == "hipLaunchKernel")
end = min(len(rows), last + 30)
seg = rows[max(0, end - n):end]
prev = None
counts = {}
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    counts[r["Function"]] = counts.get(r["Function"], 0) + 1
    if r["Function"] not in ("hipEventSynchronize", "hipEventElapsedTime", "hipGetLastError", "__hipPushCallConfiguration", "__hipPopCallConfiguration", "hipEventCreate", "hipEventRecord"):
        print("%-28s dur %8.1f us  gap_before %8.1f us" % (r["Function"], (e - s) / 1e3, gap))
print("calls in the tail:", counts)
PY
