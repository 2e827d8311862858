#!/bin/bash
# gpurun helper: why does q5_count_kernel take 0.845 ms on one box / process and 0.915 ms on another?  Per DISPATCH of the kernel:
# wall duration (kernel trace) next to GRBM_GUI_ACTIVE (effective clock = GUI_ACTIVE / duration, MI355X_MICROARCH.md "DVFS
# give-back"), SQ wave cycles / busy cycles, and the L2's memory-side requests + hits per XCD instance.  Own rocprofv3 run per
# counter group; counters never combined with API traces.  Output: gpurun_out/q5_clock/summary.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/q5_clock
rm -rf "$OUT"; mkdir -p "$OUT"
cmd="python bench.py --query 5 --steps ${STEPS:-12} --warmup 2 --no-also --no-cpu"
rocm-smi --showclocks --showpower > "$OUT/smi_before.txt" 2>&1
# un-profiled reference line (per-step times through FLOCK_BENCH_STEP_TIMES)
FLOCK_BENCH_STEP_TIMES=1 $cmd > "$OUT/plain.json" 2> "$OUT/plain.err"
for grp in "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ"; do
  tag=$(echo $grp | tr ' ' '+')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/q5clk_$tag -- $cmd > "$OUT/run_$tag.log" 2>&1
  mkdir -p "$OUT/$tag"
  for f in $(find /tmp/q5clk_$tag -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do cp "$f" "$OUT/$tag/"; done
  grep '^{' "$OUT/run_$tag.log" | tail -1 > "$OUT/$tag/bench.json"
done
rocm-smi --showclocks --showpower > "$OUT/smi_after.txt" 2>&1
python - "$OUT" <<'PY' > "$OUT/summary.txt" 2>&1
import csv, glob, sys, os, json, collections
out = sys.argv[1]
try:
    j = json.loads(open(out + "/plain.json").read().strip().splitlines()[-1])
    print("un-profiled: ms/step", j["ms_per_step"], "count kernel avg ms", j["roofline"]["avg_launch_ms"]); print(open(out + "/plain.err").read()[-600:])
except Exception as e:
    print("plain run unreadable", e)
for d in sorted(glob.glob(out + "/*/")):
    cc = glob.glob(d + "*counter_collection.csv"); kt = glob.glob(d + "*kernel_trace.csv")
    if not cc or not kt: continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    per = collections.OrderedDict()
    for r in csv.DictReader(open(cc[0])):
        if "q5_count_kernel<false" not in r["Kernel_Name"] and "q5_count_kernel<(bool)0" not in r["Kernel_Name"] and "q5_count_kernel" not in r["Kernel_Name"]: continue
        if "slow" in r["Kernel_Name"]: continue
        per.setdefault(r["Dispatch_Id"], collections.OrderedDict())
        k = r["Counter_Name"]
        per[r["Dispatch_Id"]][k] = per[r["Dispatch_Id"]].get(k, 0.0) + float(r["Counter_Value"])
    print("==", os.path.basename(d.rstrip("/")))
    for did, c in per.items():
        ns = dur.get(did, (0, ""))[0]
        line = "  dispatch %6s  %8.1f us " % (did, ns / 1e3)
        for k, v in c.items():
            line += " %s=%.4g" % (k, v)
        if "GRBM_GUI_ACTIVE" in c and ns:
            line += "  -> effective clock %.3f GHz" % (c["GRBM_GUI_ACTIVE"] / ns)
        if "SQ_BUSY_CYCLES" in c and ns:
            line += "  busy/ns %.3f" % (c["SQ_BUSY_CYCLES"] / ns)
        print(line)
PY
cat "$OUT/summary.txt" | head -120
rm -f "$OUT"/run_*.log
# q3 at the BASELINE size (1e8 events): kernel stats of the whole call
mkdir -p $PWD/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q3_100 -- python bench.py --query 3 --seconds 100 --steps 5 --warmup 2 --no-also --no-cpu > /tmp/q3_100.log 2>&1
f=$(find /tmp/prof_q3_100 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/prof/q3_1e8_kernel_stats.csv
grep '^{' /tmp/q3_100.log | tail -1 > gpurun_out/prof/q3_1e8_bench_under_rocprof.json
head -25 gpurun_out/prof/q3_1e8_kernel_stats.csv
