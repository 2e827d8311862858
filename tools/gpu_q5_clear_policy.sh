#!/bin/bash
# gpurun helper: the mechanism behind "cache policy" (DESIGN.md section 4).  q5_count_kernel under rocprofv3 PMC passes with the counters'
# clear done (a) with cached stores right before the count pass -- round 1's behaviour, FLOCKGPU_Q5_PLAIN_CLEAR=1 -- and (b) with non-temporal
# stores after the previous call: duration, the L2's memory-side WRITE requests during the count pass (write-back of the clear's dirty lines
# shows up there), read requests, hits / misses.  Own run per counter group; no API traces with counters.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/q5_clear_policy; rm -rf "$OUT"; mkdir -p "$OUT"
# needs the experimental library (FLOCKGPU_BUILD_EXPERIMENTAL=1 python -m flock_amd.build): the shipped one ignores FLOCKGPU_Q5_PLAIN_CLEAR
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
cmd="python bench.py --query 5 --steps 8 --warmup 2 --no-also --no-cpu"
for mode in plain nontemporal; do
  if [ "$mode" = "plain" ]; then export FLOCKGPU_Q5_PLAIN_CLEAR=1; else unset FLOCKGPU_Q5_PLAIN_CLEAR; fi
  $cmd 2>/dev/null | tail -1 > "$OUT/${mode}_plain_run.json"
  for grp in "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
    tag=$(echo $grp | tr ' ' '+')
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/cp_${mode}_$tag -- $cmd > /dev/null 2>&1
    mkdir -p "$OUT/$mode/$tag"
    for f in $(find /tmp/cp_${mode}_$tag -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do cp "$f" "$OUT/$mode/$tag/"; done
  done
done
python - "$OUT" <<'PY' > "$OUT/summary.txt" 2>&1
import csv, glob, sys, os, json, collections
out = sys.argv[1]
for mode in ("plain", "nontemporal"):
    try:
        j = json.loads(open(f"{out}/{mode}_plain_run.json").read())
        print(f"== clear with {mode} stores: un-profiled ms/step {j['ms_per_step']}, count kernel {j['roofline']['avg_launch_ms']} ms, clear kernel counted separately below")
    except Exception as e:
        print(mode, "plain run unreadable", e)
    for d in sorted(glob.glob(f"{out}/{mode}/*/")):
        cc = glob.glob(d + "*counter_collection.csv"); kt = glob.glob(d + "*kernel_trace.csv")
        if not cc or not kt: continue
        dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
        for kern in ("q5_count_kernel<false>", "q5_count_kernel<(bool)0>", "q5_clear_kernel", "q5_scan_kernel<false>", "q5_range_kernel"):
            acc = collections.OrderedDict(); n = 0; t = 0
            seen = set()
            for r in csv.DictReader(open(cc[0])):
                if kern not in r["Kernel_Name"]: continue
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"]); n += 1; t += dur.get(r["Dispatch_Id"], 0)
            if n:
                print("   %-26s %-44s launches %3d avg %8.1f us  " % (kern, os.path.basename(d.rstrip('/')), n, t / n / 1e3) + "  ".join("%s=%.4g" % (k, v / n) for k, v in acc.items()))
PY
cat "$OUT/summary.txt"
