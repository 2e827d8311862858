#!/bin/bash
# gpurun helper: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE PMC passes (each in its own run, as
# MI355X_MICROARCH.md prescribes) for every query's bench line.  Summaries land in gpurun_out/prof/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
summarise() {  # $1 = rocprof output dir, $2 = counter, $3 = out csv
python - "$1" "$2" "$3" <<'PY'
import csv, glob, sys, collections
d, counter, out = sys.argv[1:4]
acc = collections.OrderedDict()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter: continue
        k = r["Kernel_Name"]
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
with open(out, "w") as o:
    o.write("kernel,launches,avg_%s_KB\n" % counter)
    for k, (n, v) in acc.items():
        o.write('"%s",%d,%.3f\n' % (k, n, v / n))
PY
}
for q in ${QUERIES:-5 2 8 3 3_1e8 7 9 13}; do
  extra=""; [ "$q" = "3" ] && extra="--seconds 1000"
  qa=$q; [ "$q" = "3_1e8" ] && qa=3
  cmd="python bench.py --query $qa $extra --steps 6 --warmup 3 --no-also --no-cpu"
  # the kernel-stats run with bench.py's own default step counts (50 warm-up + 100 timed: the chip's clocks settle over the first dozen calls,
  # DESIGN section 4 "The clock ramp"); the counter passes stay short (counter collection serialises the launches, the clocks do not matter to bytes)
  scmd="python bench.py --query $qa $extra --steps ${STATS_STEPS:-100} --warmup ${STATS_WARMUP:-50} --no-also --no-cpu"
  rm -rf /tmp/prof_q$q /tmp/pmc_FETCH_SIZE_q$q /tmp/pmc_WRITE_SIZE_q$q   # (a box can serve several calls: an earlier call's files must not be picked up)
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q$q -- $scmd > "$OUT/q${q}_stats_run.log" 2>&1
  f=$(find /tmp/prof_q$q -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/q${q}_kernel_stats.csv"
  grep '^{' "$OUT/q${q}_stats_run.log" | tail -1 > "$OUT/q${q}_bench_under_rocprof.json"
  for c in ${PMC_COUNTERS-FETCH_SIZE WRITE_SIZE}; do   # (PMC_COUNTERS="" : kernel stats only)
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${c}_q$q -- $cmd > "$OUT/q${q}_${c}_run.log" 2>&1
    summarise /tmp/pmc_${c}_q$q $c "$OUT/q${q}_pmc_${c}.csv"
  done
  rm -f "$OUT"/q${q}_*_run.log
done
# the side entries bench.py reports under `also` (q11, YSB, JSON ingest) and the general-path rows: kernel stats + PMC passes each
for side in ${SIDES:-q11 ysb json}; do
  cmd="python bench.py --only-side $side --steps 3 --no-cpu"
  rm -rf /tmp/prof_$side /tmp/pmc_FETCH_SIZE_$side /tmp/pmc_WRITE_SIZE_$side
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$side -- $cmd > "$OUT/${side}_stats_run.log" 2>&1
  f=$(find /tmp/prof_$side -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${side}_kernel_stats.csv"
  grep '^{' "$OUT/${side}_stats_run.log" | tail -1 > "$OUT/${side}_bench_under_rocprof.json"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${c}_$side -- $cmd > "$OUT/${side}_${c}_run.log" 2>&1
    summarise /tmp/pmc_${c}_$side $c "$OUT/${side}_pmc_${c}.csv"
  done
  rm -f "$OUT"/${side}_*_run.log
done
for gen in ${GENERALS:-q3_general q8_general q5_uniform q3_hash q8_hash}; do
  cmd="python bench.py --only-general $gen --steps 3"
  rm -rf /tmp/prof_$gen /tmp/pmc_FETCH_SIZE_$gen /tmp/pmc_WRITE_SIZE_$gen
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$gen -- $cmd > "$OUT/${gen}_stats_run.log" 2>&1
  f=$(find /tmp/prof_$gen -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${gen}_kernel_stats.csv"
  grep '^{' "$OUT/${gen}_stats_run.log" | tail -1 > "$OUT/${gen}_bench_under_rocprof.json"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${c}_$gen -- $cmd > "$OUT/${gen}_${c}_run.log" 2>&1
    summarise /tmp/pmc_${c}_$gen $c "$OUT/${gen}_pmc_${c}.csv"
  done
  rm -f "$OUT"/${gen}_*_run.log
done
ls -la "$OUT"
# the "next" rows that bench.py only reports under `also` (q11, YSB, JSON ingest, q4, the 1e9 variants): one kernel-stats run
# of the whole default bench
if [ "${FULL:-1}" = "1" ]; then
  rm -rf /tmp/prof_all
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_all -- python bench.py --steps 3 --warmup 1 --no-cpu > "$OUT/all_stats_run.log" 2>&1
  f=$(find /tmp/prof_all -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/all_kernel_stats.csv"
  grep '^{' "$OUT/all_stats_run.log" | tail -1 > "$OUT/all_bench_under_rocprof.json"
  rm -f "$OUT/all_stats_run.log"
fi
ls -la "$OUT" | tail -8
