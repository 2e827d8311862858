#!/bin/bash
# rocprofv3 kernel stats + FETCH / WRITE PMC passes of q3 at its BASELINE size (1e8 events: the five-launch sequence)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof; mkdir -p "$OUT"
cmd="python bench.py --query 3 --steps 10 --warmup 3 --no-also --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q3s -- $cmd > "$OUT/q3_1e8_stats_run.log" 2>&1
f=$(find /tmp/prof_q3s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/q3_1e8_kernel_stats.csv"
grep '^{' "$OUT/q3_1e8_stats_run.log" | tail -1 > "$OUT/q3_1e8_bench_under_rocprof.json"
rm -f "$OUT/q3_1e8_stats_run.log"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${c}_q3s -- $cmd > "$OUT/q3_1e8_${c}_run.log" 2>&1
  python - /tmp/pmc_${c}_q3s $c "$OUT/q3_1e8_pmc_${c}.csv" <<'PY'
import csv, glob, sys, collections
d, counter, out = sys.argv[1:4]
acc = collections.OrderedDict()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter: continue
        a = acc.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
with open(out, "w") as o:
    o.write("kernel,launches,avg_%s_KB\n" % counter)
    for k, (n, v) in acc.items():
        o.write('"%s",%d,%.3f\n' % (k, n, v / n))
PY
  rm -f "$OUT/q3_1e8_${c}_run.log"
done
head -12 "$OUT/q3_1e8_kernel_stats.csv" | cut -c1-120
