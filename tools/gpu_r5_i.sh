#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_plan_round4.py tests/test_plan_round5.py tests/test_plan_boundary.py tests/test_stage_plans.py tests/test_gpu_q11.py tests/test_session_windows.py -q -m gpu --maxfail=20 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r5i_tests.log
tail -n 6 gpurun_out/r5i_tests.log
timeout 600 python bench.py --only-side arch --no-cpu > gpurun_out/r5i_arch.json 2> gpurun_out/r5i_arch.err
python - <<'PY'
import json
a=json.load(open('gpurun_out/r5i_arch.json'))
for k in ("sort","join"):
    e=a[k]
    for m in ("fused","generic"):
        x=e.get(m,{})
        print(k,m,x.get("ms_per_execute"),x.get("first_execute_ms"),x.get("result_rows"),(x.get("roofline") or {}).get("frac"),x.get("error"),x.get("kernels_ms_per_execute"))
PY
# re-profile the arch row on this build
OUT=$PWD/gpurun_out/prof; mkdir -p "$OUT"
summarise() {
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
side=arch
cmd="python bench.py --only-side $side --steps 3 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$side -- $cmd > "$OUT/${side}_stats_run.log" 2>&1
f=$(find /tmp/prof_$side -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${side}_kernel_stats.csv"
grep '^{' "$OUT/${side}_stats_run.log" | tail -1 > "$OUT/${side}_bench_under_rocprof.json"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${c}_$side -- $cmd > "$OUT/${side}_${c}_run.log" 2>&1
  summarise /tmp/pmc_${c}_$side $c "$OUT/${side}_pmc_${c}.csv"
done
rm -f "$OUT"/${side}_*_run.log
ls -la $OUT | tail -5
