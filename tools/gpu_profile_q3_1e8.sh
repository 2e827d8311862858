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
head -12 "$OUT/q3_1e8_kernel_stats.csv" | cut -c1-120
