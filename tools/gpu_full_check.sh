#!/bin/bash
mkdir -p gpurun_out/r3x
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3x/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3x/tests.log
( time timeout 900 python bench.py > gpurun_out/r3x/bench.out 2> gpurun_out/r3x/bench.err ) 2> gpurun_out/r3x/bench.time; echo "bench rc=$?"; tail -3 gpurun_out/r3x/bench.time
tail -1 gpurun_out/r3x/bench.out | wc -c; tail -1 gpurun_out/r3x/bench.out
cp gpurun_out/bench_also.json gpurun_out/r3x/bench_also_default.json
python __graft_entry__.py --smoke > gpurun_out/r3x/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3x/smoke.log
