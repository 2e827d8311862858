#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -q -m gpu --maxfail=20 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/full_tests_all.log
tail -n 6 gpurun_out/full_tests_all.log
timeout 900 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err
tail -c 3800 gpurun_out/full_bench.json
cp gpurun_out/bench_also.json gpurun_out/full_bench_also.json
