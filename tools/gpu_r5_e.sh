#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_comm_ipc.py tests/test_bench_contract.py tests/test_gpu_comm.py -q -m gpu --maxfail=20 -p no:cacheprovider 2>&1 | tail -70 > gpurun_out/r5e_tests.log
tail -n 50 gpurun_out/r5e_tests.log
