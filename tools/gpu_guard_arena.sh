#!/bin/bash
# gpurun helper: the GPU tests with every buffer of the library's own arena in guarded memory (experimental build,
# FLOCKGPU_GUARD_ARENA=1: exactly the bytes the host code asked for, ending where mapped address space ends) -- a kernel that reads or
# writes past what was requested for it faults.  Runs under pytest-xdist (one worker) so that a faulting test is reported as a crashed
# worker and the run goes on.  (The multi-process exchange tests are left out: hipIpcGetMemHandle does not export the virtual-memory mappings
# guarded buffers are made of -- the ranks' send buffers live in the arena.)  TESTS="tests/test_gpu_parity.py ..." narrows it.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out/${TAG:-guard_arena}; mkdir -p $OUT
cp flock_amd/libflockgpu.so /tmp/shipped.so
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
FLOCKGPU_GUARD_ARENA=1 timeout ${TEST_TIMEOUT:-2400} python -m pytest ${TESTS:-tests} -q -m gpu -p no:cacheprovider -n 1 --deselect tests/test_abi_symbols.py --deselect tests/test_gpu_comm_ipc.py --deselect tests/test_bench_contract.py::test_two_ranks_on_one_gpu_run_the_configured_workload_and_the_exchange > $OUT/tests.log 2>&1
echo "tests rc=$?"
cp /tmp/shipped.so flock_amd/libflockgpu.so
grep -n 'crashed\|FAILED\|ERROR\|passed\|failed' $OUT/tests.log | tail -40
