#!/bin/bash
# round 3, first GPU call: the bench contract on hardware
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r3a
python -m pytest tests/test_bench_contract.py -m gpu -x -q > gpurun_out/r3a/contract_test.log 2>&1; echo "contract rc=$?"
( time python bench.py ) > gpurun_out/r3a/bench_default.out 2> gpurun_out/r3a/bench_default.err; echo "bench rc=$?"
tail -c 4200 gpurun_out/r3a/bench_default.out | tail -1 | wc -c
cp gpurun_out/bench_also.json gpurun_out/r3a/bench_also_default.json
FLOCK_BENCH_SHARED_GPU=1 FLOCK_BENCH_EXCHANGE_TIMEOUT=120 timeout 600 python bench.py --gpus 2 --seconds 100 --steps 3 --warmup 1 --no-cpu > gpurun_out/r3a/bench_2rank_shared.out 2> gpurun_out/r3a/bench_2rank_shared.err; echo "2rank rc=$?"
tail -1 gpurun_out/r3a/bench_2rank_shared.out | head -c 1500
tail -5 gpurun_out/r3a/contract_test.log
