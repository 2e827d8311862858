#!/bin/bash
mkdir -p gpurun_out/r3s
timeout 900 python -m pytest tests/test_stage_plans.py -x -q -m gpu > gpurun_out/r3s/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3s/tests.log
timeout 300 python tools/gpu_collect_breakdown.py > gpurun_out/r3s/breakdown.txt 2>&1; cat gpurun_out/r3s/breakdown.txt | grep -v amdgpu.ids
timeout 300 python bench.py --only-side plan_stages --steps 20 > gpurun_out/r3s/ps.out 2> gpurun_out/r3s/ps.err; tail -c 2500 gpurun_out/r3s/ps.out
