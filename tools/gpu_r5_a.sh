#!/bin/bash
# round 5, first GPU call: the new plan-path tests, the plan suites the changes touch, the arch harness rows
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_plan_round5.py -q -m gpu --maxfail=12 -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r5a_tests_new.log
timeout 900 python -m pytest tests/test_plan_round4.py tests/test_plan_boundary.py tests/test_stage_plans.py tests/test_plan_q4_q9_ysb.py tests/test_session_windows.py -q -m gpu --maxfail=12 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r5a_tests_plan.log
timeout 600 python bench.py --only-side arch > gpurun_out/r5a_arch.json 2> gpurun_out/r5a_arch.err
tail -5 gpurun_out/r5a_tests_new.log gpurun_out/r5a_tests_plan.log
tail -c 1500 gpurun_out/r5a_arch.err
head -c 6000 gpurun_out/r5a_arch.json
