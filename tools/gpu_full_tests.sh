#!/bin/bash
# gpurun helper: the whole GPU suite (as the driver runs it), the summary lines only
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/${TAG:-full}
timeout ${TEST_TIMEOUT:-2400} python -m pytest tests/ -q -m gpu -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/${TAG:-full}/tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG:-full}/tests.log | tail -15
