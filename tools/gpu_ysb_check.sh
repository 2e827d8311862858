#!/bin/bash
# gpurun helper: YSB parity + the ysb side entry of the bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
(timeout 600 python -m pytest tests/test_gpu_ysb.py tests/test_plan_q4_q9_ysb.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3)
python - <<'PY'
import bench
from flock_amd import GpuContext
ctx = GpuContext(0)
for i in range(2):
    e = bench.ysb_side(ctx, 1_000_000, 10, True, None)
    print("ysb", e["ms_per_step"], e["value"], e["roofline"]["frac"], e["roofline"].get("kernels_ms"))
PY
