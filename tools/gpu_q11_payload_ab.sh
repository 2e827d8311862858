#!/bin/bash
# q11's packed sort payload against the gather (experimental library knob): tests, then bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
show='import json,sys
d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}
print(d.get("ms_per_step"), r.get("kernel"), r.get("avg_launch_ms"), r.get("frac"), r.get("kernels_ms_per_step") or r.get("kernels_ms"))'
if [ "${TESTS:-1}" = "1" ]; then
timeout 900 python -m pytest tests/test_gpu_q11.py tests/test_next_goldens.py tests/test_session_windows.py tests/test_gpu_parity.py tests/test_nexmark_goldens.py -q -m gpu --maxfail=10 -p no:cacheprovider -k "q11 or q3 or session or group_rows" 2>&1 | tail -15
fi
echo "--- q11 shipped (packed payload)"
for i in 1 2; do timeout 300 python bench.py --only-side q11 --steps 5 --no-cpu 2>/dev/null | tail -1 | python -c "$show"; done
cp flock_amd/libflockgpu.so /tmp/shipped.so
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
echo "--- q11 gather (experimental knob)"
FLOCKGPU_Q11_GATHER=1 timeout 300 python bench.py --only-side q11 --steps 5 --no-cpu 2>/dev/null | tail -1 | python -c "$show"
cp /tmp/shipped.so flock_amd/libflockgpu.so
