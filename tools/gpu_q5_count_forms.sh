#!/bin/bash
# q5_count_kernel forms A/B on one box, alternating (experimental build: FLOCKGPU_Q5_COUNT = wg | wave1 | wave2 | wave4 ...): FORMS="wg wave1 wave4"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${TAG:-q5_forms}; mkdir -p $OUT
cp flock_amd/libflockgpu.so /tmp/shipped.so
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
show='import json,sys
d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}
print(sys.argv[1], "ms/step", d["ms_per_step"], "count", r.get("avg_launch_ms"), "frac", r.get("frac"), "rows", d["config"].get("result_rows"), {k: round(v, 4) for k, v in (r.get("kernels_ms") or {}).items()})'
for round in 1 2 ${ROUNDS}; do for form in ${FORMS:-wg wave1 wave2 wave4}; do
  FLOCKGPU_Q5_COUNT=$form FLOCK_BENCH_VERBOSE=1 python bench.py --query 5 --no-also --no-cpu --steps 20 --warmup 5 2>$OUT/$form.err | tail -1 > $OUT/$form.json
  python - "$form" <<PY
import json, sys
try:
    full = [l for l in open("$OUT/$form.err").read().splitlines() if l.startswith("{")]
    d = json.loads(full[-1]) if full else json.loads(open("$OUT/$form.json").read())
    r = d.get("roofline") or {}
    print(sys.argv[1], "ms/step", d["ms_per_step"], "count", r.get("avg_launch_ms"), "frac", r.get("frac"), "rows", d["config"].get("result_rows"), {k: round(v, 4) for k, v in (r.get("kernels_ms") or {}).items()})
except Exception as e:
    print(sys.argv[1], "failed", e); print(open("$OUT/$form.err").read()[-800:])
PY
done; done
if [ -n "$PARITY" ]; then
  for form in $PARITY; do
    FLOCKGPU_Q5_COUNT=$form timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_nexmark_goldens.py -q -m gpu -k "q5" -x -p no:cacheprovider 2>&1 | tail -3
  done
fi
cp /tmp/shipped.so flock_amd/libflockgpu.so
