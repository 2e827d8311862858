#!/bin/bash
# q5 A/B of environment knobs on the experimental build, alternating: VARIANTS="base FLOCKGPU_Q5_NO_ZB=1 ..." (a variant = one VAR=VALUE or "base")
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${TAG:-q5_env_ab}; mkdir -p $OUT
cp flock_amd/libflockgpu.so /tmp/shipped.so
cp flock_amd/libflockgpu_experimental.so flock_amd/libflockgpu.so
for round in 1 2 ${ROUNDS}; do for v in ${VARIANTS:-base}; do
  name=$(echo "$v" | tr '=' '_')
  if [ "$v" = "base" ]; then envs=""; else envs="$v"; fi
  env $envs FLOCK_BENCH_VERBOSE=1 python bench.py --query ${QUERY:-5} --no-also --no-cpu --steps 20 --warmup 5 2>$OUT/$name.err | tail -1 > $OUT/$name.json
  python - "$v" <<PY
import json, sys
try:
    full = [l for l in open("$OUT/$name.err").read().splitlines() if l.startswith("{")]
    d = json.loads(full[-1]) if full else json.loads(open("$OUT/$name.json").read())
    r = d.get("roofline") or {}
    print(sys.argv[1], "ms/step", d["ms_per_step"], "dominant", r.get("avg_launch_ms"), "frac", r.get("frac"), "rows", d["config"].get("result_rows"), {k: round(v, 4) for k, v in (r.get("kernels_ms") or {}).items()})
except Exception as e:
    print(sys.argv[1], "failed", e); print(open("$OUT/$name.err").read()[-800:])
PY
done; done
cp /tmp/shipped.so flock_amd/libflockgpu.so
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | tail -8; fi
