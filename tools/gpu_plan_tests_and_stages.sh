cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_plan_round4.py tests/test_plan_round5.py tests/test_plan_boundary.py tests/test_stage_plans.py tests/test_plan_q4_q9_ysb.py tests/test_session_windows.py -q -m gpu --maxfail=20 -p no:cacheprovider 2>&1 | tail -30
timeout 300 python bench.py --only-side plan_stages > gpurun_out/r5j_plan_stages.json 2>/dev/null
python - <<'PY'
import json
s=json.load(open('gpurun_out/r5j_plan_stages.json'))
for q in ("q3","q5","q8"):
    e=s[q]; print(q, e["whole_plan_ms"], e["staged_ms"], e["one_instance_ms"], e["on_device_ms"], e["on_device_over_whole"], e["top_kernels_ms_per_run"])
PY
