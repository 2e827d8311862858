cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3 4; do python bench.py --query 3 --no-also --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(d['ms_per_step'], r.get('kernel'), r.get('avg_launch_ms'), r.get('frac'))"; done
