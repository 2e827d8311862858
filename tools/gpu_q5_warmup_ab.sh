cd "${GRAFT_REPO_ROOT:-/root/repo}"
for args in "--steps 20 --warmup 5" "--steps 50 --warmup 30" "--steps 20 --warmup 5" "--steps 50 --warmup 30" "--steps 100 --warmup 50"; do
  python bench.py --query 5 --no-also --no-cpu $args 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$args', 'step', d['ms_per_step'], 'count', r['avg_launch_ms'], r['frac'], r.get('launch_ms_spread'))"
done
