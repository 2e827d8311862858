cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
for v in "" "1"; do
  if [ -n "$v" ]; then export FLOCKGPU_Q5_NO_PRECLEAN=1; else unset FLOCKGPU_Q5_NO_PRECLEAN; fi
  python bench.py --query 5 --no-also --no-cpu --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('no_preclean=$v', d['ms_per_step'], 'count', r['avg_launch_ms'], 'range', r['kernels_ms'].get('q5_range_kernel'))"
done; done
