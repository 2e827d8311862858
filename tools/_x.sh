cd "$GRAFT_REPO_ROOT"
for y in 3 2 1 0; do
  for secs in 1000 100; do
    FLOCKGPU_Q3_YSHIFT=$y python bench.py --query 3 --seconds $secs --no-also --no-cpu --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('yshift $y secs $secs', d['ms_per_step'], d['roofline']['kernels_ms'].get('q3_build_kernel'))"
  done
done
