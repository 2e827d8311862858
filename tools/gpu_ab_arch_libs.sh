#!/bin/bash
# gpurun helper: A/B of library variants on one box over the arch/ops harness rows: LIBS="libflockgpu.so libflockgpu_experimental_x.so" KERNEL=utf8_emit_long_kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
cp flock_amd/libflockgpu.so /tmp/shipped.so
for round in 1 2; do for lib in $LIBS; do
  src=flock_amd/$lib; [ "$lib" = "libflockgpu.so" ] && src=/tmp/shipped.so
  cp $src flock_amd/libflockgpu.so
  python bench.py --only-side arch 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d.get('arch_ops',d)
for row in '${ROWS:-join}'.split():
    m=a[row]['generic']; print('$lib', row, 'ms', m.get('ms_per_execute'), '${KERNEL}', (m.get('kernels_ms_per_execute') or {}).get('${KERNEL}'), m.get('error'))"
done; done
cp /tmp/shipped.so flock_amd/libflockgpu.so
