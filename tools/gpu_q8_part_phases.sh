#!/bin/bash
# gpurun helper: q8's grouped hash path -- the tests, then also.q8_hash with the shipped library and with the bucket join cut
# short after each of its phases (experimental builds: FLOCKGPU_BUILD_DEFINES="-DQ8_JSTOP=n": the bucket join without seller inserts (1), up to "is the id a seller" (2), without the flag stores (3))
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out/${TAG:-q8part}; mkdir -p $OUT
[ -n "$SKIP_TESTS" ] || timeout 600 python -m pytest tests/test_gpu_parity.py -k q8 -q -m gpu -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
cp flock_amd/libflockgpu.so /tmp/shipped.so
for lib in libflockgpu.so $LIBS; do
  src=flock_amd/$lib; [ "$lib" = "libflockgpu.so" ] && src=/tmp/shipped.so
  cp $src flock_amd/libflockgpu.so
  timeout 300 python bench.py --only-general q8_hash --steps 5 2>$OUT/$lib.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$lib', 'ms/step', d['ms_per_step'], r.get('kernels_ms'))"
done
cp /tmp/shipped.so flock_amd/libflockgpu.so
