#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "q5" > $O/q5_tests.log 2>&1; echo "q5 tests rc=$?"; tail -15 $O/q5_tests.log
timeout 400 python bench.py --only-general q5_uniform --steps 5 > $O/q5_uniform.out 2> $O/q5_uniform.err; echo "q5_uniform rc=$?"; tail -3 $O/q5_uniform.err
tail -1 $O/q5_uniform.out | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline'] and d['roofline']['kernels_ms'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-also --no-cpu > $O/q5.out 2> $O/q5.err; echo "q5 rc=$?"; tail -1 $O/q5.out | cut -c1-400
