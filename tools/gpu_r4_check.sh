#!/bin/bash
# gpurun helper (round 4): every GPU test (all failures listed, not only the first), then the default bench line.
OUT=gpurun_out/${TAG:-r4a}; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -40 $OUT/tests.log
if [ -z "$NO_BENCH" ]; then
  ( time timeout 900 python bench.py > $OUT/bench.out 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.time
  tail -1 $OUT/bench.out
  cp gpurun_out/bench_also.json $OUT/bench_also.json 2>/dev/null
fi
