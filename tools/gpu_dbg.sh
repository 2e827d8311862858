#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/dbg; mkdir -p $O
timeout 600 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex bt -ex "info threads" --args python -m pytest tests/test_gpu_parity.py -x -q -k "five_launch" > $O/gdb.log 2>&1
echo "gdb rc=$?"; grep -n "SIGABRT\|signal\|#[0-9]" $O/gdb.log | head -60
