#!/bin/bash
# SQ counter passes (two sets of 8) for the kernels whose names match $2 under command $1; prints per-launch averages
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
CMD="$1"; PAT="$2"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$tag -- $CMD > /tmp/run_$tag.log 2>&1; tail -2 /tmp/run_$tag.log | cut -c1-300
  python - /tmp/pmc_$tag "$PAT" <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: [0, 0.0])
pat = re.compile(sys.argv[2])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = pat.search(r["Kernel_Name"])
        if not m: continue
        k = m.group(0)
        a = acc[(k, r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()): print(k, c, n, round(v / n, 1))
PY
done
