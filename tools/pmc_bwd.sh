#!/bin/bash
# usage (GPU box): tools/pmc_bwd.sh <out-dir> [env assignments...]  -> SQ counters of the compositor kernels for bench.py
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/pmc_bwd}; shift || true
for kv in "$@"; do export "$kv"; done
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 3 --warmup 2 --profile-run > $OUT/$n.log 2>&1
  python3 $R/tools/pmc_summary.py $(find $OUT/$n -name '*counter_collection.csv' | head -1) "composite" > $OUT/$n.txt
}
run insts SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
run active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
cat $OUT/insts.txt $OUT/active.txt
