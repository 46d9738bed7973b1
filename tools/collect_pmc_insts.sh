#!/bin/bash
# Instruction-mix PMC passes only (two rocprofv3 --pmc runs of a short bench).  usage (GPU box): tools/collect_pmc_insts.sh <out-dir>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$n.log 2>&1
}
run insts SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
run active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
cd $R
for n in insts active; do python3 tools/pmc_summary.py $(find $OUT/$n -name "*counter_collection.csv" | head -1) "composite"; done
