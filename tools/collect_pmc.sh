#!/bin/bash
# Collect per-kernel PMC counters for bench.py in separate passes (one counter group per run).
# usage (GPU box): tools/collect_pmc.sh <out-dir>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 3 --warmup 2 --profile-run > $OUT/$n.log 2>&1
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/${n}_cal -- python3 $R/tools/pmc_calibrate.py > $OUT/${n}_cal.log 2>&1 || true
}
run_step() { # name, counters...: the 4-view step on one stream (the per-step projection launches are among its kernels)
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 3 --warmup 2 --profile-run --profile-views 4 > $OUT/$n.log 2>&1
}
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_EA0_ATOMIC_sum
run_step fetch_step FETCH_SIZE
run_step write_step WRITE_SIZE TCC_EA0_ATOMIC_sum
run insts SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
run active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
ls $OUT
