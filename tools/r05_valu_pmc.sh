#!/bin/bash
# usage (GPU box): tools/r05_valu_pmc.sh <out-dir>  -> the VALU roofline counters of the two compositor sweeps:
# wave-instructions, cycles the vector ALU was busy, active lanes per instruction, in passes of their own (rocprofv3 --pmc with
# --kernel-trace only), on bench.py --profile-run (single-view steps on one stream, the protocol `roofline` is quoted on).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/r05_valu}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/list_avail.txt 2>&1 || true
run() { n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 3 --warmup 2 --profile-run > $OUT/$n.log 2>&1 || { echo "pass $n failed"; tail -5 $OUT/$n.log; return 0; }
  python3 $R/tools/pmc_summary.py $(find $OUT/$n -name '*counter_collection.csv' | head -1) "composite" > $OUT/$n.txt
  echo "pass $n done"
}
run lanes SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run busy SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run lds SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS
run mix SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU
# the kernels' durations in the same protocol (for cycles = duration x clock)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $R/bench.py --profile-run --steps 30 --warmup 5 > $OUT/stats.log 2>&1
python3 $R/tools/prof_summary.py $OUT/stats/p_kernel_stats.csv > $OUT/stats_table.md || true
cat $OUT/lanes.txt $OUT/busy.txt $OUT/lds.txt $OUT/mix.txt 2>/dev/null
