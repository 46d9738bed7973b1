#!/bin/bash
# usage (on the GPU box): tools/prof_bench.sh NAME [bench args] -> gpurun_out/NAME/ kernel stats table of bench.py
name=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/$name/log.txt 2>&1 || exit 1
cd $R && python tools/prof_summary.py gpurun_out/$name/p_kernel_stats.csv > gpurun_out/$name/summary.md
sed -n 1,26p gpurun_out/$name/summary.md
