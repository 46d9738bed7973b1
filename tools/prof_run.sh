#!/bin/bash
# usage (GPU box): tools/prof_run.sh NAME [env assignments...] -> kernel stats table of `bench.py --profile-run` (single-view steps)
name=$1; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -o p -- python3 $R/bench.py --profile-run --steps 30 --warmup 5 > $R/gpurun_out/$name/log.txt 2>&1 || { tail -5 $R/gpurun_out/$name/log.txt; exit 1; }
cd $R && python tools/prof_summary.py gpurun_out/$name/p_kernel_stats.csv > gpurun_out/$name/summary.md
sed -n 1,24p gpurun_out/$name/summary.md
