#!/bin/bash
# usage (GPU box): tools/prof_script.sh NAME KERNEL_REGEX script.py [args]  -> avg duration (us) of matching kernels under rocprofv3
name=$1; pat=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -o p -- python3 "$@" > $R/gpurun_out/$name/log.txt 2>&1 || exit 1
cd $R && python tools/prof_summary.py gpurun_out/$name/p_kernel_stats.csv | grep -E "$pat"
