#!/bin/bash
# usage (on the GPU box): tools/round_artifacts.sh TAG  -> gpurun_out/TAG/: default bench line, rocprof kernel stats of the
# same command, PMC traffic passes (fetch / write) -- the inputs of profiles/<TAG>_*.  Copy what should be judged.
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
bash tools/hb.sh python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
echo "rocprof stats done"
# the kernels ALONE: single-view steps on one stream (in the multi-view step above a kernel's duration includes the time it
# spends waiting beside another stream's sweep)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_solo -o p -- python3 $R/bench.py --profile-run --steps 30 --warmup 5 > $O/bench_solo.json 2> $O/bench_solo.err
echo "rocprof solo stats done"
cd $R
python tools/prof_summary.py $O/stats/p_kernel_stats.csv > $O/summary_table.md
python tools/prof_summary.py $O/stats_solo/p_kernel_stats.csv > $O/summary_table_solo.md
echo "summary done"
bash tools/collect_pmc.sh gpurun_out/$tag/pmc > $O/pmc.log 2>&1
echo "pmc done"
# the VALU roofline passes of the two sweeps (lanes / busy / lds / mix + the kernels' durations in the same protocol)
bash tools/r05_valu_pmc.sh gpurun_out/$tag/valu > $O/valu.log 2>&1
echo "valu pmc done"
python tools/make_traffic_json.py $O/pmc "$tag" $O/traffic.json $O/valu > $O/traffic.log
ls $O
