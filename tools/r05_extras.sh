#!/bin/bash
# (GPU box) r05 evidence besides tools/round_artifacts.sh: the driver's command, config C5 (bench line + kernel stats), the
# training-step leg with kernel stats and its host share, the parity evidence files (incl. the C2 step of 4 cameras), the
# careful-path statistics of the forward walk.  Everything lands under gpurun_out/r05x; tools/make_r05_extras.py composes
# the profiles/ files from it.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r05x; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "driver-style bench done"
python bench.py --workload C5 --no-cpu-baseline --no-train-step --steps 20 --warmup 3 --views 1 > $O/bench_c5.json 2> $O/bench_c5.err
echo "C5 bench done"
GSPLAT_AMD_EVIDENCE=$O python -m pytest tests/test_hip_parity.py -x -q -m gpu -p no:cacheprovider -k "config_c2 or config_c3 or stage" > $O/parity.log 2>&1 || true
tail -2 $O/parity.log
python tools/bench_train_step.py > $O/train_step.log 2>&1
tail -2 $O/train_step.log
python tools/train_step_host.py 60 > $O/train_host.txt 2>&1 || true
GSPLAT_AMD_VIEW_CALLS=0 python tools/train_step_host.py 60 > $O/train_host_nochain.txt 2>&1 || true
python tools/analyze_careful.py 300 4 > $O/careful.txt 2>&1 || true
echo "host + careful done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o p -- python3 $R/bench.py --workload C5 --profile-run --steps 10 --warmup 3 > $O/bench_c5_solo.json 2> $O/bench_c5_solo.err
echo "C5 stats done"
FUSED_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -o p -- python3 $R/tools/bench_train_step.py > $O/train_step_prof.log 2>&1
echo "train step stats done"
cd $R
python tools/prof_summary.py $O/stats_c5/p_kernel_stats.csv > $O/c5_table.md
python tools/prof_summary.py $O/stats_train/p_kernel_stats.csv > $O/train_table.md
ls $O
