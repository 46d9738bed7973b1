#!/bin/bash
# (GPU box) r04 evidence besides tools/round_artifacts.sh: the driver's command, the boxes-off ablation (bench line + SQ
# counters of the sweeps), config C5 (bench line + kernel stats), the training-step leg with kernel stats, the parity
# evidence files.  Everything lands under gpurun_out/r04x; tools/make_r04_extras.py composes the profiles/ files from it.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04x; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "driver-style bench done"
GSPLAT_AMD_QUAD_BOXES=0 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-step > $O/bench_noboxes.json 2> $O/bench_noboxes.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-step > $O/bench_boxes.json 2> $O/bench_boxes.err
echo "boxes A/B done"
bash tools/pmc_bwd.sh gpurun_out/r04x/pmc_boxes > $O/pmc_boxes.log 2>&1
bash tools/pmc_bwd.sh gpurun_out/r04x/pmc_noboxes GSPLAT_AMD_QUAD_BOXES=0 > $O/pmc_noboxes.log 2>&1
echo "sweep counters done"
python bench.py --workload C5 --no-cpu-baseline --no-train-step --steps 20 --warmup 3 --views 1 > $O/bench_c5.json 2> $O/bench_c5.err
echo "C5 bench done"
GSPLAT_AMD_EVIDENCE=$O python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "config_c2 or config_c3 or stage" > $O/parity.log 2>&1 || true
tail -2 $O/parity.log
python tools/bench_train_step.py > $O/train_step.log 2>&1
tail -2 $O/train_step.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o p -- python3 $R/bench.py --workload C5 --profile-run --steps 10 --warmup 3 > $O/bench_c5_solo.json 2> $O/bench_c5_solo.err
echo "C5 stats done"
FUSED_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_train -o p -- python3 $R/tools/bench_train_step.py > $O/train_step_prof.log 2>&1
echo "train step stats done"
cd $R
python tools/prof_summary.py $O/stats_c5/p_kernel_stats.csv > $O/c5_table.md
python tools/prof_summary.py $O/stats_train/p_kernel_stats.csv > $O/train_table.md
ls $O
