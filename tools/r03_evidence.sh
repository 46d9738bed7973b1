#!/bin/bash
# (GPU box) everything behind profiles/r03_*: round artifacts (default bench + rocprof stats + PMC traffic), the C3 training
# step under rocprof, the C5 view, the driver-style short bench.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
( while true; do sleep 60; echo "[hb] $(date +%T)"; done ) & HB=$!
bash tools/round_artifacts.sh r03 > gpurun_out/r03_artifacts.log 2>&1; echo "artifacts rc=$?"
O=$R/gpurun_out/r03x; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "driver-style rc=$?"
python tools/bench_train_step.py > $O/train_step.log 2>&1; echo "train rc=$?"; tail -2 $O/train_step.log
( cd /tmp && export TMPDIR=/tmp && FUSED_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_prof -o p -- python3 $R/tools/bench_train_step.py > $O/train_prof.log 2>&1 ); echo "train prof rc=$?"
python tools/prof_summary.py $O/train_prof/p_kernel_stats.csv > $O/train_step_table.md
python bench.py --workload C5 --no-cpu-baseline --steps 20 --warmup 3 --views 1 > $O/c5.json 2> $O/c5.err; echo "c5 rc=$?"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_prof -o p -- python3 $R/bench.py --workload C5 --profile-run --steps 10 --warmup 3 > $O/c5_prof.log 2>&1 ); echo "c5 prof rc=$?"
python tools/prof_summary.py $O/c5_prof/p_kernel_stats.csv > $O/c5_table.md
kill $HB 2>/dev/null
ls $O
# sweep of views per step x streams (same box)
for b in 1 2 4 8; do for s in 1 2 3; do
  [ $b == 1 ] && [ $s != 1 ] && continue
  python bench.py --no-cpu-baseline --views 1 --steps 50 --views-per-gpu $b --view-streams $s > $O/sweep_${b}_${s}.json 2>/dev/null
  python - "$O/sweep_${b}_${s}.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "ms/view", d["config"]["ms_per_view"], "single", d["single_view"]["ms"])
PY
done; done
