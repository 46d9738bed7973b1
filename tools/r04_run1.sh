#!/bin/bash
# (GPU box) r04: tests touched this round, mask A/B, the train step under rocprof
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04a; mkdir -p $O
python -m pytest tests/test_amp.py tests/test_hip_parity.py -x -q -m gpu -k "non_finite or bucketed or depth_table or reports_what or population or masks or random_shapes or one_pass or views_equal" > $O/tests.log 2>&1; tail -3 $O/tests.log
for m in 0.4 0.25 off; do
  if [ $m == off ]; then export GSPLAT_AMD_QUAD_MASKS=0; else export GSPLAT_AMD_QUAD_MASKS=1 GSPLAT_AMD_MASK_DEPTH=$m; fi
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --views 1 > $O/bench_$m.json 2> $O/bench_$m.err
  python - "$O/bench_$m.json" $m <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["ms_per_step"], d["single_view"]["ms"], d["stage_ms"])
PY
done
unset GSPLAT_AMD_QUAD_MASKS GSPLAT_AMD_MASK_DEPTH
( cd /tmp && export TMPDIR=/tmp && FUSED_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_prof -o p -- python3 $R/tools/bench_train_step.py > $O/train_prof.log 2>&1 ); echo "train prof rc=$?"; tail -1 $O/train_prof.log
python tools/prof_summary.py $O/train_prof/p_kernel_stats.csv 13 > $O/train_step_table.md; head -30 $O/train_step_table.md
