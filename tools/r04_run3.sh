#!/bin/bash
# (GPU box) r04: full GPU suite, qbox masks A/B
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04d; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
for m in 1 0; do
  GSPLAT_AMD_QUAD_BOXES=$m python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-step > $O/bench_$m.json 2> $O/bench_$m.err
  python - "$O/bench_$m.json" $m <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("masks", sys.argv[2], d["ms_per_step"], d["single_view"]["ms"], d["stage_ms"], d["rotating"]["ms_per_view"], d.get("wide_depth"))
PY
done
