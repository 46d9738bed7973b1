#!/bin/bash
# (GPU box) r04: the alpha exponent on the matrix pipe -- parity subset, then the bench line per library variant
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04m; mkdir -p $O
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "stage_composite or forward_backward or config_c2 or backward_sweep or cull_modes or depth_render or c3_density_crop" > $O/tests.log 2>&1; tail -3 $O/tests.log
for v in default "$@"; do
  lib=""; [ $v != default ] && lib=$R/3d-gaussian-splatting_amd/lib/ab_$v.so
  GSPLAT_AMD_LIB=$lib python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train-step > $O/bench_$v.json 2> $O/bench_$v.err
  python - "$O/bench_$v.json" $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["config"]["ms_per_view"], d["single_view"]["ms"], d["stage_ms"])
PY
done
