#!/bin/bash
# usage: tools/r03_ab_multi.sh TAG variant...  -> the default 4-view two-stream bench per library variant ("base" = in-tree)
O=gpurun_out/$1; mkdir -p $O; shift
for v in "$@"; do
  lib=$PWD/3d-gaussian-splatting_amd/lib/ab_$v.so; [ "$v" == "base" ] && lib=$PWD/3d-gaussian-splatting_amd/lib/libgsplat_amd.so
  GSPLAT_AMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --views 1 --steps 100 > $O/$v.json 2> $O/$v.err
  python - "$O/$v.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "ms/view", d["config"]["ms_per_view"], "single", d["single_view"]["ms"], d["stage_ms"], flush=True)
PY
done
