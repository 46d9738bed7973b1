#!/bin/bash
# usage: tools/r03_exp.sh TAG  -> A/B bench lines for env / lib variants + PMC of the compositors
O=gpurun_out/$1; mkdir -p $O
line() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["value"], d["config"]["ms_per_view"], d["single_view"]["ms"], d["stage_ms"])
PY
}
GSPLAT_AMD_QUAD_MASKS=0 timeout -k 10 300 python bench.py --no-cpu-baseline --views 1 > $O/nomask.json 2> $O/nomask.err; line $O/nomask.json
timeout -k 10 300 python bench.py --no-cpu-baseline --views 1 > $O/mask.json 2> $O/mask.err; line $O/mask.json
for v in "$@"; do
  [ "$v" == "$1" ] && continue
  GSPLAT_AMD_LIB=$PWD/3d-gaussian-splatting_amd/lib/ab_$v.so timeout -k 10 300 python bench.py --no-cpu-baseline --views 1 > $O/$v.json 2> $O/$v.err; line $O/$v.json
done
bash tools/pmc_bwd.sh gpurun_out/$1/pmc > $O/pmc.log 2>&1; cat $O/pmc/insts.txt $O/pmc/active.txt | grep -A9 "composite_fwd\|composite_bwd_2p\|quadrant_mask"
