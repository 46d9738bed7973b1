#!/bin/bash
O=gpurun_out/r03_sweep2; mkdir -p $O
run() { # name, env, args
  env $2 python bench.py --no-cpu-baseline --views 1 --steps 60 $3 > $O/$1.json 2>/dev/null
  python - "$O/$1.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "ms/view", d["config"]["ms_per_view"], "single", d["single_view"]["ms"], "wait", d["host_wait_ms_per_step"])
PY
}
run b4s2 "A=1" "--views-per-gpu 4 --view-streams 2"
run b4s3 "A=1" "--views-per-gpu 4 --view-streams 3"
run b4s4 "A=1" "--views-per-gpu 4 --view-streams 4"
run b6s3 "A=1" "--views-per-gpu 6 --view-streams 3"
run b4s3_noreorder "GSPLAT_AMD_BWD_REORDER=0" "--views-per-gpu 4 --view-streams 3"
run b4s2_table "GSPLAT_AMD_DEPTH_TABLE=1" "--views-per-gpu 4 --view-streams 2"
run b4s2_again "A=1" "--views-per-gpu 4 --view-streams 2"
