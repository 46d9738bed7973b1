#!/bin/bash
# views per step x streams with the deferred speculation check (alternate streams)
O=gpurun_out/r03_sweep3; mkdir -p $O
run() { # name, args
  timeout -k 10 300 python bench.py --no-cpu-baseline --views 1 --steps 60 --overlap alternate $2 > $O/$1.json 2>/dev/null
  python - "$O/$1.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "ms/view", d["config"]["ms_per_view"], "single", d["single_view"]["ms"], "wait", d["host_wait_ms_per_step"], flush=True)
PY
}
run b4s2 "--views-per-gpu 4 --view-streams 2"
run b4s3 "--views-per-gpu 4 --view-streams 3"
run b4s4 "--views-per-gpu 4 --view-streams 4"
run b6s3 "--views-per-gpu 6 --view-streams 3"
run b8s2 "--views-per-gpu 8 --view-streams 2"
run b8s4 "--views-per-gpu 8 --view-streams 4"
run b2s2 "--views-per-gpu 2 --view-streams 2"
run b4s2_nodefer "--views-per-gpu 4 --view-streams 2 --no-defer-check"
