#!/bin/bash
# usage: tools/r03_direct.sh TAG  -> the binning tests, then single-view stage timings with the two-pass / one-pass tile binning
O=gpurun_out/$1; mkdir -p $O
tools/hb.sh timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "one_pass or fused_binning or bucketed or depth_table" > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log; [ $rc -ne 0 ] && exit $rc
for d in 0 1; do
  GSPLAT_AMD_BIN_DIRECT=$d timeout -k 10 300 python bench.py --no-cpu-baseline --views 1 --views-per-gpu 1 --steps 100 > $O/direct$d.json 2> $O/direct$d.err || exit 1
  python - "$O/direct$d.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d["single_view"]["ms"], d["stage_ms"], d["config"]["grad_l1_single_view"])
PY
done
