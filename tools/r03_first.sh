#!/bin/bash
# first GPU call of round 3: suite + bench matrix
O=gpurun_out/r03_first; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
for cfg in "" "--view-streams 1" "--views-per-gpu 1" "--views-per-gpu 8" "--views-per-gpu 2"; do
  name=$(echo "b$cfg" | tr -d ' -')
  timeout -k 10 300 python bench.py --no-cpu-baseline $cfg > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"
  python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["ms_per_step"], d["config"]["ms_per_view"], d["single_view"], d["warmup_effective"], d.get("rotating"), d["stage_ms"])
except Exception as e: print("$name parse failed", e)
PY
done
GSPLAT_AMD_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline > $O/gloo2.json 2> $O/gloo2.err; echo "gloo2 rc=$?"; tail -2 $O/gloo2.err; cut -c1-600 $O/gloo2.json
