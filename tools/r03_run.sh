#!/bin/bash
# usage: tools/r03_run.sh TAG [pytest -k expr]  -> targeted tests, full suite, bench line
O=gpurun_out/$1; mkdir -p $O
if [ -n "$2" ]; then
  timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "$2" -s > $O/pytest_k.log 2>&1; rc=$?; echo "pytest -k rc=$rc"; grep -i "tightness\|passed\|failed\|Error" $O/pytest_k.log | tail -15
  [ $rc -ne 0 ] && tail -40 $O/pytest_k.log && exit 1
fi
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -ne 0 ] && tail -60 $O/pytest.log && exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["config"]["ms_per_view"], d["single_view"], d.get("rotating"))
print(d["stage_ms"])
PY
