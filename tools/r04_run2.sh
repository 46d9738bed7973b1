#!/bin/bash
# (GPU box) r04: full GPU suite, C5 with the 12-bit LSD passes (bench + rocprof), loss kernel alone
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
O=$R/gpurun_out/r04b; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --workload C5 --no-cpu-baseline --steps 20 --warmup 3 --views 1 > $O/c5.json 2> $O/c5.err; echo "c5 rc=$?"
python - "$O/c5.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("C5", d["ms_per_step"], d["single_view"]["ms"], d["stage_ms"])
PY
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_prof -o p -- python3 $R/bench.py --workload C5 --profile-run --steps 10 --warmup 3 > $O/c5_prof.log 2>&1 ); echo "c5 prof rc=$?"
python tools/prof_summary.py $O/c5_prof/p_kernel_stats.csv > $O/c5_table.md; head -28 $O/c5_table.md
