#!/bin/bash
# (GPU box) shader / memory clock and power of GPU 0 sampled while bench.py's 4-view steps run: what clock do the sweeps get?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python bench.py --steps 1500 --warmup 10 --no-cpu-baseline --no-train-step > gpurun_out/clock_bench.json 2>/dev/null &
BP=$!
sleep 14
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo
  sleep 0.4
done
wait $BP
python -c "
import json
d=json.loads(open('gpurun_out/clock_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['ms_per_view'])"
