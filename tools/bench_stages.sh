#!/bin/bash
# usage: tools/bench_stages.sh [bench args]  -> prints compact stage timings
python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu.ids | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['cull_mode'], d['value'], d['ms_per_step'], d['config']['intersections_M'], d['stage_ms'])"
