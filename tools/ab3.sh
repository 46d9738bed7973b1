#!/bin/bash
# usage: tools/ab3.sh "bench args A" "bench args B" ...  -> per argument set: Mpix/s, ms per view, single view, stages (twice, alternating)
for round in 1 2; do
for a in "$@"; do
  lib=$(echo "$a" | grep -o 'LIB=[^ ]*' | cut -d= -f2); rest=$(echo "$a" | sed 's/LIB=[^ ]*//')
  if [ -n "$lib" ]; then export GSPLAT_AMD_LIB=$PWD/3d-gaussian-splatting_amd/lib/$lib; else unset GSPLAT_AMD_LIB; fi
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step $rest 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$a]', d['value'], 'ms/view', d['config']['ms_per_view'], 'single', d['single_view']['ms'], {k: round(v,4) for k,v in d['stage_ms'].items()})"
done
done
