#!/bin/bash
# usage: tools/ab2.sh libA.so libB.so ...  (paths under 3d-gaussian-splatting_amd/lib/; "-" = the default library)
# -> per library: Mpix/s, ms per view (4-view step), single-view ms, stage times -- same box, back to back, twice
for round in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset GSPLAT_AMD_LIB; else export GSPLAT_AMD_LIB=$PWD/3d-gaussian-splatting_amd/lib/$lib; fi
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-train-step 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['value'], 'ms/view', d['config']['ms_per_view'], 'single', d['single_view']['ms'], {k: round(v,4) for k,v in d['stage_ms'].items()})"
done
done
