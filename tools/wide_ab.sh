#!/bin/bash
# usage (GPU box): tools/wide_ab.sh "ENV=.. ENV=.." ...  -> per environment: the headline, a view alone, the backward sweep, and the
# wide_depth / rotating passes with their stage times (what the hot-gaussian replicas are for: GSPLAT_AMD_HOT_SPLATS=0 / 1)
for envs in "$@"; do
  env $envs python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-train-step 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$envs] ms/view', d['config']['ms_per_view'], 'fixed', d['ms_per_view_fixed_cameras'], 'single', d['single_view']['ms'], 'fwd', d['stage_ms']['gsa_composite_fwd'], 'bwd', d['stage_ms']['gsa_composite_bwd'], 'pbwd', d['stage_ms']['gsa_project_bwd'], 'wide', d['wide_depth']['ms_per_view'], d['wide_depth']['stage_ms'], 'rot', d['rotating']['ms_per_view'])"
done
