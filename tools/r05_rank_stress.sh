#!/bin/bash
# usage (GPU box): tools/r05_rank_stress.sh OUT  -> repeats bench.py's 4-rank x 2-view gloo rehearsal, with and without the r05 rank path
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/${1:-gpurun_out/r05_stress}; mkdir -p $O; cd $R
ulimit -c 0
export GSPLAT_AMD_BENCH_BACKEND=gloo
run() { tag=$1; shift; for rep in 1 2 3 4 5 6; do
  timeout -k 10 120 python bench.py --gpus 4 --workload T --steps 3 --warmup 1 --views-per-gpu 2 --view-streams 2 --ring 16 --no-extras --views 1 --no-cpu-baseline "$@" > $O/${tag}_$rep.out 2> $O/${tag}_$rep.err
  echo "$tag rep $rep rc=$?" >> $O/summary.txt
  grep -h "finalize: implausible\|AcceleratorError\|Memory access fault\|HIP error\|GsaError" $O/${tag}_$rep.err | head -3 >> $O/summary.txt
done; }
run rank_batch
run per_view --no-batch-backward
run rank_batch_1stream --view-streams 1
cat $O/summary.txt
