#!/bin/bash
# (GPU box) one process, the rank-form step of the tiny workload on a moving ring, with one side stream made to lag far behind
# the other (GSPLAT_AMD_DEBUG_LAG): a cross-stream lifetime bug shows up deterministically instead of once in a while
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05y
A="--workload T --steps 6 --warmup 1 --views-per-gpu 2 --view-streams 2 --ring 16 --no-train-step --no-extras --no-cpu-baseline --views 1 --no-batch-backward"
for lag in "1:20000000" "0:20000000" "1:200000000"; do
  GSPLAT_AMD_DEBUG_BINS=1 GSPLAT_AMD_DEBUG_LAG=$lag timeout -k 10 400 python bench.py $A > gpurun_out/r05y/out_$lag.json 2> gpurun_out/r05y/err_$lag.txt; echo "lag $lag rc=$?"; grep -c implausible gpurun_out/r05y/err_$lag.txt; grep -v "amdgpu.ids" gpurun_out/r05y/err_$lag.txt | tail -4 | cut -c1-400
done
