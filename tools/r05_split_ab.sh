#!/bin/bash
# usage (GPU box): tools/r05_split_ab.sh OUT  -> single-view ms (bench.py --profile-run) for the projection split variants
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/${1:-gpurun_out/r05_split}; mkdir -p $O; cd $R
run() { tag=$1; shift; for rep in 1 2; do env "$@" python bench.py --profile-run --steps 80 --warmup 20 2>/dev/null | tail -1 | sed "s/^/$tag rep$rep /" >> $O/ab.txt; done; }
run split_off GSPLAT_AMD_SPLIT_PROJECTION=0
run split_on GSPLAT_AMD_SPLIT_PROJECTION=1
run split_nostage GSPLAT_AMD_SPLIT_PROJECTION=1 GSPLAT_AMD_RECORDS_NOSTAGE=1
run split_blocks256 GSPLAT_AMD_SPLIT_PROJECTION=1 GSPLAT_AMD_RECORDS_BLOCKS=256
run split_blocks512 GSPLAT_AMD_SPLIT_PROJECTION=1 GSPLAT_AMD_RECORDS_BLOCKS=512
run split_blocks1024 GSPLAT_AMD_SPLIT_PROJECTION=1 GSPLAT_AMD_RECORDS_BLOCKS=1024
run split_blocks2048 GSPLAT_AMD_SPLIT_PROJECTION=1 GSPLAT_AMD_RECORDS_BLOCKS=2048
cat $O/ab.txt
