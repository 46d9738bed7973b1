#!/bin/bash
# usage: tools/r03_pmc2.sh TAG variant...  -> SQ counters of the compositor kernels per library variant
for v in "${@:2}"; do
  lib=$PWD/3d-gaussian-splatting_amd/lib/ab_$v.so; [ "$v" == "base" ] && lib=$PWD/3d-gaussian-splatting_amd/lib/libgsplat_amd.so
  bash tools/pmc_bwd.sh gpurun_out/$1/pmc_$v GSPLAT_AMD_LIB=$lib > gpurun_out/$1/pmc_$v.log 2>&1
  echo "== $v"; grep -A9 "composite_bwd_2p" gpurun_out/$1/pmc_$v/insts.txt gpurun_out/$1/pmc_$v/active.txt | grep -v "^--"
done
