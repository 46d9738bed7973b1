#!/bin/bash
# usage: tools/r03_prof_variants.sh TAG "kernel name pattern" variant...  -> rocprof averages of the matching kernels per library variant ("base" = in-tree)
tag=$1; pat=$2; shift; shift
for v in "$@"; do
  lib=$PWD/3d-gaussian-splatting_amd/lib/ab_$v.so; [ "$v" == "base" ] && lib=$PWD/3d-gaussian-splatting_amd/lib/libgsplat_amd.so
  echo "== $v"; tools/prof_run.sh $tag/$v GSPLAT_AMD_LIB=$lib | grep -i "$pat"
done
