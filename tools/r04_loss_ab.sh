#!/bin/bash
# (GPU box) the loss kernel alone for a few strip geometries (library variants built by tools/build_variant.sh)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python tools/r04_loss_bench.py
for f in 3d-gaussian-splatting_amd/lib/ab_loss_*.so; do echo $f; GSPLAT_AMD_LIB=$R/$f python tools/r04_loss_bench.py; done
