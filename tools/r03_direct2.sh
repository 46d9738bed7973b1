#!/bin/bash
# usage: tools/r03_direct2.sh TAG -> binning tests, per-kernel profile of the one-pass binning, phase traces, A/B of the block -> XCD mapping
O=gpurun_out/$1; mkdir -p $O
tools/hb.sh timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "one_pass or fused_binning" > $O/tests.log 2>&1; rc=$?
tail -3 $O/tests.log; [ $rc -ne 0 ] && exit $rc
tools/prof_run.sh $1/prof GSPLAT_AMD_BIN_DIRECT=1 | grep -i "emit\|colscan\|scatter\|hist\|scan_rows\|bsearch\|bucket" 
for d in 1 0; do
GSPLAT_AMD_BIN_DIRECT=$d GSPLAT_AMD_LIB=$PWD/3d-gaussian-splatting_amd/lib/ab_trace.so timeout -k 10 300 python tools/trace_sort.py > $O/trace$d.txt 2>&1 && grep -A1 "emit_" $O/trace$d.txt
done
tools/r03_ab.sh $1 "GSPLAT_AMD_BIN_DIRECT=1" base noxcd
tools/r03_ab.sh $1/two "GSPLAT_AMD_BIN_DIRECT=0" base
