#!/bin/bash
# usage: tools/ab.sh libA.so libB.so ...   -> compact stage timings of bench.py per library (same C ABI)
for lib in "$@"; do
  echo "== $lib"
  GSPLAT_AMD_LIB=$PWD/$lib tools/bench_stages.sh || exit 1
done
