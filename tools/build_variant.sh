#!/bin/bash
# usage: tools/build_variant.sh NAME -DFLAG -DFLAG ... (separate arguments)  -> 3d-gaussian-splatting_amd/lib/ab_NAME.so (same sources, extra hipcc flags)
set -e
name=$1; shift
R=$(cd $(dirname $0)/.. && pwd); C=$R/3d-gaussian-splatting_amd/csrc; L=$R/3d-gaussian-splatting_amd/lib; T=/tmp/gsa_var_$name
rm -rf $T; mkdir -p $T
for f in abi project sort composite train_step view; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -I$R/include -I$C "$@" -c $C/$f.hip -o $T/$f.o &
done
wait; for f in abi project sort composite train_step view; do [ -f $T/$f.o ] || { echo "compile of $f failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $T/*.o -o $L/ab_$name.so
echo built $L/ab_$name.so
