#!/bin/bash
# A/B builds: tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>  ->  curve25519-dalek_amd/lib/libc25519hip_<name>.so
# (the TUNING objects of every other translation unit are reused -- a variant library honours the C25519_* knobs; select the library with C25519_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/../curve25519-dalek_amd/csrc"
name=$1; src=$2; shift 2
make -s -j8 tune
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -Wno-unused-value -Wno-unused-result -DC25519_TUNING"
mkdir -p ab
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o ab/${src%.hip}_$name.o
objs=""
for o in kernels finish capi msm msm_sort msm_sort_matrix verify small reduce single extra accum mid diag; do
  if [ "$o" = "${src%.hip}" ]; then objs="$objs ab/${o}_$name.o"; else objs="$objs tune/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libc25519hip_$name.so $objs
echo built ../lib/libc25519hip_$name.so
