#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_ab_small_stride.txt
: > $out
export MIDRANGE_SIZES=256,1024,4096,8192,12000
T=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so; V=$PWD/curve25519-dalek_amd/lib/libc25519hip_s44.so
for rep in 0 1; do
echo "## table entries 44 words apart in LDS, rep $rep" >> $out; C25519_HIP_LIB=$V timeout 100 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## 40 words apart (as is), rep $rep" >> $out; C25519_HIP_LIB=$T timeout 100 python tools/midrange_numbers.py 2>/dev/null >> $out
done
cat $out
