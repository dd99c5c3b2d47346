#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so
export MIDRANGE_SIZES=2048,4000,4096,6000,8192,12000,16384,24000,32768,65535
out=gpurun_out/r05_ab_small_path_range.txt
: > $out
for rep in 0 1; do
echo "## default (small path up to 4095 terms), rep $rep" >> $out; timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## small path up to 65535 terms, 7-bit windows beyond 4095, rep $rep" >> $out; C25519_MSM_SMALL_MAX=65535 C25519_MSM_SMALL_C=7 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## small path up to 65535 terms, 6-bit windows beyond 4095, rep $rep" >> $out; C25519_MSM_SMALL_MAX=65535 C25519_MSM_SMALL_C=6 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
cat $out
