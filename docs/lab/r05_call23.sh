#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so
export MIDRANGE_SIZES=256,512,768,1023,1024,1536,2047,2048,3000,4095
out=gpurun_out/r05_ab_small_path_windows.txt
: > $out
for c in 0 5 6 7; do
echo "## small path, forced window width $c (0 = the layout's own: 5 below 1024 terms, 6 below 2048, 7 below 4096)" >> $out; C25519_MSM_CFORCE=$c timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
export MIDRANGE_SIZES=4096,8192,10000,12000,14000,16384
for c in 5 6; do
echo "## small path up to 65535 terms, $c-bit windows beyond 4095" >> $out; C25519_MSM_SMALL_MAX=65535 C25519_MSM_SMALL_C=$c timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
cat $out
