#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r05_ab_verify_small_range.txt
: > $out
echo "## small path up to 4095 terms (2047 signatures)" >> $out; timeout 300 python tools/verify_midrange.py 2>/dev/null >> $out
echo "## small path up to 12287 terms (6143 signatures)" >> $out; C25519_MSM_SMALL_MAX=12287 timeout 300 python tools/verify_midrange.py 2>/dev/null >> $out
cat $out
