#!/bin/bash
# round 6, gpurun call 42: key BYTES -- A_i and R_i decompressed by one launch up to VERIFY_BOTH_MAX signatures (4096 = rounds 3-5)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_verify_both.txt; : > $out
for rep in 0 1; do
for m in 4096 65536 262144; do
echo "## VERIFY_BOTH_MAX=$m, rep $rep" >> $out
C25519_HIP_LIB=$T C25519_VERIFY_BOTH_MAX=$m VERIFY_SIZES=6144,8192,16384,32768,65536,131072 timeout 300 python tools/verify_midrange.py 2>/dev/null | cut -c1-40 >> $out
done
done
cat $out
