#!/bin/bash
# round 6, gpurun call 22: where the mid path ends now that its per-kernel events are gone: 2^18 terms through it (projective records / normaliser arm) against the bucket pipeline
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
export MIDRANGE_SIZES=131072,200000,262144,400000,524288
out=gpurun_out/r06_ab_mid_upper_end.txt; : > $out
for rep in 0 1; do
echo "## release boundaries (mid path up to 2^17 terms), rep $rep" >> $out; C25519_HIP_LIB=$T timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## mid path up to 2^19, projective records up to 2^19, rep $rep" >> $out; C25519_HIP_LIB=$T C25519_MSM_MID_MAX=524288 C25519_MID_PROJ_MAX=524288 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## mid path up to 2^19, normaliser + k_accumulate above 2^17, rep $rep" >> $out; C25519_HIP_LIB=$T C25519_MSM_MID_MAX=524288 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
cat $out
