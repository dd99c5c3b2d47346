#!/bin/bash
# round 6, gpurun call 23: buckets per lane in level A of the bucket reduction (RED_LB_MIN) at mid sizes, now that the reduction is 40 % of a mid-size call
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
export MIDRANGE_SIZES=16384,32768,65536,131072,262144,1048576
out=gpurun_out/r06_ab_reduce_lb.txt; : > $out
for rep in 0 1; do
for lb in 1 2 3; do
echo "## RED_LB_MIN=$lb (at least 2^$lb buckets per lane in level A), rep $rep" >> $out; C25519_HIP_LIB=$T C25519_RED_LB_MIN=$lb timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
done
cat $out
