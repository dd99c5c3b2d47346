#!/bin/bash
# round 6, gpurun call 55: VERIFY_ORDER 0 / 1 with the keys' CACHED points (the VerifyingKey case), 2048 .. 65536 signatures
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_verify_order_small_points.txt; : > $out
for rep in 0 1 2; do for o in 0 1; do
  echo "## cached key points, VERIFY_ORDER=$o rep $rep" >> $out
  C25519_HIP_LIB=$T C25519_VERIFY_ORDER=$o VERIFY_POINTS=1 VERIFY_SIZES=2048,3072,4096,6143,8192,16384,32768,65536 timeout 300 python tools/verify_midrange.py 2>/dev/null | cut -c1-40 >> $out
done; done
cat $out
