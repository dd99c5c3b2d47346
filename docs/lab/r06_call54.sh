#!/bin/bash
# round 6, gpurun call 54: the order in which the host enqueues verify_batch's two chains (VERIFY_ORDER 0 / 1 / 2), re-measured at the sizes the mid path now serves from
# 2048 signatures (the hash kernel of such a batch is 18 us: the timeline shows a 20 us gap behind it while the host is still launching the decompression)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_verify_order_small.txt; : > $out
for rep in 0 1; do for o in 0 1 2; do
  echo "## VERIFY_ORDER=$o rep $rep" >> $out
  C25519_HIP_LIB=$T C25519_VERIFY_ORDER=$o VERIFY_SIZES=2048,3072,4096,6143,8192,16384 timeout 300 python tools/verify_midrange.py 2>/dev/null >> $out
done; done
cat $out
