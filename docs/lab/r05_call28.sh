#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_ab_small_reduce_wide.txt
: > $out
export MIDRANGE_SIZES=1024,2048,4096,8192,12000,12287
export C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so
for rep in 0 1; do
echo "## k_small_reduce with 512 threads above 256 partials + tree over the waves' sums, rep $rep" >> $out; timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## 256 threads, wave sums added one after the other (before), rep $rep" >> $out; C25519_SMALL_REDUCE_WIDE=0 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
cat $out
unset C25519_HIP_LIB
( timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -3 )
