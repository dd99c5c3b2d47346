#!/bin/bash
# round 6, gpurun call 60: the lockstep PAIRS of the mid path (mid_long.h fe_mul_chain_n<2>) with input pins (k_mid_acc_long 168 registers + scratch -> 148, half the s_nop) against
# the definition pins (libc25519hip_pairdef.so), same box, interleaved; parity of the MSM module first
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
L=$R/curve25519-dalek_amd/lib
timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_verify.py -x -q -m gpu > gpurun_out/r06_c60_tests.log 2>&1; tail -3 gpurun_out/r06_c60_tests.log
out=gpurun_out/r06_ab_pair_pins.txt; : > $out
for rep in 0 1 2; do for lib in tune pairdef; do
  echo "## $lib rep $rep" >> $out
  C25519_HIP_LIB=$L/libc25519hip_$lib.so MIDRANGE_SIZES=8192,16384,32768,65536,131072,262144 timeout 300 python tools/midrange_numbers.py 2>/dev/null | cut -c1-56 >> $out
done; done
cat $out
