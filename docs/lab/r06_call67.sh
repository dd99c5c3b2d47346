#!/bin/bash
# round 6, gpurun call 67: the lazy sign in the mid path's bucket lanes (mid.hip mid_acc_body; k_mid_acc_long 1262 -> 1242 vector instructions per addition, no select): parity of the
# MSM / verify / extra / debug-bounds modules, then a same-box A/B against libc25519hip_midsel.so (-DC25519_ACC_SIGN_SELECT in mid.hip only)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
L=$R/curve25519-dalek_amd/lib
timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_verify.py tests/test_gpu_extra.py tests/test_gpu_debug_bounds.py tests/test_gpu_raw160.py -x -q -m gpu > gpurun_out/r06_c67_tests.log 2>&1; tail -3 gpurun_out/r06_c67_tests.log
out=gpurun_out/r06_ab_lazy_sign_mid.txt; : > $out
for rep in 0 1 2; do for lib in tune midsel; do
  echo "## $lib rep $rep" >> $out
  C25519_HIP_LIB=$L/libc25519hip_$lib.so MIDRANGE_SIZES=8192,16384,32768,65536,131072,262144 timeout 300 python tools/midrange_numbers.py 2>/dev/null | cut -c1-56 >> $out
done; done
cat $out
