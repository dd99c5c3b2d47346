#!/bin/bash
# round 6, gpurun call 71: mid-size verify_batch (k_accumulate_long: prepared records, a latency-bound call) with the sign by operand selection ('tune', now the form of that kernel)
# against the lazy sign ('longlazy', what call 69's tree had); parity of the verify / msm modules first
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
L=$R/curve25519-dalek_amd/lib
timeout 2400 python -m pytest tests/test_gpu_verify.py tests/test_gpu_msm.py tests/test_gpu_ffi.py -x -q -m gpu > gpurun_out/r06_c71_tests.log 2>&1; tail -3 gpurun_out/r06_c71_tests.log
out=gpurun_out/r06_ab_lazy_sign_long.txt; : > $out
for rep in 0 1 2; do for lib in tune longlazy; do
  echo "## $lib rep $rep (cached key points)" >> $out
  C25519_HIP_LIB=$L/libc25519hip_$lib.so VERIFY_POINTS=1 VERIFY_SIZES=2048,4096,8192,16384,32768,65536,131072 timeout 300 python tools/verify_midrange.py 2>/dev/null | cut -c1-40 >> $out
done; done
cat $out
