#!/bin/bash
# round 6, gpurun call 50: the small path hands over to the mid path at 6144 terms (MSM) / 2048 signatures (verify_batch) -- parity (MSM, verify, soak-style mixed), then the
# release library's numbers at the sizes around the old and new boundaries, host pointers from C included (tools/ffi_numbers.py)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_verify.py tests/test_gpu_extra.py -x -q -m gpu > gpurun_out/r06_c50_tests.log 2>&1; tail -4 gpurun_out/r06_c50_tests.log
out=gpurun_out/r06_small_mid_boundary_release.txt; : > $out
echo "## release library, MSM (raw points)" >> $out
MIDRANGE_SIZES=3072,4096,6143,6144,7000,8192,10240,12287,12288,16384 timeout 300 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## release library, verify_batch" >> $out
VERIFY_SIZES=1024,2047,2048,3072,4096,5120,6143,6144,8192 timeout 300 python tools/verify_midrange.py 2>/dev/null >> $out
cat $out
