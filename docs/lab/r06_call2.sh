#!/bin/bash
# round 6, gpurun call 2: first run of the mid path (mid.hip) -- parity tests of the MSM / verify / extra modules, then the mid-range numbers
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_raw160.py tests/test_gpu_extra.py tests/test_gpu_verify.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r06_c2_tests.log 2>&1
( MIDRANGE_SIZES=8192,12288,16384,32768,65536,131072,262144,524288 timeout 300 python tools/midrange_numbers.py ) > gpurun_out/r06_midrange_mid1.txt 2>&1
( VERIFY_SIZES=8192,16384,65536,131072 timeout 300 python tools/verify_midrange.py ) > gpurun_out/r06_verify_midrange_mid1.txt 2>&1
tail -25 gpurun_out/r06_c2_tests.log; cat gpurun_out/r06_midrange_mid1.txt gpurun_out/r06_verify_midrange_mid1.txt
