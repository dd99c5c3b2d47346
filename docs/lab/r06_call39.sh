#!/bin/bash
# round 6, gpurun call 39: the Horner fold in lane form on AVX-512 IFMA (doublings AND additions): small-call phases, parity of the small modules
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
grep -m1 "model name" /proc/cpuinfo; grep -c avx512ifma /proc/cpuinfo
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_abi_c.py tests/test_gpu_extra.py -x -q -m gpu > gpurun_out/r06_c39_tests.log 2>&1; tail -3 gpurun_out/r06_c39_tests.log
PHASES_VERIFY_SIZES=4,8,16,64,128 timeout 600 python tools/small_call_phases.py > gpurun_out/r06_small_call_phases_fold.txt 2>&1; cat gpurun_out/r06_small_call_phases_fold.txt
