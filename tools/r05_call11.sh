#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_gpu_msm.py tests/test_gpu_verify.py tests/test_gpu_extra.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r05_c11_tests.log 2>&1
bash tools/gpu_ab.sh r05i tools/ab_r05_i.cfg > /dev/null 2>&1
tail -4 gpurun_out/r05_c11_tests.log; cat gpurun_out/ab_r05i.log | cut -c1-150
