#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_single.py tests/test_gpu_msm.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05_c3_tests.log 2>&1
bash tools/gpu_ab.sh r05c docs/lab/ab_r05_c.cfg > /dev/null 2>&1
tail -5 gpurun_out/r05_c3_tests.log; cat gpurun_out/ab_r05c.log
