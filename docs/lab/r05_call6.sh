#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_single.py tests/test_gpu_extra.py tests/test_gpu_ffi.py tests/test_gpu_verify.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r05_c6_tests.log 2>&1
bash tools/gpu_ab.sh r05f docs/lab/ab_r05_f.cfg > /dev/null 2>&1
tail -5 gpurun_out/r05_c6_tests.log; cat gpurun_out/ab_r05f.log
