#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_shim_mock.py tests/test_gpu_abi_c.py tests/test_gpu_ffi.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05_c15_tests.log 2>&1
( timeout 300 python tools/small_call_phases.py ) > gpurun_out/r05_c15_phases.log 2>&1
tail -15 gpurun_out/r05_c15_tests.log; tail -16 gpurun_out/r05_c15_phases.log
