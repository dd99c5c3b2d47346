#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_verify.py tests/test_gpu_ffi.py tests/test_gpu_abi_c.py tests/test_gpu_shim_mock.py tests/test_gpu_single.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05_c11_tests.log 2>&1
( timeout 300 python tools/small_call_phases.py ) > gpurun_out/r05_c11_phases.log 2>&1
export C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so
( C25519_VERIFY_HOST_MAX=128 timeout 300 python tools/small_call_phases.py ) > gpurun_out/r05_c11_phases_h128.log 2>&1
tail -15 gpurun_out/r05_c11_tests.log; tail -16 gpurun_out/r05_c11_phases.log; tail -16 gpurun_out/r05_c11_phases_h128.log
