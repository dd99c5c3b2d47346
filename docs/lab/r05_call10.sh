#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_ffi.py tests/test_gpu_abi_c.py tests/test_gpu_shim_mock.py tests/test_gpu_verify.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r05_c10_tests.log 2>&1
( timeout 300 python tools/small_call_phases.py ) > gpurun_out/r05_c10_phases_zc.log 2>&1
export C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so
( C25519_ZERO_COPY_MAX=0 timeout 300 python tools/small_call_phases.py ) > gpurun_out/r05_c10_phases_copy.log 2>&1
( timeout 300 python tools/small_call_phases.py ) > gpurun_out/r05_c10_phases_zc_tune.log 2>&1
tail -4 gpurun_out/r05_c10_tests.log; cat gpurun_out/r05_c10_phases_zc.log; cat gpurun_out/r05_c10_phases_copy.log
