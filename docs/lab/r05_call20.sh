#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_verify.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r05_c20_tests.log 2>&1
export PHASES_VERIFY_SIZES=4,16,32,64,96,128
( timeout 300 python tools/small_call_phases.py | tail -15 ) > gpurun_out/r05_c20_phases.log 2>&1
export C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so
( C25519_VERIFY_HOST_MAX=128 timeout 300 python tools/small_call_phases.py | tail -15 ) > gpurun_out/r05_c20_phases_h128.log 2>&1
tail -3 gpurun_out/r05_c20_tests.log; cat gpurun_out/r05_c20_phases.log gpurun_out/r05_c20_phases_h128.log
