#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
bash tools/gpu_ab.sh r05g docs/lab/ab_r05_g.cfg > /dev/null 2>&1
( C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_x25519lds.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "x25519" 2>&1 | tail -4 ) > gpurun_out/r05_c7_tests.log 2>&1
cat gpurun_out/ab_r05g.log; tail -3 gpurun_out/r05_c7_tests.log
