#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05_c5_tests.log 2>&1
( C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so C25519_PUBLISH=0 timeout 300 python tools/small_call_phases.py ) > gpurun_out/r05_c5_phases_before.txt 2>&1
( timeout 300 python tools/small_call_phases.py ) > gpurun_out/r05_c5_phases_after.txt 2>&1
bash tools/gpu_ab.sh r05e docs/lab/ab_r05_e.cfg > /dev/null 2>&1
tail -6 gpurun_out/r05_c5_tests.log; cat gpurun_out/r05_c5_phases_before.txt gpurun_out/r05_c5_phases_after.txt | grep -v "^columns\|^page\|^from"; cat gpurun_out/ab_r05e.log
