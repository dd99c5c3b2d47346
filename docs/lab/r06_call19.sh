#!/bin/bash
# round 6, gpurun call 19: 5 .. 16 terms in ONE 1024-thread block (k_small_cols<4>, no k_small_reduce launch): parity of every small size, phases, A/B against one group per block
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_verify.py tests/test_gpu_debug_bounds.py -m gpu -x -q -k "every_size or small or sum_of_squares or host_hashing or lost_publication or bound" 2>&1 | tail -4 ) > gpurun_out/r06_c19_tests.log 2>&1
( PHASES_VERIFY_SIZES=1,2,4,7,8,16,64 timeout 300 python tools/small_call_phases.py ) > gpurun_out/r06_small_call_phases_wide.txt 2>&1
( C25519_HIP_LIB=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so C25519_SMALL_WIDE=0 PHASES_VERIFY_SIZES=1,2,4,7,8,16,64 timeout 300 python tools/small_call_phases.py ) > gpurun_out/r06_small_call_phases_narrow.txt 2>&1
tail -3 gpurun_out/r06_c19_tests.log; echo "== wide (default)"; head -30 gpurun_out/r06_small_call_phases_wide.txt | grep -v "^columns\|^page\|^from"; echo "== SMALL_WIDE=0"; head -30 gpurun_out/r06_small_call_phases_narrow.txt | grep -v "^columns\|^page\|^from"
