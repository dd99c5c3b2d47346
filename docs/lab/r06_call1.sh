#!/bin/bash
# round 6, gpurun call 1 (box 1): the full GPU suite on the hardened small path, the small-call soak, the synchronisation-quantum table, baselines of the mid range
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06_c1_tests.log 2>&1
( timeout 900 python tools/soak_small.py 200000 1 ) > gpurun_out/r06_soak_box1.txt 2>&1
( timeout 300 python tools/sync_quantum.py ) > gpurun_out/r06_sync_quantum.txt 2>&1
( MIDRANGE_SIZES=8192,12288,16384,32768,65536,131072,262144,524288 timeout 300 python tools/midrange_numbers.py ) > gpurun_out/r06_midrange_before.txt 2>&1
( VERIFY_SIZES=8192,16384,65536,262144 timeout 300 python tools/verify_midrange.py ) > gpurun_out/r06_verify_midrange_before.txt 2>&1
( timeout 300 python tools/precomp_numbers.py ) > gpurun_out/r06_precomp.txt 2>&1
tail -5 gpurun_out/r06_c1_tests.log; tail -12 gpurun_out/r06_soak_box1.txt; cat gpurun_out/r06_sync_quantum.txt; cat gpurun_out/r06_midrange_before.txt gpurun_out/r06_verify_midrange_before.txt gpurun_out/r06_precomp.txt
