#!/bin/bash
# round 6, gpurun call 18: lean small direct path, k_prep_small_verify with ten-column products, column-form trees in k_mid_long -- full suite, small-call phases, verify mid range, bench
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r06_c18_tests.log 2>&1
( timeout 300 python tools/small_call_phases.py ) > gpurun_out/r06_small_call_phases.txt 2>&1
( VERIFY_SIZES=4096,8192,16384,32768,65536,131072 timeout 400 python tools/verify_midrange.py ) > gpurun_out/r06_verify_midrange_final2.txt 2>&1
( timeout 600 python tools/soak_small.py 100000 4 ) > gpurun_out/r06_soak_box4.txt 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r06_bench_default_d.json 2> gpurun_out/r06_bench_default_d.err
tail -3 gpurun_out/r06_c18_tests.log; cat gpurun_out/r06_small_call_phases.txt | head -30; cat gpurun_out/r06_verify_midrange_final2.txt; tail -13 gpurun_out/r06_soak_box4.txt; tail -3 gpurun_out/r06_bench_default_d.err; tail -c 900 gpurun_out/r06_bench_default_d.json
