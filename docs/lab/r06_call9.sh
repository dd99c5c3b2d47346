#!/bin/bash
# round 6, gpurun call 9 (box 3 of the soak): full GPU suite, soak, instruction-rate probes with the many-to-one permute patterns, mid-range numbers, the driver's bench command
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06_c9_tests.log 2>&1
( timeout 900 python tools/soak_small.py 200000 3 ) > gpurun_out/r06_soak_box3.txt 2>&1
( timeout 600 python tools/probes.py ) > gpurun_out/r06_instruction_rates.txt 2>&1
( MIDRANGE_SIZES=4096,8192,12288,16384,32768,65536,131072,262144,524288,1048576 timeout 300 python tools/midrange_numbers.py ) > gpurun_out/r06_midrange_final.txt 2>&1
( VERIFY_SIZES=4096,8192,16384,32768,65536,131072,262144 timeout 400 python tools/verify_midrange.py ) > gpurun_out/r06_verify_midrange_final.txt 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r06_bench_default_b.json 2> gpurun_out/r06_bench_default_b.err
tail -4 gpurun_out/r06_c9_tests.log; tail -14 gpurun_out/r06_soak_box3.txt; tail -22 gpurun_out/r06_instruction_rates.txt; cat gpurun_out/r06_midrange_final.txt gpurun_out/r06_verify_midrange_final.txt; tail -3 gpurun_out/r06_bench_default_b.err; tail -c 1200 gpurun_out/r06_bench_default_b.json
