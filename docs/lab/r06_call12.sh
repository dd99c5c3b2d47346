#!/bin/bash
# round 6, gpurun call 12: the host fold's doubling chain on AVX-512 IFMA -- full suite, small-call phases, small / mid numbers, the driver's bench command
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r06_c12_tests.log 2>&1
( timeout 300 python tools/small_call_phases.py ) > gpurun_out/r06_small_call_phases.txt 2>&1
( MIDRANGE_SIZES=1,16,256,1024,4096,12288,16384,65536,131072,1048576 timeout 300 python tools/midrange_numbers.py ) > gpurun_out/r06_midrange_ifma.txt 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r06_bench_default_c.json 2> gpurun_out/r06_bench_default_c.err
tail -3 gpurun_out/r06_c12_tests.log; cat gpurun_out/r06_small_call_phases.txt | tail -40; cat gpurun_out/r06_midrange_ifma.txt; tail -3 gpurun_out/r06_bench_default_c.err; tail -c 1500 gpurun_out/r06_bench_default_c.json
