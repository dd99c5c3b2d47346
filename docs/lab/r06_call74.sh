#!/bin/bash
# round 6, gpurun call 74: the final tree (host transcript changed since call 69): whole GPU suite, smoke, the driver's bench command, the strict rate
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c74_tests.log 2>&1; tail -3 gpurun_out/r06_c74_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r06_bench_c74.json 2> gpurun_out/r06_bench_c74.err; tail -c 600 gpurun_out/r06_bench_c74.json
(cd tools && timeout 600 python strict_rate.py) > gpurun_out/r06_strict_rate_c74.txt 2>&1; cat gpurun_out/r06_strict_rate_c74.txt
