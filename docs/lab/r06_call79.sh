#!/bin/bash
# round 6, gpurun call 79 (and, with the register-resident z squeeze, call 80): the final tree with the lane-pair Keccak-f in the host transcript: whole GPU suite, smoke, strict rate, the driver's bench command
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c80_tests.log 2>&1; tail -3 gpurun_out/r06_c80_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
out=gpurun_out/r06_strict_rate_c80.txt; : > $out
for rep in 0 1 2; do (cd tools && timeout 300 python transcript_rate.py) >> $out 2>&1; done
(cd tools && timeout 600 python strict_rate.py) >> $out 2>&1; grep -v amdgpu.ids $out
timeout 900 python bench.py > gpurun_out/r06_bench_c80.json 2> gpurun_out/r06_bench_c80.err; tail -c 300 gpurun_out/r06_bench_c80.json
