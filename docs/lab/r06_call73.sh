#!/bin/bash
# round 6, gpurun call 73: the host transcript with the in-block append_message and the written-out z squeeze (transcript_host.h): verify / ffi / parity modules, then the strict z-mode's rate
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_verify.py tests/test_gpu_ffi.py tests/test_gpu_parity.py tests/test_gpu_abi_c.py -x -q -m gpu > gpurun_out/r06_c73_tests.log 2>&1; tail -3 gpurun_out/r06_c73_tests.log
out=gpurun_out/r06_strict_rate.txt; : > $out
grep -m1 "model name" /proc/cpuinfo >> $out
for rep in 0 1 2; do (cd tools && timeout 300 python transcript_rate.py) >> $out 2>&1; done
(cd tools && timeout 600 python strict_rate.py) >> $out 2>&1
g++ -O3 -std=c++17 -I curve25519-dalek_amd/csrc tools/keccak_bench.cpp -o /tmp/kb && /tmp/kb >> $out 2>&1
cat $out
