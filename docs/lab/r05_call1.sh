#!/bin/bash
# round 5, GPU call 1: parity of the cross-lane-fetch constant-time fixed base, the probes behind it, first A/B arms
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_single.py tests/test_gpu_extra.py tests/test_gpu_field.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r05_c1_tests.log 2>&1
( timeout 600 python tools/probes.py ) > gpurun_out/r05_c1_probes.txt 2>&1
bash tools/gpu_ab.sh r05a docs/lab/ab_r05_a.cfg > /dev/null 2>&1
tail -5 gpurun_out/r05_c1_tests.log; tail -14 gpurun_out/r05_c1_probes.txt; cat gpurun_out/ab_r05a.log
