#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -k "variants or every_size or identity" 2>&1 | tail -15 ) > gpurun_out/r05_c2_tests.log 2>&1
( timeout 600 python tools/probes.py ) > gpurun_out/r05_c2_probes.txt 2>&1
bash tools/gpu_ab.sh r05b docs/lab/ab_r05_b.cfg > /dev/null 2>&1
tail -5 gpurun_out/r05_c2_tests.log; tail -12 gpurun_out/r05_c2_probes.txt; cat gpurun_out/ab_r05b.log
