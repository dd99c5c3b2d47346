#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for i in $(seq 1 14); do
  timeout 500 python -m pytest tests/test_gpu_verify.py -m gpu -x -q > gpurun_out/r05_c27_tests.log 2>&1
  rc=$?
  tail -1 gpurun_out/r05_c27_tests.log
  if [ $rc -ne 0 ]; then echo "iteration $i rc=$rc"; tail -80 gpurun_out/r05_c27_tests.log; cp gpurun_out/r05_c27_tests.log gpurun_out/r05_c27_fail.log; break; fi
done
echo "done $i"
