#!/bin/bash
# round 6, gpurun call 41: verify_batch with its own window rule in the mid range -- parity, then 2^13 .. 2^18 against the MSM's rule (VERIFY_C=-1)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_verify.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r06_c41_tests.log 2>&1; tail -3 gpurun_out/r06_c41_tests.log
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_verify_window2.txt; : > $out
for rep in 0 1; do
for c in 0 -1; do
  for lg in 13 14 15 16 17 18; do
    line=$(env C25519_HIP_LIB=$T C25519_VERIFY_C=$c timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 30 --warmup 3 2>/dev/null | tail -1)
    python3 - $c $lg "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[3]); print("VERIFY_C=%-3s 2^%s  %.4f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done
done
cat $out
