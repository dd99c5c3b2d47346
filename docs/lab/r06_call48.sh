#!/bin/bash
# round 6, gpurun call 48: the mid path's sort with a window's row split over R blocks per slice (k_mid_count / k_mid_offsets / k_mid_place; MID_SORT_SPLIT=1: the single kernel) -- parity, then sizes
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 1800 python -m pytest tests/test_gpu_msm.py tests/test_gpu_verify.py -x -q -m gpu > gpurun_out/r06_c48_tests.log 2>&1; tail -4 gpurun_out/r06_c48_tests.log
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_mid_sort_split.txt; : > $out
for rep in 0 1; do
for sp in 1 0 2 4 8; do
  echo "## MID_SORT_SPLIT=$sp rep $rep (0 = the rule: 2 parts from 2^15 terms, 4 from 2^17; 1 = the single kernel)" >> $out
  C25519_HIP_LIB=$T C25519_MID_SORT_SPLIT=$sp MIDRANGE_SIZES=16384,32768,65536,131072,262144 timeout 200 python tools/midrange_numbers.py 2>/dev/null | cut -c1-48 >> $out
  for lg in 14 15 16 17; do
    line=$(env C25519_HIP_LIB=$T C25519_MID_SORT_SPLIT=$sp timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 30 --warmup 3 2>/dev/null | tail -1)
    python3 - $sp $lg "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[3]); print("verify_batch MID_SORT_SPLIT=%s 2^%s  %.4f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done
done
cat $out
