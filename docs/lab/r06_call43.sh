#!/bin/bash
# round 6, gpurun call 43: slices per window in k_mid_sort (MID_SORT_BLOCKS = the number of blocks wanted) with the 16-bit windows verify_batch now uses from 49 152 signatures
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_mid_sort_blocks2.txt; : > $out
for rep in 0 1; do
for b in 128 64 32; do
  for lg in 14 15 16 17; do
    line=$(env C25519_HIP_LIB=$T C25519_MID_SORT_BLOCKS=$b timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 30 --warmup 3 2>/dev/null | tail -1)
    python3 - $b $lg "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[3]); print("verify_batch MID_SORT_BLOCKS=%-4s 2^%s  %.4f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
  echo "## msm MID_SORT_BLOCKS=$b rep $rep" >> $out
  C25519_HIP_LIB=$T C25519_MID_SORT_BLOCKS=$b MIDRANGE_SIZES=16384,65536,131072,262144 timeout 200 python tools/midrange_numbers.py 2>/dev/null | cut -c1-48 >> $out
done
done
cat $out
