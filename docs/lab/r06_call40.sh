#!/bin/bash
# round 6, gpurun call 40: the window width of mid-size device-z-mode verify_batch (VERIFY_C of the tuning build; 0 = the MSM's rule for 2n + 1 terms)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_verify_window.txt; : > $out
for rep in 0 1; do
for c in 0 12 13 14 15 16; do
  for lg in 13 14 15 16; do
    line=$(env C25519_HIP_LIB=$T C25519_VERIFY_C=$c timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 30 --warmup 3 2>/dev/null | tail -1)
    python3 - $c $lg "$line" >> $out <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[3]); print("VERIFY_C=%-3s 2^%s  %.4f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
except Exception as e:
    print("VERIFY_C=%-3s 2^%s  FAILED" % (sys.argv[1], sys.argv[2]))
PY
  done
done
done
cat $out
