#!/bin/bash
# round 6, gpurun call 31: the order in which the host enqueues verify_batch's two chains (VERIFY_ORDER 0 / 1 / 2 of the tuning build), 2^13 .. 2^16 signatures
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_verify_order.txt; : > $out
for rep in 0 1; do
for ord in 0 1 2; do
  for lg in 13 14 15 16; do
    line=$(env C25519_HIP_LIB=$T C25519_VERIFY_ORDER=$ord timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 30 --warmup 3 2>/dev/null | tail -1)
    python3 - $ord $lg "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[3])
print("VERIFY_ORDER=%s  2^%s  %.4f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done
done
cat $out
echo "## key BYTES (no cached points), VERIFY_ORDER 0 / 1 / 2" >> $out
for ord in 0 1 2; do C25519_HIP_LIB=$T C25519_VERIFY_ORDER=$ord VERIFY_SIZES=8192,16384,65536 timeout 200 python tools/verify_midrange.py 2>/dev/null | cut -c1-40 >> $out; done
tail -16 $out
