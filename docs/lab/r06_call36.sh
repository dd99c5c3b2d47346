#!/bin/bash
# round 6, gpurun call 36: the cap on a bucket lane's list chosen from the Poisson tail (MID_LONG_TARGET expected long lists; 0 = max(48, 3 x mean) as before) with the
# over-long lists inside the accumulation's launch (MID_RAW_FUSED); parity first
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_verify.py -x -q -m gpu > gpurun_out/r06_c36_tests.log 2>&1; tail -5 gpurun_out/r06_c36_tests.log
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_mid_cap.txt; : > $out
for rep in 0 1; do
for arm in "C25519_MID_LONG_TARGET=0 C25519_MID_RAW_FUSED=0" "C25519_MID_LONG_TARGET=0" "C25519_MID_LONG_TARGET=64" "C25519_MID_LONG_TARGET=256" "C25519_MID_LONG_TARGET=1024" "C25519_MID_LONG_TARGET=256 C25519_MID_RAW_FUSED=0"; do
echo "## $arm, rep $rep" >> $out
env C25519_HIP_LIB=$T $arm MIDRANGE_SIZES=12288,16384,32768,65536,131072,262144 timeout 200 python tools/midrange_numbers.py 2>/dev/null | cut -c1-48 >> $out
done
done
for arm in "C25519_MID_LONG_TARGET=0" "C25519_MID_LONG_TARGET=64" "C25519_MID_LONG_TARGET=256" "C25519_MID_LONG_TARGET=1024"; do
  for lg in 13 14 15 16 17; do
    line=$(env C25519_HIP_LIB=$T $arm timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 30 --warmup 3 2>/dev/null | tail -1)
    python3 - "$arm" $lg "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[3])
print("verify_batch %-40s 2^%s  %.4f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"]))
PY
  done
done
cat $out
