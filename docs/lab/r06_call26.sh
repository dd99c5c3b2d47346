#!/bin/bash
# round 6, gpurun call 26: verify_batch of 2^13 .. 2^17 signatures -- the over-long lists in the accumulation's own launch (MID_LONG_BESIDE=1: the pair of launches on two
# streams as before) and the whole hash chain enqueued ahead of the decompression (CHAIN_FIRST=0: only k_hram ahead, as before); parity tests of the two modules first
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_msm.py -x -q -m gpu > gpurun_out/r06_c26_tests.log 2>&1; tail -3 gpurun_out/r06_c26_tests.log
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_verify_mid_launches.txt; : > $out
for rep in 0 1; do
for arm in "" "C25519_MID_LONG_BESIDE=1" "C25519_CHAIN_FIRST=0" "C25519_MID_LONG_BESIDE=1 C25519_CHAIN_FIRST=0"; do
  for lg in 13 14 15 16 17; do
    line=$(env C25519_HIP_LIB=$T $arm timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 30 --warmup 3 2>/dev/null | tail -1)
    python3 - "$arm" $lg "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[3])
print("%-52s 2^%s  %.4f ms" % (sys.argv[1] or "(default)", sys.argv[2], d["ms_per_step"]))
PY
  done
done
done
cat $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_c26 -o v14 -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n 14 --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_c26.log 2>&1
python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_c26/v14_results.db 19 0 > $R/gpurun_out/r06_timeline_mid_verify_2p14_b.txt 2>&1
cat $R/gpurun_out/r06_timeline_mid_verify_2p14_b.txt | cut -c1-110
