#!/bin/bash
# round 6, gpurun call 33: the one-pass normaliser of VerifyingKey points (PREP_AFFINE_FIRST=0 of the tuning build: the general normaliser as before) -- parity, then 2^14 .. 2^20
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 1500 python -m pytest tests/test_gpu_verify.py -x -q -m gpu > gpurun_out/r06_c33_tests.log 2>&1; tail -5 gpurun_out/r06_c33_tests.log
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_prep_affine.txt; : > $out
for rep in 0 1; do
for arm in "" "C25519_PREP_AFFINE_FIRST=0"; do
  for lg in 14 16 18 19 20; do
    line=$(env C25519_HIP_LIB=$T $arm timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 30 --warmup 3 2>/dev/null | tail -1)
    python3 - "$arm" $lg "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[3])
print("%-30s 2^%s  %.4f ms" % (sys.argv[1] or "(default)", sys.argv[2], d["ms_per_step"]))
PY
  done
done
done
cat $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_c33_20 -o v -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n 20 --steps 5 --warmup 2 > $R/gpurun_out/raw/kt_c33_20.log 2>&1
python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_c33_20/v_results.db 25 0 > $R/gpurun_out/r06_timeline_verify_2p20_affine.txt 2>&1
cut -c1-110 $R/gpurun_out/r06_timeline_verify_2p20_affine.txt
