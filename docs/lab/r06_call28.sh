#!/bin/bash
# round 6, gpurun call 28: verify_batch of 2^13 .. 2^17 signatures with the MSM on the hash chain's stream and the record published by the last reduction block
# (MID_ON_CHAIN=0: the MSM on the main stream behind two hand-overs, as before; VERIFY_DIRECT=0: copy + synchronise); parity tests of the two modules first
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 1500 python -m pytest tests/test_gpu_verify.py tests/test_gpu_msm.py -x -q -m gpu > gpurun_out/r06_c28_tests.log 2>&1; tail -5 gpurun_out/r06_c28_tests.log
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_verify_on_chain.txt; : > $out
for rep in 0 1; do
for arm in "" "C25519_VERIFY_DIRECT=0" "C25519_MID_ON_CHAIN=0" "C25519_MID_ON_CHAIN=0 C25519_CHAIN_FIRST=0 C25519_MID_LONG_BESIDE=1"; do
  for lg in 13 14 15 16 17; do
    line=$(env C25519_HIP_LIB=$T $arm timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 30 --warmup 3 2>/dev/null | tail -1)
    python3 - "$arm" $lg "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[3])
print("%-76s 2^%s  %.4f ms" % (sys.argv[1] or "(default)", sys.argv[2], d["ms_per_step"]))
PY
  done
done
done
cat $out
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r06_timeline_mid_verify_c.txt
for lg in 14 16; do
rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_c28_$lg -o v -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_c28_$lg.log 2>&1
echo "== 2^$lg" >> $R/gpurun_out/r06_timeline_mid_verify_c.txt
python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_c28_$lg/v_results.db 18 0 >> $R/gpurun_out/r06_timeline_mid_verify_c.txt 2>&1
done
cut -c1-110 $R/gpurun_out/r06_timeline_mid_verify_c.txt
