#!/bin/bash
# round 6, gpurun call 35: k_mid_acc_coop (the wave-cooperative gather in the mid path's accumulation of projective records): parity, then MID_COOP_MIN 0 (never) / 2^14 / 2^16 (the default)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 1500 python -m pytest tests/test_gpu_msm.py -x -q -m gpu > gpurun_out/r06_c35_tests.log 2>&1; tail -5 gpurun_out/r06_c35_tests.log
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_mid_coop.txt; : > $out
for rep in 0 1; do
for m in 0 16384 65536; do
echo "## MID_COOP_MIN=$m, rep $rep" >> $out
C25519_HIP_LIB=$T C25519_MID_COOP_MIN=$m MIDRANGE_SIZES=16384,32768,65536,131072,200000,262144 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
done
cat $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_c35 -o v -- python $R/bench.py --no-cpu-baseline --no-sub --workload msm --log2n 18 --steps 10 --warmup 3 > $R/gpurun_out/raw/kt_c35.log 2>&1
python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_c35/v_results.db 8 0 | cut -c1-110
