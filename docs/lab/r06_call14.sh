#!/bin/bash
# round 6, gpurun call 14: k_mid_acc with a prefetched record for calls whose buckets fit the machine at once -- A/B of the threshold (tuning build), MSM tests
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
( timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_raw160.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r06_c14_tests.log 2>&1
export MIDRANGE_SIZES=12288,16384,24000,32768,49152,65536,131072
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_mid_prefetch.txt; : > $out
for rep in 0 1; do
for w in 0 2048 3072 6144; do
echo "## MID_PREFETCH_WAVES=$w (a record prefetched across the addition when the buckets make at most $w waves), rep $rep" >> $out; C25519_HIP_LIB=$T C25519_MID_PREFETCH_WAVES=$w timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
done
cd /tmp && export TMPDIR=/tmp
for lg in 14 15; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_mid_$lg -o mid_$lg -- python $R/bench.py --no-cpu-baseline --no-sub --workload msm --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_mid_$lg.log 2>&1
  python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_mid_$lg/mid_${lg}_results.db k_mid_front 1 > $R/gpurun_out/r06_timeline_mid_msm_2p$lg.txt 2>&1
done
cd $R; rm -rf gpurun_out/raw/*/*.db
tail -3 gpurun_out/r06_c14_tests.log; cat $out; for lg in 14 15; do echo "== 2^$lg"; cut -c1-110 gpurun_out/r06_timeline_mid_msm_2p$lg.txt | head -12; done
