#!/bin/bash
# round 6, gpurun call 6: mid path with the global bucket order and lane-masked LDS atomics; A/B arms of its knobs (tuning build)
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/gpurun_out/raw
cd $R
( timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_raw160.py tests/test_gpu_extra.py tests/test_gpu_verify.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r06_c6_tests.log 2>&1
export MIDRANGE_SIZES=12288,16384,32768,65536,131072,262144
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_mid_knobs.txt; : > $out
for rep in 0 1; do
echo "## release defaults (projective records up to 2^15 terms, 128 sort blocks, mid path up to 2^17), rep $rep" >> $out; timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## round-5 pipeline (MSM_MID_MAX=0), rep $rep" >> $out; C25519_HIP_LIB=$T C25519_MSM_MID_MAX=0 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## mid path up to 2^18, projective records up to 2^17, rep $rep" >> $out; C25519_HIP_LIB=$T C25519_MSM_MID_MAX=262144 C25519_MID_PROJ_MAX=131072 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## mid path up to 2^18, normaliser from 12288 terms (MID_PROJ_MAX=0), rep $rep" >> $out; C25519_HIP_LIB=$T C25519_MSM_MID_MAX=262144 C25519_MID_PROJ_MAX=0 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## mid path up to 2^18, 256 sort blocks, rep $rep" >> $out; C25519_HIP_LIB=$T C25519_MSM_MID_MAX=262144 C25519_MID_SORT_BLOCKS=256 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
( VERIFY_SIZES=8192,16384,65536,131072 timeout 300 python tools/verify_midrange.py ) > gpurun_out/r06_verify_midrange_mid4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for lg in 14 16 17; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_mid_$lg -o mid_$lg -- python $R/bench.py --no-cpu-baseline --no-sub --workload msm --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_mid_$lg.log 2>&1
  python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_mid_$lg/mid_${lg}_results.db k_mid_front 1 > $R/gpurun_out/r06_timeline_mid_msm_2p$lg.txt 2>&1
done
for lg in 14 16; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_midv_$lg -o midv_$lg -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_midv_$lg.log 2>&1
  python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_midv_$lg/midv_${lg}_results.db k_slot_init 1 > $R/gpurun_out/r06_timeline_mid_verify_2p$lg.txt 2>&1
done
cd $R
rm -rf gpurun_out/raw/*/*.db
tail -5 gpurun_out/r06_c6_tests.log; cat $out gpurun_out/r06_verify_midrange_mid4.txt
for f in gpurun_out/r06_timeline_mid_msm_2p14.txt gpurun_out/r06_timeline_mid_msm_2p16.txt gpurun_out/r06_timeline_mid_msm_2p17.txt gpurun_out/r06_timeline_mid_verify_2p14.txt gpurun_out/r06_timeline_mid_verify_2p16.txt; do echo "== $f"; head -32 $f | cut -c1-110; done
