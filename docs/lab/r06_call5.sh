#!/bin/bash
# round 6, gpurun call 4: mid path, second version (one product per stage in the reduction, large sort slices, projective records up to 2^15 terms / normaliser + k_accumulate above)
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/gpurun_out/raw
cd $R
( timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_raw160.py tests/test_gpu_extra.py tests/test_gpu_verify.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r06_c5_tests.log 2>&1
( MIDRANGE_SIZES=8192,12288,16384,32768,65536,131072,262144,524288,2097152 timeout 300 python tools/midrange_numbers.py ) > gpurun_out/r06_midrange_mid3.txt 2>&1
( VERIFY_SIZES=8192,16384,65536,131072 timeout 300 python tools/verify_midrange.py ) > gpurun_out/r06_verify_midrange_mid3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for lg in 14 16 18 21; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_mid_$lg -o mid_$lg -- python $R/bench.py --no-cpu-baseline --no-sub --workload msm --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_mid_$lg.log 2>&1
  python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_mid_$lg/mid_${lg}_results.db k_mid_front 1 > $R/gpurun_out/r06_timeline_mid_msm_2p$lg.txt 2>&1
done
python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_mid_21/mid_21_results.db k_prep_raw2 1 > $R/gpurun_out/r06_timeline_msm_2p21.txt 2>&1
for lg in 14 16; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_midv_$lg -o midv_$lg -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_midv_$lg.log 2>&1
  python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_midv_$lg/midv_${lg}_results.db k_slot_init 1 > $R/gpurun_out/r06_timeline_mid_verify_2p$lg.txt 2>&1
done
cd $R
rm -rf gpurun_out/raw/*/*.db
tail -5 gpurun_out/r06_c5_tests.log; cat gpurun_out/r06_midrange_mid3.txt gpurun_out/r06_verify_midrange_mid3.txt
for f in gpurun_out/r06_timeline_mid_* gpurun_out/r06_timeline_msm_2p21.txt; do echo "== $f"; head -32 $f | cut -c1-110; done
