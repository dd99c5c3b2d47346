#!/bin/bash
# round 6, gpurun call 11: the mid path's own long-list cap (max(48, 3 x mean)) -- verify / MSM modules, verify_batch across the mid range, timelines at 2^15 / 2^16 signatures
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/gpurun_out/raw; cd $R
( timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_verify.py tests/test_gpu_extra.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r06_c11_tests.log 2>&1
( VERIFY_SIZES=6144,8192,16384,32768,65536,131072 timeout 400 python tools/verify_midrange.py ) > gpurun_out/r06_verify_midrange_cap.txt 2>&1
( MIDRANGE_SIZES=12288,16384,32768,65536,131072 timeout 300 python tools/midrange_numbers.py ) > gpurun_out/r06_midrange_cap.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for lg in 15 16; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_midv_$lg -o midv_$lg -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_midv_$lg.log 2>&1
  python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_midv_$lg/midv_${lg}_results.db k_slot_init 1 > $R/gpurun_out/r06_timeline_mid_verify_2p$lg.txt 2>&1
done
cd $R; rm -rf gpurun_out/raw/*/*.db
tail -3 gpurun_out/r06_c11_tests.log; cat gpurun_out/r06_verify_midrange_cap.txt gpurun_out/r06_midrange_cap.txt; for lg in 15 16; do echo "== 2^$lg"; cut -c1-110 gpurun_out/r06_timeline_mid_verify_2p$lg.txt | sed -n 13,30p; done
