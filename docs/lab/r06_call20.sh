#!/bin/bash
# round 6, gpurun call 20: the MSM's mid path without the phase ring (event records between kernels are ~5 us gaps on the GPU) -- numbers, timeline, bench at mid sizes
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
( timeout 1200 python -m pytest tests/test_gpu_msm.py tests/test_gpu_multi.py tests/test_gpu_ffi.py -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/r06_c20_tests.log 2>&1
( MIDRANGE_SIZES=12288,16384,32768,65536,131072 timeout 300 python tools/midrange_numbers.py ) > gpurun_out/r06_midrange_noring.txt 2>&1
( timeout 200 python bench.py --no-cpu-baseline --no-sub --workload msm --log2n 14 --steps 200 --warmup 20 | tail -c 700 ) > gpurun_out/r06_bench_2p14.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for lg in 14 16; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_mid_$lg -o mid_$lg -- python $R/bench.py --no-cpu-baseline --no-sub --workload msm --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_mid_$lg.log 2>&1
  python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_mid_$lg/mid_${lg}_results.db k_mid_front 1 > $R/gpurun_out/r06_timeline_mid_msm_2p$lg.txt 2>&1
done
cd $R; rm -rf gpurun_out/raw/*/*.db
tail -3 gpurun_out/r06_c20_tests.log; cat gpurun_out/r06_midrange_noring.txt; cat gpurun_out/r06_bench_2p14.txt; echo; for lg in 14 16; do echo "== 2^$lg"; cut -c1-110 gpurun_out/r06_timeline_mid_msm_2p$lg.txt | head -12; done
