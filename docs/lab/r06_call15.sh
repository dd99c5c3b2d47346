#!/bin/bash
# round 6, gpurun call 15: verify_batch in two halves -- parity (verify / multi / ffi modules), A/B against one pass, timeline at 2^20
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
( timeout 1500 python -m pytest tests/test_gpu_verify.py tests/test_gpu_multi.py tests/test_gpu_ffi.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06_c15_tests.log 2>&1
bash tools/gpu_ab.sh r06_verify_split docs/lab/ab_r06_verify_split.cfg > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_v20 -o v20 -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n 20 --steps 10 --warmup 3 > $R/gpurun_out/raw/kt_v20.log 2>&1
python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_v20/v20_results.db k_slot_init 2 > $R/gpurun_out/r06_timeline_verify_2p20_split.txt 2>&1
cd $R; rm -rf gpurun_out/raw/*/*.db
tail -12 gpurun_out/r06_c15_tests.log; cat gpurun_out/ab_r06_verify_split.log; head -60 gpurun_out/r06_timeline_verify_2p20_split.txt | cut -c1-110; tail -5 gpurun_out/ab_err.log
