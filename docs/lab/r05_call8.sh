#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_debug_bounds.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r05_c8_tests.log 2>&1
bash tools/gpu_ab.sh r05h docs/lab/ab_r05_h.cfg > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/raw/kt_msm24 -o msm -- python $R/bench.py --workload msm --log2n 24 --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/raw/kt_msm24.log 2>&1
python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_msm24/msm_results.db 95 15 > $R/gpurun_out/r05_c8_msm24_timeline.txt 2>&1
cd $R
tail -4 gpurun_out/r05_c8_tests.log; cat gpurun_out/ab_r05h.log; head -40 gpurun_out/r05_c8_msm24_timeline.txt | cut -c1-120
