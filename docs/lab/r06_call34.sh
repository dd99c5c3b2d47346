#!/bin/bash
# round 6, gpurun call 34: counters of the mid path's kernels at 2^18 and 2^16 terms (why does k_mid_acc<0> run at 0.38 of the multiplier peak where k_accumulate reaches 0.5?)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
bash tools/pmc_quick.sh mid18 --workload msm --log2n 18
bash tools/pmc_quick.sh mid16 --workload msm --log2n 16
grep -A14 "k_mid_acc" gpurun_out/pq_mid18.txt | head -40
grep -A14 "k_mid_acc" gpurun_out/pq_mid16.txt | head -20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_c34 -o v -- python $R/bench.py --no-cpu-baseline --no-sub --workload msm --log2n 18 --steps 10 --warmup 3 > $R/gpurun_out/raw/kt_c34.log 2>&1
python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_c34/v_results.db 8 0 | cut -c1-110
