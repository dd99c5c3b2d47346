#!/bin/bash
# round 6, gpurun call 27: kernel timeline of verify_batch 2^14 / 2^16 signatures with the over-long lists inside the accumulation's launch
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
cd /tmp && export TMPDIR=/tmp
for lg in 14 16; do
rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_c27_$lg -o v -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_c27_$lg.log 2>&1
echo "== 2^$lg" >> $R/gpurun_out/r06_timeline_mid_verify_b.txt
python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_c27_$lg/v_results.db 19 0 >> $R/gpurun_out/r06_timeline_mid_verify_b.txt 2>&1
done
cut -c1-110 $R/gpurun_out/r06_timeline_mid_verify_b.txt
