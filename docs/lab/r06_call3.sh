#!/bin/bash
# round 6, gpurun call 3: kernel timelines of the mid path at 2^14 / 2^16 / 2^18 terms and verify_batch 2^14 / 2^16
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/gpurun_out/raw
cd /tmp && export TMPDIR=/tmp
for lg in 14 16 18; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_mid_$lg -o mid_$lg -- python $R/bench.py --no-cpu-baseline --no-sub --workload msm --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_mid_$lg.log 2>&1
  python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_mid_$lg/mid_${lg}_results.db k_mid_front 2 > $R/gpurun_out/r06_timeline_mid_msm_2p$lg.txt 2>&1
done
for lg in 14 16; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_midv_$lg -o midv_$lg -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_midv_$lg.log 2>&1
  python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_midv_$lg/midv_${lg}_results.db k_slot_init 2 > $R/gpurun_out/r06_timeline_mid_verify_2p$lg.txt 2>&1
done
cd $R
for f in gpurun_out/r06_timeline_mid_*; do echo "== $f"; head -40 $f | cut -c1-110; done
