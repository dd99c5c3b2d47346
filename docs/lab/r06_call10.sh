#!/bin/bash
# round 6, gpurun call 10: why is verify_batch of 2^15 signatures slower than 2^16?  timelines of both; host CPU flags of the box
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/gpurun_out/raw
grep -m1 "model name" /proc/cpuinfo > $R/gpurun_out/r06_host_cpu.txt; grep -m1 flags /proc/cpuinfo | tr ' ' '\n' | grep -E "avx512|adx|bmi2|sha_ni|vaes" | tr '\n' ' ' >> $R/gpurun_out/r06_host_cpu.txt; nproc >> $R/gpurun_out/r06_host_cpu.txt
cd /tmp && export TMPDIR=/tmp
for lg in 15 16; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_midv_$lg -o midv_$lg -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_midv_$lg.log 2>&1
  python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_midv_$lg/midv_${lg}_results.db k_slot_init 1 > $R/gpurun_out/r06_timeline_mid_verify_2p$lg.txt 2>&1
done
cd $R; rm -rf gpurun_out/raw/*/*.db
cat gpurun_out/r06_host_cpu.txt; for lg in 15 16; do echo "== 2^$lg"; cut -c1-110 gpurun_out/r06_timeline_mid_verify_2p$lg.txt | head -30; done
