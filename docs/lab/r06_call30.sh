#!/bin/bash
# round 6, gpurun call 30: the device z-mode's tree on BLAKE2b (v5) -- parity (kernels = host restatement = hashlib restatement), verify_batch 2^13 .. 2^20, timelines
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 1500 python -m pytest tests/test_gpu_verify.py tests/test_gpu_multi.py tests/test_gpu_ffi.py -x -q -m gpu > gpurun_out/r06_c30_tests.log 2>&1; tail -5 gpurun_out/r06_c30_tests.log
out=gpurun_out/r06_verify_blake2b.txt; : > $out
for rep in 0 1; do
  for lg in 13 14 15 16 17 18 20; do
    line=$(timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 30 --warmup 3 2>/dev/null | tail -1)
    python3 - $lg "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print("verify_batch 2^%s  %.4f ms" % (sys.argv[1], d["ms_per_step"]))
PY
  done
done
cat $out
timeout 300 python tools/verify_call_phases.py > gpurun_out/r06_verify_call_phases_b.txt 2>&1; cat gpurun_out/r06_verify_call_phases_b.txt
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r06_timeline_mid_verify_d.txt
for lg in 14 16; do
rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_c30_$lg -o v -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n $lg --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_c30_$lg.log 2>&1
echo "== 2^$lg" >> $R/gpurun_out/r06_timeline_mid_verify_d.txt
python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_c30_$lg/v_results.db 18 0 >> $R/gpurun_out/r06_timeline_mid_verify_d.txt 2>&1
done
rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_c30_20 -o v -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n 20 --steps 5 --warmup 2 > $R/gpurun_out/raw/kt_c30_20.log 2>&1
echo "== 2^20" >> $R/gpurun_out/r06_timeline_mid_verify_d.txt
python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_c30_20/v_results.db 23 0 >> $R/gpurun_out/r06_timeline_mid_verify_d.txt 2>&1
cut -c1-110 $R/gpurun_out/r06_timeline_mid_verify_d.txt
