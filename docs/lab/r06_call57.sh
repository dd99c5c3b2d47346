#!/bin/bash
# round 6, gpurun call 57: k_accumulate's gathers with a scalar LDS base and 32-bit byte offsets from a scalar record base (1253 -> 1188 vector instructions per addition,
# 168 -> 158 registers): parity (MSM, verify, extra, multi), then the headline and the sub-benchmarks that use the kernel
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_verify.py tests/test_gpu_extra.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r06_c57_tests.log 2>&1; tail -4 gpurun_out/r06_c57_tests.log
out=gpurun_out/r06_ab_accumulate_gather.txt; : > $out
for rep in 0 1; do
for w in "msm 24" "msm 21" "verify 20"; do set -- $w
  line=$(timeout 300 python bench.py --no-cpu-baseline --no-sub --workload $1 --log2n $2 --steps 20 --warmup 3 2>/dev/null | tail -1)
  python3 - "$1 2^$2" "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]
print("%-12s %.4f ms  probe %.2f T  %s %.4f ms per launch  frac %.3f" % (sys.argv[1], d["ms_per_step"], r["peak"], r["kernel"].split("(")[0].strip(), r["kernel_ms_per_launch"], r["frac"]), r.get("timings_ms"))
PY
done; done
cat $out
