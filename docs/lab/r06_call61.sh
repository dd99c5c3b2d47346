#!/bin/bash
# round 6, gpurun call 61: the final tree (input pins, scalar-base gathers) -- whole GPU suite, kernel statistics / timelines / bench line of every workload
# (tools/profile_all.sh r06 nopmc), the counters of the MSM and fixed-base workloads again (the kernels changed), one soak
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c61_tests.log 2>&1; tail -4 gpurun_out/r06_c61_tests.log
bash tools/profile_all.sh r06 > gpurun_out/r06_c61_profile_all.log 2>&1; tail -3 gpurun_out/r06_c61_profile_all.log
cd $R
python - <<'PY'
import json
d = json.loads(open("gpurun_out/profiles_r06/r06_bench_default.json").read().strip().splitlines()[-1])
s = d["summary"]
print(d["ms_per_step"], d["roofline"]["peak"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["frac"])
for k in ("msm_2p24", "verify_batch_2p20", "fixed_base_2p20", "fixed_base_2p20_vartime_tables", "x25519_2p20"): print(k, s[k])
print(json.dumps(s["mid_n_ms"]))
PY
timeout 600 python tools/soak_small.py 200000 24 > gpurun_out/r06_soak_seed24.txt 2>&1; grep -E "^soak_small|counters" gpurun_out/r06_soak_seed24.txt | cut -c1-330
timeout 600 python tools/ffi_numbers.py > gpurun_out/r06_ffi_numbers_c61.txt 2>&1
timeout 300 python tools/small_call_times.py > gpurun_out/r06_small_call_times_c61.txt 2>&1; cat gpurun_out/r06_small_call_times_c61.txt
