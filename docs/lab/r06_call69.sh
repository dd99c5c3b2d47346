#!/bin/bash
# round 6, gpurun call 69: the FINAL tree -- whole GPU suite, every workload's kernel statistics, timelines, counters and the bench line (tools/profile_all.sh r06), a soak
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c69_tests.log 2>&1; tail -4 gpurun_out/r06_c69_tests.log
bash tools/profile_all.sh r06 > gpurun_out/r06_c69_profile_all.log 2>&1; tail -2 gpurun_out/r06_c69_profile_all.log
cd $R
python - <<'PY'
import json
d = json.loads(open("gpurun_out/profiles_r06/r06_bench_default.json").read().strip().splitlines()[-1])
s = d["summary"]
print(d["ms_per_step"], d["roofline"]["peak"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["frac"], d["roofline"]["frac_of_theoretical"])
for k in ("msm_2p24", "verify_batch_2p20", "fixed_base_2p20", "fixed_base_2p20_vartime_tables", "x25519_2p20"): print(k, s[k])
print(json.dumps(s["mid_n_ms"]))
PY
timeout 600 python tools/soak_small.py 200000 27 > gpurun_out/r06_soak_seed27.txt 2>&1; grep -E "^soak_small|counters" gpurun_out/r06_soak_seed27.txt | cut -c1-330
timeout 600 python bench.py > gpurun_out/r06_bench_default_m.json 2>/dev/null; tail -c 300 gpurun_out/r06_bench_default_m.json
