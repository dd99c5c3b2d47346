#!/bin/bash
# round 6, gpurun call 59: k_mul_base_wide's table fetch with a scalar LDS base, a 32-bit entry offset from the scalar table base and the pieces by the instruction's offset field
# (1146 -> 1129 vector instructions per window); the tree with the input pins and the new gathers: whole GPU suite, bench line, vartime fixed base three times
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c59_tests.log 2>&1; tail -4 gpurun_out/r06_c59_tests.log
for rep in 0 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-sub --workload fixed_base --fixed-base-variant vartime --steps 50 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('fixed_base vartime 2^20  %.4f ms  probe %.2f T  kernel %.4f ms  frac %.3f' % (d['ms_per_step'], r['peak'], r['kernel_ms_per_launch'], r['frac']))"; done | tee gpurun_out/r06_c59_wide.txt
timeout 600 python bench.py > gpurun_out/r06_bench_default_j.json 2> gpurun_out/r06_bench_default_j.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_default_j.json").read().strip().splitlines()[-1])
s = d["summary"]
print(d["ms_per_step"], d["roofline"]["peak"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["frac"])
for k in ("msm_2p24", "verify_batch_2p20", "fixed_base_2p20", "fixed_base_2p20_vartime_tables", "x25519_2p20"): print(k, s[k])
print(json.dumps(s["mid_n_ms"])); print(json.dumps(s["small_n_us_at_sizes"]))
PY
