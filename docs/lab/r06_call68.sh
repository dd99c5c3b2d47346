#!/bin/bash
# round 6, gpurun call 68: the lazy sign in the mid path's bucket lanes (A/B against libc25519hip_midsel.so) and in k_mul_base_wide (vartime fixed base: three runs, to compare with
# 0.526 ms per launch of call 59); the whole GPU suite first; the bench line last
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
L=$R/curve25519-dalek_amd/lib
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c68_tests.log 2>&1; tail -3 gpurun_out/r06_c68_tests.log
out=gpurun_out/r06_ab_lazy_sign_mid.txt; : > $out
for rep in 0 1 2; do for lib in tune midsel; do
  echo "## $lib rep $rep" >> $out
  C25519_HIP_LIB=$L/libc25519hip_$lib.so MIDRANGE_SIZES=8192,16384,32768,65536,131072,262144 timeout 300 python tools/midrange_numbers.py 2>/dev/null | cut -c1-56 >> $out
done; done
cat $out
for rep in 0 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-sub --workload fixed_base --fixed-base-variant vartime --steps 50 --warmup 3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('fixed_base vartime 2^20  %.4f ms  probe %.2f T  kernel %.4f ms  frac %.3f' % (d['ms_per_step'], r['peak'], r['kernel_ms_per_launch'], r['frac']))"; done | tee gpurun_out/r06_c68_wide.txt
timeout 600 python bench.py > gpurun_out/r06_bench_default_l.json 2> /dev/null; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_default_l.json").read().strip().splitlines()[-1])
s = d["summary"]
print(d["ms_per_step"], d["roofline"]["peak"], d["roofline"]["kernel_ms_per_launch"], d["roofline"]["frac"])
for k in ("msm_2p24", "verify_batch_2p20", "fixed_base_2p20", "fixed_base_2p20_vartime_tables", "x25519_2p20"): print(k, s[k])
print(json.dumps(s["mid_n_ms"]))
PY
