#!/bin/bash
# round 6, gpurun call 66: k_accumulate with the sign handled LAZILY on the accumulator (two conditional negations as v_xad_u32: 20 instructions) instead of by operand selection (40
# v_cndmask_b32): the whole GPU suite (the bound-checking build's tests included), then a same-box A/B against libc25519hip_signsel.so (-DC25519_ACC_SIGN_SELECT)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
L=$R/curve25519-dalek_amd/lib
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c66_tests.log 2>&1; tail -4 gpurun_out/r06_c66_tests.log
out=gpurun_out/r06_ab_lazy_sign.txt; : > $out
for rep in 0 1 2 3; do for lib in tune signsel; do for w in "msm 21" "msm 24" "verify 20"; do set -- $w
  line=$(timeout 300 python bench.py --lib $L/libc25519hip_$lib.so --no-cpu-baseline --no-sub --workload $1 --log2n $2 --steps 20 --warmup 3 2>/dev/null | tail -1)
  python3 - "$lib $1 2^$2 rep $rep" "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]; t = r.get("timings_ms", {})
print("%-30s %.4f ms  probe %.2f T  k_accumulate %.4f ms per launch" % (sys.argv[1], d["ms_per_step"], r["peak"], t.get("k_accumulate_ms_per_launch", 0)))
PY
done; done; done
cat $out
