#!/bin/bash
# round 6, gpurun call 63: terms per pass of the 2^24-term call (MSM_PASS_TERMS; 1.75 M since round 3: the pass's gather records inside the MALL) -- every pass costs a ~150 us gap in
# front of its accumulation (the next pass's k_part2g finishes only once the running k_accumulate has drained: profiles/r06_msm_2p24_last_call_timeline.txt)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_pass_terms.txt; : > $out
for rep in 0 1; do for pt in 1750000 1400000 2097152 2400000 2800000 3400000 4194304; do
  line=$(env C25519_MSM_PASS_TERMS=$pt timeout 300 python bench.py --lib $T --no-cpu-baseline --no-sub --workload msm --log2n 24 --steps 20 --warmup 3 2>/dev/null | tail -1)
  python3 - "$pt" "$rep" "$line" >> $out <<'PY'
import json, sys
d = json.loads(sys.argv[3]); r = d["roofline"]
print("MSM_PASS_TERMS=%-8s rep %s  %.4f ms  probe %.2f T  k_accumulate %.4f ms x %d launches  %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], r["peak"], r["kernel_ms_per_launch"], r["launches_per_step"], r.get("timings_ms")))
PY
done; done
cat $out
