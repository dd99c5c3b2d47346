#!/bin/bash
# A/B sweep on the GPU box: each line = one bench.py run (headline only), compact output to gpurun_out/ab.log
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/ab_${1:-run}.log
: > "$out"
run() {   # label, env..., -- bench args
    local label="$1"; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    local line
    # (the C25519_* knobs exist only in the tuning build; a cfg line may still name another library)
    line=$(env C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-sub "$@" 2>>gpurun_out/ab_err.log | tail -1)
    python3 - "$label" "$line" >> "$out" <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    t = d["roofline"]["timings_ms"]
    print("%-44s %8.3f ms/step  %.3e %s  valu %.3f  dom %.3f ms  %s" % (label, d["ms_per_step"], d["value"], d["unit"], d["roofline"]["whole_call"]["frac"], d["roofline"]["kernel_ms_per_launch"] or -1,
          " ".join("%s=%.3f" % (k[:14], v) for k, v in t.items())))
except Exception as e:
    print("%-44s FAILED %s %s" % (label, e, line[:200]))
PY
}
source "$2"
cat "$out"
