#!/bin/bash
# round 6, gpurun call 52: the tree with the new small / mid boundaries (raw 6144 terms, encoded 4096, verify_batch 2048 signatures; staged upload up to 16 384 signatures):
# the whole GPU suite, a 200 000-call soak, the bench line, the C-level FFI numbers, boundary numbers for encoded points, a timeline of an 8192-term call
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c52_tests.log 2>&1; tail -4 gpurun_out/r06_c52_tests.log
timeout 600 python tools/soak_small.py 200000 21 > gpurun_out/r06_soak_seed21.txt 2>&1; grep -E "^soak_small|counters" gpurun_out/r06_soak_seed21.txt | cut -c1-400
timeout 600 python bench.py > gpurun_out/r06_bench_default_i.json 2> gpurun_out/r06_bench_default_i.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_default_i.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["peak"], json.dumps(d["summary"]["mid_n_ms"]), d["summary"]["verify_batch_2p20"][0])
print(json.dumps(d["summary"]["small_n_us_at_sizes"]))
PY
timeout 600 python tools/ffi_numbers.py > gpurun_out/r06_ffi_numbers_c52.txt 2>&1; tail -40 gpurun_out/r06_ffi_numbers_c52.txt
echo "## encoded points, release" > gpurun_out/r06_c52_encoded.txt
MIDRANGE_FMT=0 MIDRANGE_SIZES=3072,4095,4096,5120,6143,6144,8192 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> gpurun_out/r06_c52_encoded.txt; cat gpurun_out/r06_c52_encoded.txt
cd /tmp && export TMPDIR=/tmp
for w in "msm 13" "verify 12"; do set -- $w
rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_c52_$1 -o v -- python $R/bench.py --no-cpu-baseline --no-sub --workload $1 --log2n $2 --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_c52_$1.log 2>&1
echo "== $1 2^$2"; python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_c52_$1/v_results.db 8 0 | cut -c1-110
done
