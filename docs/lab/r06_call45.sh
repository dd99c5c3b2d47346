#!/bin/bash
# round 6, gpurun call 45: the bucket reduction with the segment totals pre-multiplied by 2^6 in level A (six doublings and one addition fewer in level B) -- the whole GPU suite, then the bench line and a timeline
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c45_tests.log 2>&1; tail -4 gpurun_out/r06_c45_tests.log
timeout 600 python bench.py > gpurun_out/r06_bench_default_h.json 2> gpurun_out/r06_bench_default_h.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_default_h.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["peak"], json.dumps(d["summary"]["mid_n_ms"]), d["summary"]["verify_batch_2p20"][0])
PY
cd /tmp && export TMPDIR=/tmp
for w in "msm 14" "verify 14"; do set -- $w
rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_c45_$1 -o v -- python $R/bench.py --no-cpu-baseline --no-sub --workload $1 --log2n $2 --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_c45_$1.log 2>&1
echo "== $1 2^$2"; python $R/tools/timeline_tail.py $R/gpurun_out/raw/kt_c45_$1/v_results.db 7 0 | cut -c1-100
done
