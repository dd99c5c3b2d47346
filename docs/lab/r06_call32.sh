#!/bin/bash
# round 6, gpurun call 32: the whole GPU suite, smoke and the default bench line on the tree with the BLAKE2b z-tree and verify_batch's mid path on the hash chain's stream
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c32_tests.log 2>&1; tail -5 gpurun_out/r06_c32_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r06_bench_default_g.json 2> gpurun_out/r06_bench_default_g.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_default_g.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], json.dumps(d.get("mid_n"))[:600])
PY
