#!/bin/bash
# round 6, gpurun call 56: the final tree -- whole GPU suite, the driver's bench command, kernel statistics / timelines of every workload (tools/profile_all.sh r06 nopmc),
# bench.py under torch.distributed.run with one rank (RCCL initialised), two soaks
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c56_tests.log 2>&1; tail -4 gpurun_out/r06_c56_tests.log
bash tools/profile_all.sh r06 nopmc > gpurun_out/r06_c56_profile_all.log 2>&1; tail -3 gpurun_out/r06_c56_profile_all.log
cd $R
python - <<'PY'
import json
d = json.loads(open("gpurun_out/profiles_r06/r06_bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["peak"], json.dumps(d["summary"]["mid_n_ms"]), d["summary"]["verify_batch_2p20"][0])
PY
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-sub --no-cpu-baseline > gpurun_out/r06_c56_torchrun1.json 2> gpurun_out/r06_c56_torchrun1.err; tail -c 600 gpurun_out/r06_c56_torchrun1.json; tail -3 gpurun_out/r06_c56_torchrun1.err
for seed in 22 23; do timeout 600 python tools/soak_small.py 200000 $seed > gpurun_out/r06_soak_seed$seed.txt 2>&1; grep -E "^soak_small|counters" gpurun_out/r06_soak_seed$seed.txt | cut -c1-330; done
