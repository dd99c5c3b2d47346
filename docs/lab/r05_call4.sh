#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_raw160.py tests/test_gpu_msm.py -m gpu -x -q -k "oracle or every_size" 2>&1 | tail -8 ) > gpurun_out/r05_c4_tests.log 2>&1
bash tools/gpu_ab.sh r05d docs/lab/ab_r05_d.cfg > /dev/null 2>&1
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 2 --log2n 21 --no-sub --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps(d.get('multi_gpu'), indent=1)); print(d['ms_per_step'])" ) > gpurun_out/r05_c4_multi.log 2>&1
tail -4 gpurun_out/r05_c4_tests.log; cat gpurun_out/ab_r05d.log; cat gpurun_out/r05_c4_multi.log
