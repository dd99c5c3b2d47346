#!/bin/bash
for rep in 1 2; do
for v in 0 1; do
export C25519_SORT_OVERLAP=$v
for w in msm verify; do
echo -n "overlap=$v $w: "; python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3))'
done; done; done
