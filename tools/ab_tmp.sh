#!/bin/bash
for p in 2 24 100; do
export C25519_BENCH_PROBES=$p
for rep in 1 2; do
echo -n "probes=$p: "; python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["valu"]["peak"],2))'
done; done
echo -n "with cpu baseline: "; python bench.py 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["valu"]["peak"],2))'
