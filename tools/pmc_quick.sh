#!/bin/bash
# a few derived counters for one bench invocation, one pass per counter: tools/pmc_quick.sh <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
mkdir -p $R/gpurun_out/raw
cd /tmp && export TMPDIR=/tmp
i=0
for c in "VALUBusy" "MemUnitStalled" "OccupancyPercent" "LDSBankConflict" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "MemUnitBusy" "L2CacheHit" "FETCH_SIZE"; do
  i=$((i+1))
  C25519_BENCH_PROBES=2 timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/raw/pq_${tag}_$i -o p -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-sub > $R/gpurun_out/raw/pq_${tag}_$i.log 2>&1 || echo "pass $i ($c) failed"
done
python $R/tools/pmc_summary.py $R/gpurun_out/raw/pq_${tag}_* > $R/gpurun_out/pq_$tag.txt 2>&1
