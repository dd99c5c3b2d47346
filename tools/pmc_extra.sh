#!/bin/bash
# Extra counter passes (cache behaviour of the gather-fed kernels) for one workload; run ON THE GPU BOX through gpurun.
#   bash tools/pmc_extra.sh msm r01   ->  gpurun_out/profiles_r01/r01_msm_pmc_cache.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-msm}
TAG=${2:-r01}
OUT=$R/gpurun_out/profiles_$TAG
RAW=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
i=0
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY" "MemUnitStalled" "L2CacheHit"; do
  i=$((i+1))
  C25519_BENCH_PROBES=2 timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $RAW/pmcx_${W}_$i -o p -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $RAW/pmcx_${W}_$i.log 2>&1 || echo "pass $i ($c) failed" 
done
python $R/tools/pmc_summary.py $RAW/pmcx_${W}_* > $OUT/${TAG}_${W}_pmc_cache.txt 2>&1
grep -A12 "k_accumulate\|k_mul_base_wide" $OUT/${TAG}_${W}_pmc_cache.txt | head -60
