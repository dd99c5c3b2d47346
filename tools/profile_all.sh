#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats and PMC passes for every workload.
# Raw output under gpurun_out/prof_rNN/, summaries under gpurun_out/profiles_rNN/ (copy those to profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
OUT=$R/gpurun_out/profiles_$TAG
RAW=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $RAW
# microbench FIRST: rocprofv3 counter passes can leave the GPU in a lower profiling clock state
cd $R
python - <<'PY' > $OUT/${TAG}_microbench.txt 2>&1
import curve25519_dalek_amd as pkg
e = pkg.Engine(0)
for i, nm in enumerate(["v_mad_u64_u32", "fe_mul (radix 2^25.5, 10 x u32)", "fe_sq", "fe_mul (5 x u64, u128 products)", "v_add_u32+v_xor_b32 pairs", "v_mul_lo_u32"]):
    cold = e.microbench(i, 4000) if i == 0 else None
    warm = max(e.microbench(i, 4000) for _ in range(100))          # ~60 ms of sustained load: the GPU is at its sustained clock
    print("%-36s %10.1f Gop/s%s" % (nm, warm, "   (first probe on an idle GPU: %.1f)" % cold if cold else ""))
PY
(rocm-smi --showclocks --showpower 2>/dev/null | head -30) > $OUT/${TAG}_rocm_smi.txt
cd /tmp && export TMPDIR=/tmp
for w in fixed_base x25519 msm verify; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $RAW/kt_$w -o $w -- python $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $RAW/kt_$w.log 2>&1
  python $R/tools/rocprof_summary.py $RAW/kt_$w/${w}_results.db > $OUT/${TAG}_${w}_kernel_stats.txt 2>&1
  grep -h '"metric"' $RAW/kt_$w.log >> $OUT/${TAG}_${w}_kernel_stats.txt
done
# PMC passes: counters only with --kernel-trace (never with sys/hip/hsa tracing)
for w in fixed_base verify x25519 msm; do
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "VALUBusy" "OccupancyPercent" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    C25519_BENCH_PROBES=2 timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $RAW/pmc_${w}_$i -o p -- python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $RAW/pmc_${w}_$i.log 2>&1
  done
  python $R/tools/pmc_summary.py $RAW/pmc_${w}_* > $OUT/${TAG}_${w}_pmc.txt 2>&1
done
ls -la $OUT
