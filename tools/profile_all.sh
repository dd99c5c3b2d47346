#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace stats and PMC passes for every workload of bench.py.
# Raw output under gpurun_out/prof_rNN/, summaries under gpurun_out/profiles_rNN/ (copy those to profiles/).
#   bash tools/profile_all.sh r03 [nopmc]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
OUT=$R/gpurun_out/profiles_$TAG
RAW=$R/gpurun_out/prof_$TAG
mkdir -p $OUT $RAW $R/gpurun_out/raw
# instruction-rate probes FIRST: rocprofv3 counter passes can leave the GPU in a lower profiling clock state
cd $R
python tools/probes.py > $OUT/${TAG}_instruction_rates.txt 2>&1
(rocm-smi --showclocks --showpower 2>/dev/null | head -30) > $OUT/${TAG}_rocm_smi.txt
# the driver's own command: one line with the headline and all sub-records
python bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench_default.json 2> $RAW/bench_default.err
cd /tmp && export TMPDIR=/tmp
# workload tag : bench.py arguments
declare -A ARGS=( [msm]="--workload msm --log2n 24" [msm_2p21]="--workload msm --log2n 21" [verify]="--workload verify" \
                  [fixed_base_ct]="--workload fixed_base" [fixed_base]="--workload fixed_base --fixed-base-variant vartime" [x25519]="--workload x25519" )
for w in msm msm_2p21 verify fixed_base_ct fixed_base x25519; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $RAW/kt_$w -o $w -- python $R/bench.py ${ARGS[$w]} --steps 5 --warmup 2 --no-cpu-baseline > $RAW/kt_$w.log 2>&1
  python $R/tools/rocprof_summary.py $RAW/kt_$w/${w}_results.db > $OUT/${TAG}_${w}_kernel_stats.txt 2>&1
  grep -h '"metric"' $RAW/kt_$w.log >> $OUT/${TAG}_${w}_kernel_stats.txt
done
python $R/tools/rocprof_timeline.py $RAW/kt_msm_2p21/msm_2p21_results.db k_slot_init > $OUT/${TAG}_msm_2p21_timeline.txt 2>&1
python $R/tools/timeline_tail.py $RAW/kt_msm/msm_results.db 75 15 > $OUT/${TAG}_msm_2p24_last_call_timeline.txt 2>&1
python $R/tools/rocprof_timeline.py $RAW/kt_verify/verify_results.db k_prep_basepoint > $OUT/${TAG}_verify_timeline.txt 2>&1
[ "$2" = "nopmc" ] && { ls -la $OUT; exit 0; }
# PMC passes: counters only with --kernel-trace (never with sys/hip/hsa tracing); separate passes per counter group
for w in msm_2p21 verify fixed_base_ct fixed_base x25519; do
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "VALUBusy" "OccupancyPercent" "MemUnitStalled" "L2CacheHit"; do
    i=$((i+1))
    C25519_BENCH_PROBES=2 timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $RAW/pmc_${w}_$i -o p -- python $R/bench.py ${ARGS[$w]} --steps 3 --warmup 1 --no-cpu-baseline > $RAW/pmc_${w}_$i.log 2>&1 || echo "pass $i ($c) of $w failed"
  done
  name=$w; [ "$w" = "msm_2p21" ] && name=msm
  python $R/tools/pmc_summary.py $RAW/pmc_${w}_* > $OUT/${TAG}_${name}_pmc.txt 2>&1
done
ls -la $OUT
