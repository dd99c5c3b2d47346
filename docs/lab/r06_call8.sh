#!/bin/bash
# round 6, gpurun call 8 (box 2 of the soak): full GPU suite, soak, the driver's bench command, the MALL A/B with PMC
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06_c8_tests.log 2>&1
( timeout 900 python tools/soak_small.py 200000 2 ) > gpurun_out/r06_soak_box2.txt 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r06_bench_default_a.json 2> gpurun_out/r06_bench_default_a.err
bash tools/gpu_ab.sh r06_mall docs/lab/ab_r06_mall.cfg > /dev/null 2>&1
# PMC: FETCH_SIZE and duration of k_accumulate with and without the streaming normaliser (separate passes, tuning build)
cd /tmp && export TMPDIR=/tmp
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
for arm in default nt; do
  if [ $arm = nt ]; then export C25519_PREP_NT=1 C25519_SWEEP_NT=1; else unset C25519_PREP_NT C25519_SWEEP_NT; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    C25519_HIP_LIB=$T C25519_BENCH_PROBES=2 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/raw/mall_${arm}_$c -o p -- python $R/bench.py --log2n 24 --steps 3 --warmup 1 --no-cpu-baseline --no-sub > $R/gpurun_out/raw/mall_${arm}_$c.log 2>&1 || echo "pmc $arm $c failed"
  done
done
unset C25519_PREP_NT C25519_SWEEP_NT
cd $R
python tools/pmc_summary.py gpurun_out/raw/mall_default_* > gpurun_out/r06_mall_pmc_default.txt 2>&1
python tools/pmc_summary.py gpurun_out/raw/mall_nt_* > gpurun_out/r06_mall_pmc_nt.txt 2>&1
rm -rf gpurun_out/raw/mall_*/*/*.db
tail -4 gpurun_out/r06_c8_tests.log; tail -14 gpurun_out/r06_soak_box2.txt; tail -3 gpurun_out/r06_bench_default_a.err; tail -c 2500 gpurun_out/r06_bench_default_a.json; echo; cat gpurun_out/ab_r06_mall.log; head -30 gpurun_out/r06_mall_pmc_default.txt; head -30 gpurun_out/r06_mall_pmc_nt.txt
