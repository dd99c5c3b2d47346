#!/bin/bash
# round 6, gpurun call 17: the bucket reduction's products in the ten-column form (reduce.hip -DC25519_CHAIN=0: a wave does ONE product per stage, so the chained form is one dependent
# chain of 100 multiply-adds) against the chained form; the lean small direct path
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
( timeout 600 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -k "every_size or small or sum_of_squares or mid_path" 2>&1 | tail -4 ) > gpurun_out/r06_c17_tests.log 2>&1
( timeout 300 python tools/small_call_phases.py ) > gpurun_out/r06_small_call_phases_lean.txt 2>&1
out=gpurun_out/r06_ab_reduce_columns.txt; : > $out
export MIDRANGE_SIZES=16384,65536,131072,262144,1048576,2097152
for rep in 0 1; do
echo "## chained products in the reduction (tuning build), rep $rep" >> $out; C25519_HIP_LIB=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## ten-column products in the reduction (variant redcols), rep $rep" >> $out; C25519_HIP_LIB=$R/curve25519-dalek_amd/lib/libc25519hip_redcols.so timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
cat > /tmp/ab_red.cfg <<'CFG'
for rep in 1 2; do
run "msm 2^21 chained reduction, rep $rep" -- --log2n 21 --steps 200 --warmup 20
run "msm 2^21 ten-column reduction, rep $rep" C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_redcols.so -- --log2n 21 --steps 200 --warmup 20
run "msm 2^24 chained reduction, rep $rep" -- --log2n 24 --steps 40 --warmup 5
run "msm 2^24 ten-column reduction, rep $rep" C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_redcols.so -- --log2n 24 --steps 40 --warmup 5
run "verify 2^20 chained reduction, rep $rep" -- --workload verify --log2n 20 --steps 200 --warmup 20
run "verify 2^20 ten-column reduction, rep $rep" C25519_HIP_LIB=$PWD/curve25519-dalek_amd/lib/libc25519hip_redcols.so -- --workload verify --log2n 20 --steps 200 --warmup 20
done
CFG
bash tools/gpu_ab.sh r06_reduce_columns /tmp/ab_red.cfg > /dev/null 2>&1
cat gpurun_out/ab_r06_reduce_columns.log >> $out
tail -3 gpurun_out/r06_c17_tests.log; head -12 gpurun_out/r06_small_call_phases_lean.txt; cat $out
