#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_ab_small_rotate.txt
: > $out
export MIDRANGE_SIZES=64,256,1024,2048,4096,8192,12000,12287
T=$PWD/curve25519-dalek_amd/lib/libc25519hip_tune.so; N=$PWD/curve25519-dalek_amd/lib/libc25519hip_norot.so
for rep in 0 1; do
echo "## table-building wave rotates with the block index, rep $rep" >> $out; C25519_HIP_LIB=$T timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## wave 0 builds the tables of every block (before), rep $rep" >> $out; C25519_HIP_LIB=$N timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
echo "## rotating, small path forced up to 40000 terms (where does it meet the bucket pipeline now)" >> $out
MIDRANGE_SIZES=12288,16384,20000,24000,32768 C25519_HIP_LIB=$T timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
MIDRANGE_SIZES=12288,16384,20000,24000,32768 C25519_MSM_SMALL_MAX=40000 C25519_HIP_LIB=$T timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
cat $out
( timeout 600 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -k "every_size or random_sizes or small" 2>&1 | tail -3 )
