#!/bin/bash
# round 6, gpurun call 24: window width of the mid path, re-tuned (MSM_CFORCE on the tuning build): the path has no inversion and a cheaper sort than the pipeline the widths were chosen for
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_mid_window.txt; : > $out
for rep in 0 1; do
for c in 0 11 12 13 14 15 16; do
echo "## MSM_CFORCE=$c (0 = the rule: +3 bits over log2 n - 4 up to 2^16, +2 at 2^17, +1 at 2^18), rep $rep" >> $out
case $c in
 0) sizes=12288,16384,32768,65536,131072,262144;;
 11) sizes=12288,16384;;
 12) sizes=12288,16384,32768;;
 13) sizes=12288,16384,32768,65536;;
 14) sizes=16384,32768,65536,131072,262144;;
 15) sizes=32768,65536,131072,262144;;
 16) sizes=65536,131072,262144;;
esac
C25519_HIP_LIB=$T C25519_MSM_CFORCE=$c MIDRANGE_SIZES=$sizes timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
done
cat $out
