#!/bin/bash
# round 6, gpurun call 49: where the small path (Straus tables, <= 12 287 terms since round 5, tuned against the OLD bucket pipeline) should hand over to the mid path
# (mid.hip): MSM_SMALL_MAX sweep for the MSM (raw points) and verify_batch, then window widths of the mid path at the new small sizes
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_small_mid_boundary.txt; : > $out
SZ=1536,2048,3072,4096,6144,8192,10240,12287
rm -f gpurun_out/dump_*.txt
for sm in 12287 8191 6143 4095 3071 2047 1023; do
  echo "## MSM_SMALL_MAX=$sm" >> $out
  C25519_HIP_LIB=$T C25519_MSM_SMALL_MAX=$sm MIDRANGE_DUMP=gpurun_out/dump_$sm.txt MIDRANGE_SIZES=$SZ timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
for sm in 8191 6143 4095 3071 2047 1023; do cmp gpurun_out/dump_12287.txt gpurun_out/dump_$sm.txt && echo "results of MSM_SMALL_MAX=$sm identical to the default's" >> $out; done
for c in 8 9 10 11 12 13; do
  echo "## MSM_SMALL_MAX=1023 MSM_CFORCE=$c" >> $out
  C25519_HIP_LIB=$T C25519_MSM_SMALL_MAX=1023 C25519_MSM_CFORCE=$c MIDRANGE_SIZES=$SZ timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
VS=768,1024,1536,2048,3072,4096,5120,6143
for sm in 12287 8191 4095 2047; do
  echo "## verify_batch MSM_SMALL_MAX=$sm" >> $out
  C25519_HIP_LIB=$T C25519_MSM_SMALL_MAX=$sm VERIFY_SIZES=$VS timeout 300 python tools/verify_midrange.py 2>/dev/null >> $out
done
for c in 9 10 11 12; do
  echo "## verify_batch MSM_SMALL_MAX=2047 VERIFY_C=$c" >> $out
  C25519_HIP_LIB=$T C25519_MSM_SMALL_MAX=2047 C25519_VERIFY_C=$c VERIFY_SIZES=$VS timeout 300 python tools/verify_midrange.py 2>/dev/null >> $out
done
cat $out
