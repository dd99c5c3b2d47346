#!/bin/bash
# round 6, gpurun call 51: (a) MSM over COMPRESSED points (prepared records, like verify_batch's) around the small / mid boundary: MSM_SMALL_MAX sweep with forced 12-bit windows;
# (b) verify_batch with host pointers: the staged one-copy upload (ffi_small_upload) beyond 6143 signatures (VERIFY_STAGED_MAX) against the per-array copies of the general route
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_small_mid_boundary_compressed.txt; : > $out
SZ=1536,2048,3072,4096,5120,6143
echo "## compressed points, default (small path up to 6143 terms)" >> $out
C25519_HIP_LIB=$T MIDRANGE_FMT=0 MIDRANGE_SIZES=$SZ,6144,8192,12288 timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
for c in 11 12; do
  echo "## compressed points, MSM_SMALL_MAX=1535 MSM_CFORCE=$c (the mid path over prepared records)" >> $out
  C25519_HIP_LIB=$T C25519_VERIFY_SMALL_MAX=1023 C25519_MSM_SMALL_MAX=1535 C25519_MSM_CFORCE=$c MIDRANGE_FMT=0 MIDRANGE_SIZES=$SZ timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
done
echo "## raw points, MSM_SMALL_MAX=1535 MSM_CFORCE=12 (for comparison with the same box)" >> $out
C25519_HIP_LIB=$T C25519_MSM_SMALL_MAX=1535 C25519_MSM_CFORCE=12 MIDRANGE_SIZES=$SZ timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
echo "## raw points, default" >> $out
C25519_HIP_LIB=$T MIDRANGE_SIZES=$SZ timeout 200 python tools/midrange_numbers.py 2>/dev/null >> $out
cat $out
out=gpurun_out/r06_ab_verify_staged_upload.txt; : > $out
for rep in 0 1; do for sm in 12287 32769 131073; do
  echo "## verify_batch VERIFY_STAGED_MAX=$sm rep $rep" >> $out
  C25519_HIP_LIB=$T C25519_VERIFY_STAGED_MAX=$sm VERIFY_SIZES=6143,6144,8192,16384,32768,65536 timeout 300 python tools/verify_midrange.py 2>/dev/null >> $out
done; done
cat $out
