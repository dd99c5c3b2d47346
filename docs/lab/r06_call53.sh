#!/bin/bash
# round 6, gpurun call 53: (a) k_mid_acc_long at FOUR waves per SIMD (128 registers, seven scratch accesses per addition) against three, 2^13 .. 2^18 terms;
# (b) would a hipGraph shorten a MID-SIZE call?  tools/graph_probe.cpp, second shape: seven dependent kernels, no copies, arguments patched per replay
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_ab_mid_acc_waves.txt; : > $out
rm -f gpurun_out/dumpw_*.txt
for rep in 0 1; do for w in 0 1; do
  echo "## MID_ACC_WAVES4_MIN=$w rep $rep (0 = three waves per SIMD everywhere; 1 = four everywhere)" >> $out
  C25519_HIP_LIB=$T C25519_MID_ACC_WAVES4_MIN=$w MIDRANGE_DUMP=gpurun_out/dumpw_${w}_$rep.txt MIDRANGE_SIZES=8192,16384,32768,65536,100000,131072,200000,262144 timeout 300 python tools/midrange_numbers.py 2>/dev/null | cut -c1-56 >> $out
done; done
cmp gpurun_out/dumpw_0_0.txt gpurun_out/dumpw_1_0.txt && echo "results identical" >> $out
cat $out
hipcc --offload-arch=gfx950 -O2 -w tools/graph_probe.cpp -o gpurun_out/graph_probe && { gpurun_out/graph_probe; gpurun_out/graph_probe; } > gpurun_out/r06_graph_probe.txt 2>&1; cat gpurun_out/r06_graph_probe.txt; rm -f gpurun_out/graph_probe
