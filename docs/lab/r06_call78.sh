#!/bin/bash
# round 6, gpurun call 78: Keccak-f on 128-bit LANE PAIRS (15 xmm registers: column x, rows {0,1} / {2,3} / {4}) against the plane form and the scalar form, on the GPU box's host
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
out=gpurun_out/r06_keccak_pairs.txt; : > $out
grep -m1 "model name" /proc/cpuinfo >> $out
for cc in g++ /opt/rocm/lib/llvm/bin/clang++; do echo "## $cc" >> $out; $cc -O3 -std=c++17 -I curve25519-dalek_amd/csrc -I docs/lab/r06_keccak_pairs docs/lab/r06_keccak_pairs/t.cpp -o /tmp/t && /tmp/t >> $out; done
cat $out
