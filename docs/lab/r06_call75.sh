#!/bin/bash
# round 6, gpurun call 75: scalar Keccak-f with the state in memory, row by row (docs/lab/r06_keccak_rows/) against the register form of transcript_host.h, on the GPU box's host
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
out=gpurun_out/r06_keccak_rows.txt; : > $out
grep -m1 "model name" /proc/cpuinfo >> $out
for cc in g++ /opt/rocm/lib/llvm/bin/clang++; do echo "## $cc" >> $out; $cc -O3 -std=c++17 -I curve25519-dalek_amd/csrc -I docs/lab/r06_keccak_rows docs/lab/r06_keccak_rows/t.cpp -o /tmp/t && /tmp/t >> $out; done
cat $out
