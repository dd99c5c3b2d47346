#!/bin/bash
# round 6, gpurun call 29: host clock of mid-size verify_batch calls (tools/verify_call_phases.py): is the call bound by the host's launch loop?
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
T=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so
out=gpurun_out/r06_verify_call_phases.txt; : > $out
echo "## default" >> $out; timeout 300 python tools/verify_call_phases.py >> $out 2>&1
echo "## MID_ON_CHAIN=0 (tuning build)" >> $out; C25519_HIP_LIB=$T C25519_MID_ON_CHAIN=0 timeout 300 python tools/verify_call_phases.py >> $out 2>&1
echo "## MID_ON_CHAIN=0 CHAIN_FIRST=0 MID_LONG_BESIDE=1 VERIFY_DIRECT=0 (tuning build: the state before call 26)" >> $out; C25519_HIP_LIB=$T C25519_MID_ON_CHAIN=0 C25519_CHAIN_FIRST=0 C25519_MID_LONG_BESIDE=1 timeout 300 python tools/verify_call_phases.py >> $out 2>&1
cat $out
