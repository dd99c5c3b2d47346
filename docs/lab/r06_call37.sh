#!/bin/bash
# round 6, gpurun calls 37 / 38: the soak of the published paths with the mid path's calls mixed in (tools/soak_small.py; seed = $1)
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
timeout 1500 python tools/soak_small.py ${2:-200000} ${1:-11} > gpurun_out/r06_soak_mid_seed${1:-11}.txt 2>&1; tail -25 gpurun_out/r06_soak_mid_seed${1:-11}.txt
