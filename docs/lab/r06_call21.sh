#!/bin/bash
# round 6, gpurun call 21: no per-kernel events in the mid path of verify_batch either -- full suite, verify mid range, bench at mid sizes, the driver's command
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out/raw
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r06_c21_tests.log 2>&1
( VERIFY_SIZES=4096,8192,16384,32768,65536,131072 timeout 400 python tools/verify_midrange.py ) > gpurun_out/r06_verify_midrange_noring.txt 2>&1
( timeout 200 python bench.py --no-cpu-baseline --no-sub --workload verify --log2n 14 --steps 100 --warmup 10 | tail -c 500 ) > gpurun_out/r06_bench_v2p14.txt 2>&1
( python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r06_smoke.txt 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r06_bench_default_e.json 2> gpurun_out/r06_bench_default_e.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_midv_14 -o midv_14 -- python $R/bench.py --no-cpu-baseline --no-sub --workload verify --log2n 14 --steps 20 --warmup 3 > $R/gpurun_out/raw/kt_midv_14.log 2>&1
python $R/tools/timeline_all.py $R/gpurun_out/raw/kt_midv_14/midv_14_results.db k_slot_init 1 > $R/gpurun_out/r06_timeline_mid_verify_2p14.txt 2>&1
cd $R; rm -rf gpurun_out/raw/*/*.db
tail -3 gpurun_out/r06_c21_tests.log; cat gpurun_out/r06_verify_midrange_noring.txt; cat gpurun_out/r06_bench_v2p14.txt; echo; tail -2 gpurun_out/r06_smoke.txt; tail -3 gpurun_out/r06_bench_default_e.err; tail -c 700 gpurun_out/r06_bench_default_e.json; echo; cut -c1-110 gpurun_out/r06_timeline_mid_verify_2p14.txt | sed -n 12,24p
