#!/bin/bash
# kernel-trace stats of one bench.py invocation: tools/kt.sh <tag> <bench args...>  -> gpurun_out/kt_<tag>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
mkdir -p $R/gpurun_out/raw
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/raw/kt_$tag -o $tag -- python $R/bench.py --no-cpu-baseline --no-sub "$@" > $R/gpurun_out/raw/kt_$tag.log 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/raw/kt_$tag/${tag}_results.db > $R/gpurun_out/kt_$tag.txt 2>&1
grep -h '"metric"' $R/gpurun_out/raw/kt_$tag.log >> $R/gpurun_out/kt_$tag.txt
