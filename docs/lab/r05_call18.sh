#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/raw
OUT=$R/gpurun_out/r05_timeline_small_calls.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "msm 1 1" "msm 1024 2" "verify 4 3" "verify 64 3"; do
  set -- $spec
  rm -rf $R/gpurun_out/raw/kt_s_$1_$2
  timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/raw/kt_s_$1_$2 -o s -- python $R/tools/small_trace.py $1 $2 > $R/gpurun_out/raw/kt_s_$1_$2.log 2>&1
  echo "## $1, n = $2: the kernels of the last call (start, end, duration in us; queue)" >> $OUT
  python $R/tools/timeline_tail.py $(find $R/gpurun_out/raw/kt_s_$1_$2 -name '*results.db' | head -1) $3 >> $OUT 2>&1
done
cat $OUT
cd $R
for i in 1 2; do ( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/r05_c18_soak_$i.log 2>&1; tail -1 gpurun_out/r05_c18_soak_$i.log; done
