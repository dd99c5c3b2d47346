#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/raw
OUT=$R/gpurun_out/r05_small_path_kernel_stats.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for n in 1024 4096 8192 12000; do
  rm -rf $R/gpurun_out/raw/kt_sp_$n
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/raw/kt_sp_$n -o s -- python $R/tools/small_trace.py msm $n > $R/gpurun_out/raw/kt_sp_$n.log 2>&1
  echo "## c25519_msm_vartime, $n terms, host pointers (six calls): kernels of the LAST call, then the per-kernel statistics of the run" >> $OUT
  python $R/tools/timeline_tail.py $(find $R/gpurun_out/raw/kt_sp_$n -name '*results.db' | head -1) 3 >> $OUT 2>&1
  python $R/tools/rocprof_summary.py $(find $R/gpurun_out/raw/kt_sp_$n -name '*results.db' | head -1) 2>&1 | grep -i "small\|name\|kernel" | head -6 >> $OUT
done
cat $OUT
