#!/bin/bash
# mid-range timelines: where a 2^14 .. 2^18-term call spends its time
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/raw
cd /tmp && export TMPDIR=/tmp
for cfg in "msm 18" "msm 16" "msm 14" "verify 14" "verify 16"; do
  set -- $cfg
  tag=${1}_2p$2
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/raw/kt_$tag -o $tag -- python $R/bench.py --workload $1 --log2n $2 --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/raw/kt_$tag.log 2>&1
  first=k_slot_init; [ "$1" = "verify" ] && first=k_prep_basepoint
  python $R/tools/rocprof_timeline.py $R/gpurun_out/raw/kt_$tag/${tag}_results.db $first > $R/gpurun_out/r05_timeline_$tag.txt 2>&1
  grep -h '"metric"' $R/gpurun_out/raw/kt_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$tag', d['ms_per_step'], d['roofline']['timings_ms'])"
done
cd $R; for f in gpurun_out/r05_timeline_*.txt; do echo "== $f"; cut -c1-130 $f | tail -n +2; done
