#!/bin/bash
# solo durations of the sort kernels of one 2^21-term pass (the sort serialised behind the normaliser): tools/sort_solo.sh <tag> [env...]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
mkdir -p $R/gpurun_out/raw
cd /tmp && export TMPDIR=/tmp
env C25519_HIP_LIB=$R/curve25519-dalek_amd/lib/libc25519hip_tune.so C25519_PROFILE_SERIAL_SORT=1 "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/raw/ss_$tag -o $tag -- python $R/bench.py --no-cpu-baseline --no-sub --log2n 21 --steps 10 --warmup 3 > $R/gpurun_out/raw/ss_$tag.log 2>&1
f=$(find $R/gpurun_out/raw/ss_$tag -name "*results.db" | head -1)
echo "== $tag $*"
python $R/tools/rocprof_summary.py $f 2>&1 | grep -E "k_sweep|k_seg_scan|k_part2|k_bin_totals|k_order_place|k_accumulate|k_prep_raw2" | cut -c1-140
grep -h '"metric"' $R/gpurun_out/raw/ss_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms_per_step', d['ms_per_step'])"
