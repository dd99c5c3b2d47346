#!/bin/bash
# round 6, gpurun call 62: the translation units that still use the TEN-COLUMN products (small, single, finish, msm, extra, verify: chosen in rounds 2 - 4, when the chained
# form carried an s_nop per product) built in the CHAINED form (-DC25519_CHAIN=1), one unit at a time and all together, against the tuning build; same box, interleaved
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
L=$R/curve25519-dalek_amd/lib
out=gpurun_out/r06_ab_chain_everywhere.txt; : > $out
bline() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]
print("%-40s %.4f ms  probe %.2f T  %s %.4f ms per launch  %s" % (sys.argv[1], d["ms_per_step"], r["peak"], r["kernel"].split("(")[0].strip()[8:], r["kernel_ms_per_launch"], r.get("timings_ms")))
PY
}
for rep in 0 1; do for lib in tune chainall chain_small chain_finish chain_msm chain_single; do
  echo "## $lib rep $rep: small calls" >> $out
  C25519_HIP_LIB=$L/libc25519hip_$lib.so timeout 300 python tools/small_call_times.py 2>/dev/null | grep -v "n=  256 \|n= 1024 z" >> $out
  if [ $lib = tune ] || [ $lib = chainall ] || [ $lib = chain_msm ] || [ $lib = chain_finish ]; then
    for w in "fixed_base 20" "msm 21" "msm 24" "verify 20"; do set -- $w
      line=$(timeout 300 python bench.py --lib $L/libc25519hip_$lib.so --no-cpu-baseline --no-sub --workload $1 --log2n $2 --steps 20 --warmup 3 2>/dev/null | tail -1)
      bline "$lib $1 2^$2 rep $rep" "$line" >> $out
    done
    echo "## $lib rep $rep: MSM sizes" >> $out
    C25519_HIP_LIB=$L/libc25519hip_$lib.so MIDRANGE_SIZES=2048,4096,8192,65536,400000,1048576 timeout 300 python tools/midrange_numbers.py 2>/dev/null | cut -c1-56 >> $out
  fi
  if [ $lib = tune ] || [ $lib = chainall ] || [ $lib = chain_single ]; then
    echo "## $lib rep $rep: per-item paths (tools/extra_numbers.py)" >> $out
    C25519_HIP_LIB=$L/libc25519hip_$lib.so timeout 400 python tools/extra_numbers.py 2>/dev/null | cut -c1-160 >> $out
  fi
done; done
cat $out
cp $L/libc25519hip.so /tmp/rel.so; cp $L/libc25519hip_chainall.so $L/libc25519hip.so
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c62_tests_chainall.log 2>&1; tail -4 gpurun_out/r06_c62_tests_chainall.log
cp /tmp/rel.so $L/libc25519hip.so
