#!/bin/bash
# round 6, gpurun call 58: (a) same-box A/B of k_accumulate's gather addressing (tuning build = the new 32-bit offsets, libc25519hip_gather64.so = the old 64-bit addresses);
# (b) the chained-carry pin as an INPUT of a volatile empty statement (libc25519hip_pin.so: the same instructions without ~0.5 s_nop per v_mad_u64_u32 in every chained-form
#     kernel): instruction-rate probes, small and mid-size calls, the throughput benchmarks; parity of that build by running the GPU suite with it in the release library's place
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
L=$R/curve25519-dalek_amd/lib
out=gpurun_out/r06_ab_accumulate_gather.txt; : > $out
bline() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]
print("%-34s %.4f ms  probe %.2f T  %s %.4f ms per launch  frac %.3f" % (sys.argv[1], d["ms_per_step"], r["peak"], r["kernel"].split("(")[0].strip()[8:], r["kernel_ms_per_launch"], r["frac"]))
PY
}
for rep in 0 1 2; do for lib in tune gather64; do for w in "msm 21" "msm 24" "verify 20"; do set -- $w
  line=$(timeout 300 python bench.py --lib $L/libc25519hip_$lib.so --no-cpu-baseline --no-sub --workload $1 --log2n $2 --steps 20 --warmup 3 2>/dev/null | tail -1)
  bline "$lib $1 2^$2 rep $rep" "$line" >> $out
done; done; done
cat $out
out=gpurun_out/r06_ab_pin_input.txt; : > $out
for lib in tune pin; do
  echo "## $lib: instruction-rate probes (tools/probes.py), fe rows" >> $out
  C25519_HIP_LIB=$L/libc25519hip_$lib.so timeout 300 python tools/probes.py 2>/dev/null | grep -E "fe_mul|fe_sq|fe9|three independent|3 x|chain_n|v_mad_u64_u32 \(" >> $out
done
for rep in 0 1; do for lib in tune pin; do
  echo "## $lib rep $rep: small calls (tools/small_call_times.py)" >> $out
  C25519_HIP_LIB=$L/libc25519hip_$lib.so timeout 300 python tools/small_call_times.py 2>/dev/null >> $out
  echo "## $lib rep $rep: mid-size MSM" >> $out
  C25519_HIP_LIB=$L/libc25519hip_$lib.so MIDRANGE_SIZES=4096,8192,16384,65536,262144 timeout 300 python tools/midrange_numbers.py 2>/dev/null | cut -c1-56 >> $out
  echo "## $lib rep $rep: verify_batch (cached points)" >> $out
  C25519_HIP_LIB=$L/libc25519hip_$lib.so VERIFY_POINTS=1 VERIFY_SIZES=1024,4096,16384,65536 timeout 300 python tools/verify_midrange.py 2>/dev/null | cut -c1-40 >> $out
  for w in "fixed_base 20" "x25519 20" "verify 20" "msm 21"; do set -- $w
    line=$(timeout 300 python bench.py --lib $L/libc25519hip_$lib.so --no-cpu-baseline --no-sub --workload $1 --log2n $2 --steps 20 --warmup 3 2>/dev/null | tail -1)
    bline "$lib $1 2^$2 rep $rep" "$line" >> $out
  done
done; done
cat $out
cp $L/libc25519hip.so /tmp/rel.so; cp $L/libc25519hip_pin.so $L/libc25519hip.so
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06_c58_tests_pin.log 2>&1; tail -4 gpurun_out/r06_c58_tests_pin.log
cp /tmp/rel.so $L/libc25519hip.so
