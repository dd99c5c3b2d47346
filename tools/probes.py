#!/usr/bin/env python3
"""Instruction-rate probes of the integer vector unit (c25519_microbench): giga-instructions per second for the whole chip
at 8 waves per SIMD, and at ONE wave per SIMD (which + 100: how much a lone wave can issue -- the latency-bound kernels).
    python tools/probes.py > profiles/rNN_instruction_rates.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)

e = pkg.Engine(0)
names = {0: "v_mad_u64_u32 (8 independent chains)", 22: "v_mad_u64_u32 (ONE dependent chain)", 5: "v_mul_lo_u32", 18: "v_mul_hi_u32", 16: "v_mul_u32_u24 (VOP2)",
         17: "v_mad_u32_u24", 4: "v_xad_u32 / add+xor pair", 8: "v_add_u32 (VOP2)", 13: "v_and_b32 (VOP2)", 14: "v_lshlrev_b32 (VOP2)", 12: "v_alignbit_b32",
         15: "v_and_or_b32", 19: "v_add3_u32", 20: "v_bfe_u32", 21: "v_lshl_add_u32", 10: "v_lshrrev_b64", 11: "v_lshl_add_u64",
         23: "v_lshrrev_b32 (VOP2)", 24: "v_sub_u32 (VOP2)", 25: "v_or_b32 (VOP2)", 26: "v_xor_b32 (VOP2)", 27: "v_cndmask_b32 (VOP2)", 28: "v_mov_b32", 36: "v_cndmask_b32_e64 (condition in an SGPR pair)", 37: "v_cndmask_b32_e32 (VCC written by a compare)",
         29: "v_perm_b32", 30: "v_add_u32_sdwa (src1 WORD_1)", 31: "v_lshl_or_b32", 32: "v_add_co_u32 + v_addc_co_u32 (pair = 1)",
         6: "v_mad_u64_u32 with 1 v_xad_u32 beside each", 7: "v_mad_u64_u32 with 2 v_xad_u32 beside each",
         33: "v_mad_u64_u32 with 1 v_add_u32 beside each", 34: "v_mad_u64_u32 with 2 v_add_u32 beside each", 35: "v_mad_u64_u32 with 3 v_add_u32 beside each",
         1: "fe_mul (chained carries)", 2: "fe_sq (chained carries)",
         70: "fe9_mul (9 limbs of 28.33 bits, chained: csrc/fe9_probe.h)", 71: "fe9_sq (9 limbs)"}
for _ in range(60):
    e.microbench(0, 4000)                      # bring the clock up
cus = 256
print("%-40s %12s %14s %12s %14s" % ("instruction", "Gop/s chip", "cycles/wave*", "1 wave/SIMD", "cycles/wave*"))
for which, nm in names.items():
    full = max(e.microbench(which, 4000) for _ in range(5))
    lone = max(e.microbench(which + 100, 4000) for _ in range(5))
    # cycles per wave-instruction on one SIMD at 2.4 GHz nominal: 1024 SIMDs x 64 lanes x f / rate
    cyc = lambda r: 1024 * 64 * 2.4 / r
    print("%-40s %12.1f %14.2f %12.1f %14.2f" % (nm, full, cyc(full), lone, cyc(lone)))
print("* at a nominal 2.4 GHz; fe_mul / fe_sq rows count field operations, not instructions")
# three independent field products per iteration in three code shapes, at 8 / 3 / 1 waves per SIMD (G fe_mul/s, chip):
# what k_accumulate (three waves) has to choose between
print()
print("%-44s %10s %10s %10s" % ("three independent products, shape", "8 waves", "3 waves", "1 wave"))
for w, nm in {40: "3 x fe_mul, chained, one after another", 41: "fe_mul_chain_n<3>: chained, in lockstep", 42: "3 x ten-column fe_mul"}.items():
    r = [max(e.microbench(w + o, 2000) for _ in range(5)) for o in (0, 200, 100)]
    print("%-44s %10.1f %10.1f %10.1f" % (nm, *r))

# ds_bpermute_b32 under different selector patterns (diag.hip k_probe_bpermute): the measurement behind the cross-lane table fetch of the
# constant-time fixed base (kernels.hip k_mul_base_ctp).  Throughput = eight independent permutes per trip at 8 waves per SIMD; latency = a chain of
# DEPENDENT permutes (s_waitcnt after each) at ONE wave per SIMD.  If the crossbar serialised on any selector pattern, that row would be slower.
print()
print("%-72s %14s %14s %16s" % ("ds_bpermute_b32, source lane of lane l", "Gop/s chip", "cycles/wave*", "latency cycles*"))
pats = {0: "identity (l)", 1: "all lanes pull lane 5", 2: "pseudo-random inside the own 32-lane half (k_mul_base_ctp<5>)", 3: "pairs 32 lanes apart inside a half (s, s + 32 alternating)",
        4: "pseudo-random over the whole wave", 5: "two sources only (lane 0 / lane 32)", 6: "rotate by one", 7: "pseudo-random inside the half, new selectors every trip"}
for i, k in enumerate((2, 4, 8, 16, 32)):
    pats[8 + i] = "many-to-one inside the half: groups of %d lanes pull one (scattered) source lane" % k
rows = []
for ppat, nm in pats.items():
    thr = max(e.microbench((50 + ppat) if ppat < 8 else (80 + ppat - 8), 4000) for _ in range(5))
    lat = max(e.microbench((160 + ppat) if ppat < 8 else (185 + ppat - 8), 2000) for _ in range(5))      # + 100: one wave per SIMD; the chain is dependent: rate = 1 / latency
    rows.append((thr, lat))
    print("%-72s %14.1f %14.2f %16.1f" % (nm, thr, 1024 * 64 * 2.4 / thr, 1024 * 64 * 2.4 / lat))
spread_t = max(r[0] for r in rows) / min(r[0] for r in rows)
spread_l = max(r[1] for r in rows) / min(r[1] for r in rows)
print("spread over the patterns: throughput x%.3f, latency x%.3f" % (spread_t, spread_l))
