#!/usr/bin/env python3
"""Three independent field products per iteration, three code shapes, at 8 / 3 / 1 waves per SIMD (G fe_mul/s, chip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curve25519_dalek_amd as pkg
e = pkg.Engine(0)
for _ in range(60):
    e.microbench(0, 4000)
names = {40: "3 x fe_mul, chained, one after another", 41: "fe_mul_chain_n<3>: chained, lockstep", 42: "3 x ten-column fe_mul"}
print("%-44s %10s %10s %10s" % ("shape", "8 waves", "3 waves", "1 wave"))
for w, nm in names.items():
    r = [max(e.microbench(w + o, 2000) for _ in range(5)) for o in (0, 200, 100)]
    print("%-44s %10.1f %10.1f %10.1f" % (nm, *r))
