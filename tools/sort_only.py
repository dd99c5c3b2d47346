#!/usr/bin/env python3
"""The sort of one MSM pass alone, N times (c25519_debug_sort), for a kernel trace without an accumulation beside it:
    rocprofv3 --kernel-trace --stats -d out -o s -- python tools/sort_only.py <terms> <layout_terms> [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)

n, layout = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
eng = pkg.Engine(0)
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
x[:, 31] &= 0x0F
torch.cuda.synchronize()
st = eng.lib.c25519_debug_sort(eng.ctx, x.data_ptr(), n, layout, reps)
assert st == 0, st
print("ok", n, layout, reps)
