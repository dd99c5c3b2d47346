#!/usr/bin/env python3
"""ed25519_verify_batch at batch sizes whose 2n + 1-term MSM sits around the small path's upper boundary: device-resident (both z-modes) and host pointers.
   python tools/verify_midrange.py   (sizes: VERIFY_SIZES=1024,2048,...)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)
e = pkg.Engine(0)
for _ in range(40):
    e.microbench(0, 4000)
print("%8s | %12s %12s | %12s %12s   (ms per call, median)" % ("n", "device z=0", "device z=1", "host z=0", "host z=1"))
for n in [int(v) for v in os.environ.get("VERIFY_SIZES", "1024,2047,2048,3000,4096,5000,6143,6144,8192").split(",")]:
    seeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
    dm = torch.randint(0, 256, (59 * n,), dtype=torch.uint8, device="cuda"); doff = torch.arange(0, 59 * (n + 1), 59, dtype=torch.int64, device="cuda")
    dpk, dsg = e.sign_batch_t(seeds, dm, doff)
    M = dm.cpu().numpy(); P = dpk.cpu().numpy(); S = dsg.cpu().numpy()
    dpts = e.decompress_batch_t(dpk)[1] if os.environ.get('VERIFY_POINTS') else None      # VERIFY_POINTS=1: the device-resident columns with the keys' cached points (VerifyingKey)
    msgs = [M[59 * i:59 * i + 59].tobytes() for i in range(n)]; sigs = [S[i].tobytes() for i in range(n)]; pks = [P[i].tobytes() for i in range(n)]
    row = []
    for host in (0, 1):
        for zm in (0, 1):
            fn = (lambda: e.verify_batch(msgs, sigs, pks, zm)) if host else (lambda: e.verify_batch_t(dm, doff, dsg, dpk, zm, pk_points=dpts))
            for _ in range(3): assert fn() == 0
            ts = []
            for _ in range(25):
                torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            row.append(sorted(ts)[len(ts) // 2] * 1e3)
    print("%8d | %12.3f %12.3f | %12.3f %12.3f" % (n, *row))
