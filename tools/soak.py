#!/usr/bin/env python3
"""Determinism soak on the GPU: the same 2^24-term MSM, 2^21-term MSM, 2^20-signature verify_batch (both z-modes' device
parts) and the per-item kernels, many times over; every repetition must reproduce the first result byte for byte (a race in
the stream / event / LDS-DMA choreography shows up as a flaky result long before it shows up as a wrong one).
    python tools/soak.py [repetitions]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)
E = pkg.engine
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
eng = pkg.Engine(0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
def rnd(n, w=32):
    t = torch.randint(0, 256, (n, w), dtype=torch.uint8, device=dev, generator=g)
    if w == 32: t[:, 31] &= 0x0F
    return t
t0 = time.time()
xs = rnd(1 << 24); ys = rnd(1 << 24)
pts = torch.empty((1 << 24, 160), dtype=torch.uint8, device=dev)
for lo in range(0, 1 << 24, 1 << 21): eng.mul_base_batch_vartime_t(ys[lo:lo + (1 << 21)], E.FMT_RAW160, pts[lo:lo + (1 << 21)])
del ys
first = {}
def check(name, val):
    if name not in first: first[name] = val
    assert first[name] == val, (name, "repetition differs")
n = 1 << 16
seeds = rnd(n); msgs = rnd(n); off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
pks, sigs = eng.sign_batch_t(seeds, msgs.reshape(-1), off)
for r in range(reps):
    st, out = eng.msm_vartime_t(xs, pts, E.FMT_RAW160, E.FMT_EDWARDS_Y); assert st == 0; check("msm24", bytes(out))
    st, out = eng.msm_vartime_t(xs[: 1 << 21], pts[: 1 << 21], E.FMT_RAW160, E.FMT_EDWARDS_Y); assert st == 0; check("msm21", bytes(out))
    st, out = eng.msm_vartime_t(xs[: (3 << 20) + 17], pts[: (3 << 20) + 17], E.FMT_RAW160, E.FMT_EDWARDS_Y); assert st == 0; check("msm_odd", bytes(out))
    st = eng.verify_batch_t(msgs.reshape(-1), off, sigs, pks, E.Z_DEVICE); assert st == 0, st
    check("x", bytes(eng.x25519_batch_t(xs[:4096], msgs[:4096]).cpu().numpy().tobytes()))
    check("fb", bytes(eng.mul_base_batch_t(xs[:4096]).cpu().numpy().tobytes()))
    # (r4) the small path (tables + lane-per-(window, term), small.hip), the digit-matrix range and small verify_batch calls
    for m in (1, 3, 100, 1024, 2049, 4095, 4096, 30000):
        st, out = eng.msm_vartime_t(xs[:m], pts[:m], E.FMT_RAW160, E.FMT_EDWARDS_Y); assert st == 0; check("msm_small_%d" % m, bytes(out))
    for m in (4, 256, 2047):
        st = eng.verify_batch_t(msgs.reshape(-1)[:32 * m], off[:m + 1], sigs[:m], pks[:m], E.Z_DEVICE); assert st == 0, st
torch.cuda.synchronize()
print("soak ok: %d repetitions, %.1f s" % (reps, time.time() - t0))
