#!/usr/bin/env python3
"""VartimePrecomputedStraus on the GPU: precomputed per-window tables (c25519_precomp_msm_vartime) against the plain
c25519_msm_vartime on the same static points, host-pointer entry points (scalars cross PCIe in both; the plain call also
ships and prepares the points).    python tools/precomp_numbers.py > profiles/rNN_precomp.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)

e = pkg.Engine(0, flags=pkg.engine.FLAG_VARTIME_TABLES)
rng = np.random.default_rng(1)
print("%10s %14s %14s %14s %10s %12s" % ("n_static", "create ms", "precomp ms", "plain msm ms", "speed-up", "table MB"))
for n in (128, 256, 500, 720, 1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20):
    lg = n.bit_length() - 1
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    y = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); y[:, 31] &= 0x0F
    pts = e.mul_base_batch(y, out_fmt=2)
    t0 = time.perf_counter(); h = e.precomp_create(pts, in_fmt=2); t_create = time.perf_counter() - t0
    e32, e160 = np.zeros((0, 32), np.uint8), np.zeros((0, 160), np.uint8)
    st, a = e.precomp_msm_vartime(h, x, e32, e160, in_fmt=2)
    st, b = e.msm_vartime(x, pts, in_fmt=2)
    assert a == b
    reps = 20 if lg <= 16 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        e.precomp_msm_vartime(h, x, e32, e160, in_fmt=2)
    tp = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        e.msm_vartime(x, pts, in_fmt=2)
    tm = (time.perf_counter() - t0) / reps
    import ctypes as C
    c = 5
    lgv = (n * 17).bit_length() - 1
    c = min(16, max(5, lgv - 4)); K = -(-257 // c)
    print("%10d %14.2f %14.3f %14.3f %9.2fx %12.1f" % (n, t_create * 1e3, tp * 1e3, tm * 1e3, tm / tp, n * K * 128 / 1e6))
    e.precomp_destroy(h)
