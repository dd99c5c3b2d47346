#!/usr/bin/env python3
"""Where a SMALL synchronous call spends its time (round 4 verdict, item 8): per-phase host clock of c25519_msm_vartime / ed25519_verify_batch through the
host-pointer entry points -- upload enqueued, kernels enqueued, results on the host, folded -- from the library's own timestamps
(c25519_last_call_host_us), the GPU time of the call (HIP events), the call as ctypes sees it and the Python wrapper around that.
    python tools/small_call_phases.py > profiles/rNN_small_call_phases.txt"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)
e = pkg.Engine(0); E = pkg.engine
lib = e.lib
rng = np.random.default_rng(1)
for _ in range(40): e.microbench(0, 4000)
med = lambda v: sorted(v)[len(v) // 2]
print("c25519_msm_vartime, raw points, host pointers; medians of 300 calls, microseconds")
print("%6s | %9s %9s %9s %9s | %9s | %9s %9s" % ("n", "upload", "enqueued", "on host", "folded", "GPU span", "ctypes", "Engine"))
for n in (1, 4, 16, 64, 256, 1024, 4000):
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    dx = torch.from_numpy(x).cuda(); pts = e.mul_base_batch_vartime_t(dx, E.FMT_RAW160).cpu().numpy()
    out = np.empty(32, np.uint8)
    for _ in range(10): e.msm_vartime(x, pts)
    ph, gpu, ct, py = [], [], [], []
    for _ in range(300):
        t0 = time.perf_counter(); st = lib.c25519_msm_vartime(e.ctx, x.ctypes.data, pts.ctypes.data, n, E.FMT_RAW160, E.FMT_EDWARDS_Y, out.ctypes.data); t1 = time.perf_counter()
        assert st == 0
        ct.append((t1 - t0) * 1e6); ph.append(e.last_call_host_us()); gpu.append(e.last_kernel_ms() * 1e3)
        t0 = time.perf_counter(); e.msm_vartime(x, pts); py.append((time.perf_counter() - t0) * 1e6)
    p = [med([q[i] for q in ph]) for i in range(4)]
    print("%6d | %9.1f %9.1f %9.1f %9.1f | %9.1f | %9.1f %9.1f" % (n, p[0], p[1], p[2], p[3], med(gpu), med(ct), med(py)))
print("columns: host clock since the call was entered (cumulative) -- inputs staged + upload enqueued; every kernel enqueued; results on the host (published by the last kernel into")
print("page-locked memory, polled); folded and encoded = the call returns.  GPU span: HIP events around everything the call put on its stream.  ctypes / Engine: the same call timed")
print("from Python around the bare ctypes call and around Engine.msm_vartime (numpy checks, output allocation).")
print()
print("ed25519_verify_batch (keys as bytes), host pointers; medians of 200 calls, microseconds")
print("%6s %7s | %9s %9s %9s %9s | %9s | %9s" % ("n", "z_mode", "upload", "enqueued", "on host", "returns", "GPU span", "Engine"))
for n in [int(x) for x in os.environ.get("PHASES_VERIFY_SIZES", "4,16,64,256,1024").split(",")]:
    seeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
    dm = torch.randint(0, 256, (59 * n,), dtype=torch.uint8, device="cuda"); doff = torch.arange(0, 59 * (n + 1), 59, dtype=torch.int64, device="cuda")
    dpk, dsg = e.sign_batch_t(seeds, dm, doff)
    M = dm.cpu().numpy(); P = dpk.cpu().numpy(); S = dsg.cpu().numpy()
    msgs = [M[59 * i:59 * i + 59].tobytes() for i in range(n)]; sigs = [S[i].tobytes() for i in range(n)]; pks = [P[i].tobytes() for i in range(n)]
    for zm in (0, 1):
        for _ in range(5): assert e.verify_batch(msgs, sigs, pks, zm) == 0
        ph, gpu, py = [], [], []
        for _ in range(200):
            t0 = time.perf_counter(); e.verify_batch(msgs, sigs, pks, zm); py.append((time.perf_counter() - t0) * 1e6)
            ph.append(e.last_call_host_us()); gpu.append(e.last_kernel_ms() * 1e3)
        print("%6d %7d | %9.1f %9.1f %9.1f %9.1f | %9.1f | %9.1f" % (n, zm, med([q[0] for q in ph]), med([q[1] for q in ph]), med([q[2] for q in ph]), med([q[3] for q in ph]), med(gpu), med(py)))
