#!/usr/bin/env python3
"""Where a mid-size ed25519_verify_batch call (device z-mode, inputs and cached key points on the device) spends its time: the library's own host clock
(c25519_last_call_host_us: every kernel enqueued, record on the host, verdict) beside the GPU span (HIP events) and the call as Python sees it.
    python tools/verify_call_phases.py   (sizes: VERIFY_SIZES=8192,16384,...)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)
e = pkg.Engine(0)
for _ in range(40): e.microbench(0, 4000)
med = lambda v: sorted(v)[len(v) // 2]
print("%8s | %9s %9s %9s | %9s | %9s   (microseconds, medians of 100 calls)" % ("n", "enqueued", "on host", "returns", "GPU span", "Engine"))
for n in [int(v) for v in os.environ.get("VERIFY_SIZES", "6144,8192,16384,32768,65536,131072").split(",")]:
    seeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
    dm = torch.randint(0, 256, (59 * n,), dtype=torch.uint8, device="cuda"); doff = torch.arange(0, 59 * (n + 1), 59, dtype=torch.int64, device="cuda")
    dpk, dsg = e.sign_batch_t(seeds, dm, doff)
    _, dpts, ok = e.decompress_batch_t(dpk)
    for _ in range(5): assert e.verify_batch_t(dm, doff, dsg, dpk, 1, pk_points=dpts) == 0
    ph, gpu, py = [], [], []
    for _ in range(100):
        torch.cuda.synchronize(); t0 = time.perf_counter(); e.verify_batch_t(dm, doff, dsg, dpk, 1, pk_points=dpts); py.append((time.perf_counter() - t0) * 1e6)
        ph.append(e.last_call_host_us()); gpu.append(e.last_kernel_ms() * 1e3)
    print("%8d | %9.1f %9.1f %9.1f | %9.1f | %9.1f" % (n, med([q[1] for q in ph]), med([q[2] for q in ph]), med([q[3] for q in ph]), med(gpu), med(py)))
