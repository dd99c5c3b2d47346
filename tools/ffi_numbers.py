#!/usr/bin/env python3
"""Host-pointer (FFI) figures: wall-clock per call of the entry points a Rust caller binds, at the BASELINE sizes, with
fresh output buffers (first-touch page faults inside the call), reused output buffers and page-locked buffers from
c25519_host_alloc.  Prints one line per measurement; `python tools/ffi_numbers.py > gpurun_out/ffi_numbers.txt`."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import curve25519_dalek_amd as pkg  # noqa: E402
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)

E = pkg.engine
LINK_GBS = 64.0          # PCIe Gen5 x16, per direction (measured: 56 - 57 GB/s, profiles/r03_pcie_probe.txt)


def line(what, ms_list, eng, units):
    ms, up, down = eng.last_ffi()
    best = min(ms_list)
    print("%-64s best %7.2f ms  (%s)  inside the call %6.2f ms  %.2e units/s  up %5.1f MB down %5.1f MB  link %4.1f GB/s = %.2f of %g" % (
        what, best, " ".join("%.1f" % m for m in ms_list), ms, units / (best * 1e-3), up / 1e6, down / 1e6,
        (up + down) / (best * 1e-3) / 1e9, (up + down) / (best * 1e-3) / 1e9 / LINK_GBS, LINK_GBS), flush=True)


def timed(fn, reps):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); out.append((time.perf_counter() - t0) * 1e3)
    return out


def pinned(lib, shape):
    n = int(np.prod(shape))
    p = lib.c25519_host_alloc(n)
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=shape), p


def main():
    eng = pkg.Engine(0)
    lib = eng.lib
    rng = np.random.default_rng(5)
    n = 1 << 20
    for _ in range(20):
        eng.microbench(0, 4000)                                   # clocks up
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); s[:, 31] &= 0x0F
    eng.mul_base_batch(s[:4096])
    line("mul_base 2^20 ct, fresh np.empty output every call", timed(lambda: eng.mul_base_batch(s), 6), eng, n)
    out = np.zeros((n, 32), np.uint8)
    line("mul_base 2^20 ct, reused (touched) output buffer", timed(lambda: eng.mul_base_batch(s, out=out), 6), eng, n)
    pi, p1 = pinned(lib, (n, 32)); po, p2 = pinned(lib, (n, 32)); pi[:] = s
    line("mul_base 2^20 ct, c25519_host_alloc input and output", timed(lambda: eng.mul_base_batch(pi, out=po), 6), eng, n)
    assert np.array_equal(po, out)
    ev = pkg.Engine(0, flags=E.FLAG_VARTIME_TABLES)
    ev.mul_base_batch(s[:4096])
    line("mul_base 2^20 vartime tables, reused output buffer", timed(lambda: ev.mul_base_batch(s, out=out), 6), ev, n)
    line("mul_base 2^20 vartime tables, c25519_host_alloc buffers", timed(lambda: ev.mul_base_batch(pi, out=po), 6), ev, n)
    ev.close()
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); u = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    eng.x25519_batch(k[:4096], u[:4096])
    line("x25519 2^20, fresh output", timed(lambda: eng.x25519_batch(k, u), 3), eng, n)
    line("x25519 2^20, reused output buffer", timed(lambda: eng.x25519_batch(k, u, out=out), 4), eng, n)
    line("x25519 public keys 2^20 (fixed base), reused output", timed(lambda: eng.x25519_base_batch(k, out=out), 4), eng, n)
    m = 1 << 21
    x = rng.integers(0, 256, size=(m, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    dpts = eng.mul_base_batch_vartime_t(torch.from_numpy(x).cuda(), E.FMT_RAW160)
    pts = dpts.cpu().numpy(); enc = eng.compress_batch_t(dpts).cpu().numpy()
    eng.msm_vartime(x[:4096], pts[:4096])
    line("msm 2^21 raw 160-byte points (402 MB up)", timed(lambda: eng.msm_vartime(x, pts), 4), eng, m)
    line("msm 2^21 compressed points (134 MB up)", timed(lambda: eng.msm_vartime(x, enc, in_fmt=0), 4), eng, m)
    line("msm 2^20 raw", timed(lambda: eng.msm_vartime(x[:n], pts[:n]), 4), eng, n)
    del dpts
    # verify_batch through host pointers (numpy arrays prepared once; the python list packing is not part of the measurement)
    dseeds = torch.from_numpy(k).cuda()
    dmsg = torch.from_numpy(u).cuda().reshape(-1)
    doff = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device="cuda")
    dpk, dsig = eng.sign_batch_t(dseeds, dmsg, doff)
    _, dpp, _ = eng.decompress_batch_t(dpk)
    hmsg = np.ascontiguousarray(u.reshape(-1)); hoff = doff.cpu().numpy().astype(np.uint64); hsig = dsig.cpu().numpy(); hpk = dpk.cpu().numpy(); hpp = dpp.cpu().numpy()

    def vb(zmode, pp):
        eng._bind_stream()
        st = lib.ed25519_verify_batch_keys(eng.ctx, hmsg.ctypes.data, hoff.ctypes.data, hsig.ctypes.data, hpk.ctypes.data, pp.ctypes.data if pp is not None else None, n, zmode)
        assert st == 0, st
    vb(1, None)
    line("verify_batch 2^20 device z-mode, keys as 32 bytes (176 MB up)", timed(lambda: vb(1, None), 4), eng, n)
    line("verify_batch 2^20 device z-mode, + the keys' points (344 MB up)", timed(lambda: vb(1, hpp), 4), eng, n)
    line("verify_batch 2^20 strict transcript z-mode, keys as bytes", timed(lambda: vb(0, None), 2), eng, n)
    for p in (p1, p2):
        lib.c25519_host_free(p)


if __name__ == "__main__":
    main()
