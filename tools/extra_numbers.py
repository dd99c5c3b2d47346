#!/usr/bin/env python3
"""Side numbers quoted in DESIGN.md / README.md that bench.py does not print (run on the GPU box): the per-item paths
(variable base, double base, per-signature verify, keygen / sign), (de)compression, with the constant-time default and
with the fast tables (C25519_FLAG_VARTIME_TABLES).    python tools/extra_numbers.py > profiles/rNN_extra_numbers.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)

t0 = time.perf_counter(); e = pkg.Engine(0); t1 = time.perf_counter()
print("ctx_create, first in the process (both table sets): %.1f ms" % ((t1 - t0) * 1e3))
t0 = time.perf_counter(); ev = pkg.Engine(0, flags=pkg.engine.FLAG_VARTIME_TABLES); t1 = time.perf_counter()
print("ctx_create, again:                                   %.1f ms" % ((t1 - t0) * 1e3))
g = torch.Generator(device="cuda"); g.manual_seed(7)


def rnd(n, w=32):
    return torch.randint(0, 256, (n, w), dtype=torch.uint8, device="cuda", generator=g)


def best(eng, f, reps=5):
    """best GPU time of the WHOLE call (torch events on the stream the engine launches on; c25519_last_kernel_ms brackets only
    the last fixed-base launch of a multi-kernel call -- round 2's and the first round-3 sign_batch figure were that)"""
    for _ in range(40):
        eng.microbench(0, 4000)             # sustained clock first (see bench.py)
    f(); b = 1e9
    for _ in range(reps):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); f(); t1.record(); t1.synchronize()
        b = min(b, t0.elapsed_time(t1))
    return b


n = 1 << 20
s = rnd(n); s[:, 31] &= 0x0F
raw = ev.mul_base_batch_t(s, out_fmt=2); enc = ev.mul_base_batch_t(s)
print("%-52s %10s %10s" % ("kernel time, ms", "const-time", "vartime"))
def row(name, f):
    print("%-52s %10.3f %10.3f" % (name, best(e, lambda: f(e)), best(ev, lambda: f(ev))))
row("mul_base 2^20 -> raw160", lambda x: x.mul_base_batch_t(s, out_fmt=2))
row("mul_base 2^20 -> compressed", lambda x: x.mul_base_batch_t(s))
row("x25519 public keys 2^20 (fixed-base path)", lambda x: x.x25519_base_batch_t(s))
row("compress_batch 2^20", lambda x: x.compress_batch_t(raw))
row("decompress_batch 2^20", lambda x: x.decompress_batch_t(enc))
row("mul_batch (variable base) 2^16", lambda x: x.mul_batch_t(s[:65536], raw[:65536]))
row("double_base_batch 2^16 (vartime by contract)", lambda x: x.double_base_batch_t(s[:65536], raw[:65536], s[65536:131072]))
m = 1 << 16
seeds = rnd(m); msgs = rnd(m, 59).reshape(-1); off = torch.arange(0, 59 * (m + 1), 59, dtype=torch.int64, device="cuda")
pk, sg = e.sign_batch_t(seeds, msgs, off)
row("keygen_batch 2^16", lambda x: x.keygen_batch_t(seeds))
row("sign_batch 2^16", lambda x: x.sign_batch_t(seeds, msgs, off))
row("verify_each 2^16 (public scalars in both)", lambda x: x.verify_each_t(msgs, off, sg, pk, False))
row("verify_each strict 2^16", lambda x: x.verify_each_t(msgs, off, sg, pk, True))
