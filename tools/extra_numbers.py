#!/usr/bin/env python3
"""Side numbers quoted in DESIGN.md / README.md that bench.py does not print (run on the GPU box):
context creation time, per-signature verify, keygen/sign, the 2^24-term single-call MSM, (de)compression."""
import sys
import time

import torch

import curve25519_dalek_amd as pkg

t0 = time.perf_counter(); e = pkg.Engine(0); t1 = time.perf_counter()
print("ctx_create (default, radix-2^16 tables): %.1f ms" % ((t1 - t0) * 1e3))
t0 = time.perf_counter(); e9 = pkg.Engine(0, window=9); t1 = time.perf_counter(); e9.close()
print("ctx_create (flags=9, LDS comb):          %.1f ms" % ((t1 - t0) * 1e3))
g = torch.Generator(device="cuda"); g.manual_seed(7)


def rnd(n, w=32):
    return torch.randint(0, 256, (n, w), dtype=torch.uint8, device="cuda", generator=g)


def best(f, reps=5):
    for _ in range(40):
        e.microbench(0, 4000)             # sustained clock first (see bench.py)
    f(); b = 1e9
    for _ in range(reps):
        f(); b = min(b, e.last_kernel_ms())
    return b


n = 1 << 20
s = rnd(n); s[:, 31] &= 0x0F
raw = e.mul_base_batch_t(s, out_fmt=2); enc = e.mul_base_batch_t(s)
print("mul_base 2^20 -> raw160:        %.3f ms" % best(lambda: e.mul_base_batch_t(s, out_fmt=2)))
print("mul_base 2^20 -> compressed:    %.3f ms" % best(lambda: e.mul_base_batch_t(s)))
print("compress_batch 2^20:            %.3f ms" % best(lambda: e.compress_batch_t(raw)))
print("decompress_batch 2^20:          %.3f ms" % best(lambda: e.decompress_batch_t(enc)))
print("mul_batch (variable base) 2^16: %.3f ms" % best(lambda: e.mul_batch_t(s[:65536], raw[:65536])))
print("double_base_batch 2^16:         %.3f ms" % best(lambda: e.double_base_batch_t(s[:65536], raw[:65536], s[65536:131072])))
m = 1 << 16
seeds = rnd(m); msgs = rnd(m, 59).reshape(-1); off = torch.arange(0, 59 * (m + 1), 59, dtype=torch.int64, device="cuda")
pk, sg = e.sign_batch_t(seeds, msgs, off)
print("keygen_batch 2^16:              %.3f ms" % best(lambda: e.keygen_batch_t(seeds)))
print("sign_batch 2^16:                %.3f ms" % best(lambda: e.sign_batch_t(seeds, msgs, off)))
print("verify_each 2^16:               %.3f ms" % best(lambda: e.verify_each_t(msgs, off, sg, pk, False)))
print("verify_each strict 2^16:        %.3f ms" % best(lambda: e.verify_each_t(msgs, off, sg, pk, True)))
N = 1 << 24
x = rnd(N); x[:, 31] &= 0x0F
P = e.mul_base_batch_t(x, out_fmt=2)
print("msm 2^24 terms, one call (8 passes): %.2f ms" % best(lambda: e.msm_vartime_t(x, P, in_fmt=2, out_fmt=0), 2))
