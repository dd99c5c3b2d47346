#!/usr/bin/env python3
"""Soak of the SMALL paths that end in a record published by the last kernel and polled by the host (small.hip small_direct, msm.hip wait_published):
c25519_msm_vartime / _dev of 1 .. 12 287 terms and ed25519_verify_batch of 1 .. 128 signatures (both z-modes, keys as bytes and as cached points), mixed at
random, on several long-lived contexts of one process and on fresh ones, with large calls (the bucket pipeline, the general verify_batch) interleaved.
EVERY result is checked (MSM: (sum x_i y_i) B from the oracle's fixed-base multiplication; verify_batch: the known verdict of a good / a tampered batch); every call is
timed; the contexts' counters (c25519_ctx_counter: publications that blocked on the stream, lost publications, direct publications) are logged at the end.

    python tools/soak_small.py [calls=200000] [seed=1]        -> text on stdout (profiles/r06_soak_small.txt is the concatenation over boxes)

A call that takes longer than 50 ms is listed; a call that never returns is cut by the watchdog (faulthandler) with the stack and the last progress line."""
import faulthandler
import os
import random
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)
from oracle import orc       # (the checker; tools/ are test infrastructure)

L = 2**252 + 27742317777372353535851937790883648493
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
faulthandler.enable()                     # (a crash inside the library prints the Python stack of the call)
faulthandler.dump_traceback_later(600, exit=True)

NMAX = 12287
e0 = pkg.Engine(0)
rng = np.random.default_rng(seed)
x = rng.integers(0, 256, size=(NMAX, 32), dtype=np.uint8); x[:, 31] &= 0x0F
y = rng.integers(0, 256, size=(NMAX, 32), dtype=np.uint8); y[:, 31] &= 0x0F
pts = e0.mul_base_batch(y, out_fmt=2)                              # P_i = y_i B as raw 160-byte points
dx, dp = torch.from_numpy(x).cuda(), torch.from_numpy(pts).cuda()
xi = [int.from_bytes(x[i].tobytes(), "little") for i in range(NMAX)]
yi = [int.from_bytes(y[i].tobytes(), "little") for i in range(NMAX)]
prefix = [0]
for a, b in zip(xi, yi):
    prefix.append((prefix[-1] + a * b) % L)
_want = {}


def want_msm(n):
    if n not in _want:
        _want[n] = orc.ed_compress(orc.ed_mul_base(prefix[n].to_bytes(32, "little")))
    return _want[n]


NS = 128
seeds = [bytes(rnd.randrange(256) for _ in range(32)) for _ in range(NS)]
msgs = [bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 1, 31, 64, 111, 112, 200]))) for _ in range(NS)]
pks, sigs = e0.sign_batch(seeds, msgs)
P = [bytes(p) for p in pks]; S = [bytes(s) for s in sigs]
assert orc.ed25519_verify(P[3], msgs[3], S[3]) == 0
_, raw_keys, ok = e0.decompress_batch(np.frombuffer(b"".join(P), np.uint8).reshape(NS, 32)); assert ok.all()
# large calls interleaved: a bucket-pipeline MSM and a general-path verify_batch on the device
NL = 1 << 17
gl = torch.Generator(device="cuda"); gl.manual_seed(seed)
lx = torch.randint(0, 256, (NL, 32), dtype=torch.uint8, device="cuda", generator=gl); lx[:, 31] &= 0x0F
lraw = e0.mul_base_batch_vartime_t(lx, out_fmt=2)
st, large_want = e0.msm_vartime_t(lx, lraw, in_fmt=2, out_fmt=0); assert st == 0
NV = 1 << 12
vseeds = torch.randint(0, 256, (NV, 32), dtype=torch.uint8, device="cuda", generator=gl)
vm = torch.randint(0, 256, (NV * 16,), dtype=torch.uint8, device="cuda", generator=gl); voff = torch.arange(0, NV * 16 + 1, 16, dtype=torch.int64, device="cuda")
vpk, vsg = e0.sign_batch_t(vseeds, vm, voff)

# (r6, late) the MID path publishes its record the same way (mid.hip / reduce.hip k_reduce_b4pub): MSMs of 12 288 .. 2^17 terms on the device (a prefix of the
# large call's inputs: sum lx_i^2 B) and device-z-mode verify_batch of 6144 .. 20 000 signatures (key bytes / cached points; a copy with R_0 tampered)
lxi = [int.from_bytes(b.tobytes(), "little") for b in lx.cpu().numpy()]
sq = [0]
for a in lxi:
    sq.append((sq[-1] + a * a) % L)
_want_mid = {}


def want_mid(n):
    if n not in _want_mid:
        _want_mid[n] = orc.ed_compress(orc.ed_mul_base(sq[n].to_bytes(32, "little")))
    return _want_mid[n]


NM = 20000
mseeds = torch.randint(0, 256, (NM, 32), dtype=torch.uint8, device="cuda", generator=gl)
mm = torch.randint(0, 256, (NM * 24,), dtype=torch.uint8, device="cuda", generator=gl); moff = torch.arange(0, NM * 24 + 1, 24, dtype=torch.int64, device="cuda")
mpk, msg_ = e0.sign_batch_t(mseeds, mm, moff)
_, mpts, mok = e0.decompress_batch_t(mpk); assert bool(mok.all())
msg_bad = msg_.clone(); msg_bad[0, 3] ^= 1
assert e0.verify_batch_t(mm[:24 * 7000], moff[:7001], msg_[:7000], mpk[:7000], 1) == 0

ctxs = [e0, pkg.Engine(0), pkg.Engine(0)]
lat = {}
slow = []
wrong = 0
t_start = time.perf_counter()


def log_n():
    return min(NMAX, max(1, int(2 ** rnd.uniform(0, 13.6))))


def note(kind, n, dt):
    lat.setdefault(kind, []).append(dt)
    if dt > 50e-3:
        slow.append((kind, n, dt))


fresh_every, large_every = 4000, 1500
closed, nfresh = [0, 0, 0], 0
done = 0
while done < calls:
    r = rnd.random()
    if done and done % fresh_every == 0:                             # a fresh context replaces a long-lived one (its first call allocates everything)
        i = rnd.randrange(1, len(ctxs))
        for w in range(3):
            closed[w] += ctxs[i].counter(w)
        ctxs[i].close(); ctxs[i] = pkg.Engine(0); nfresh += 1
    eng = rnd.choice(ctxs)                                           # (after the replacement: a closed Engine has no context)
    if done and done % large_every == 0:
        t0 = time.perf_counter(); st, got = eng.msm_vartime_t(lx, lraw, in_fmt=2, out_fmt=0); note("large msm 2^17 (device)", NL, time.perf_counter() - t0)
        wrong += (st != 0 or got != large_want)
        t0 = time.perf_counter(); st = eng.verify_batch_t(vm, voff, vsg, vpk, rnd.randrange(2)); note("large verify_batch 2^12 (device)", NV, time.perf_counter() - t0)
        wrong += (st != 0)
    if r < 0.04:
        n = rnd.randrange(12288, NL + 1)
        t0 = time.perf_counter(); st, got = eng.msm_vartime_t(lx[:n], lraw[:n], in_fmt=2, out_fmt=0); note("mid msm 12288 .. 2^17 (device)", n, time.perf_counter() - t0)
        wrong += (st != 0 or got != want_mid(n))
    elif r < 0.08:
        n = rnd.randrange(6144, NM + 1)
        cached = rnd.random() < 0.5
        tamper = rnd.random() < 0.2
        t0 = time.perf_counter()
        st = eng.verify_batch_t(mm[:24 * n], moff[:n + 1], (msg_bad if tamper else msg_)[:n], mpk[:n], 1, pk_points=mpts[:n] if cached else None)
        note("mid verify_batch 6144 .. 20000, device z%s" % (" cached points" if cached else ""), n, time.perf_counter() - t0)
        wrong += (st != (3 if tamper else 0))
    elif r < 0.30:
        n = log_n()
        t0 = time.perf_counter(); st, got = eng.msm_vartime(x[:n], pts[:n], in_fmt=2, out_fmt=0); note("msm host pointers", n, time.perf_counter() - t0)
        wrong += (st != 0 or got != want_msm(n))
    elif r < 0.55:
        n = log_n()
        t0 = time.perf_counter(); st, got = eng.msm_vartime_t(dx[:n], dp[:n], in_fmt=2, out_fmt=0); note("msm device pointers", n, time.perf_counter() - t0)
        wrong += (st != 0 or got != want_msm(n))
    else:
        n = min(NS, max(1, int(2 ** rnd.uniform(0, 7.05))))
        lo = rnd.randrange(0, NS - n + 1)
        zm = rnd.randrange(2)
        cached = rnd.random() < 0.3
        tamper = rnd.random() < 0.2
        s = S[lo:lo + n]
        if tamper:
            s = list(s); b = bytearray(s[-1]); b[5] ^= 1; s[-1] = bytes(b)
        t0 = time.perf_counter()
        st = eng.verify_batch(msgs[lo:lo + n], s, P[lo:lo + n], zm, pk_points=raw_keys[lo:lo + n] if cached else None)
        note("verify_batch z=%d%s" % (zm, " cached points" if cached else ""), n, time.perf_counter() - t0)
        wrong += (st != (3 if tamper else 0))
    done += 1
    if done % 20000 == 0:
        print("# %d calls, %.1f s, wrong %d" % (done, time.perf_counter() - t_start, wrong), flush=True)
        faulthandler.cancel_dump_traceback_later(); faulthandler.dump_traceback_later(600, exit=True)

total_s = time.perf_counter() - t_start
dev = torch.cuda.get_device_name(0)
print("soak_small: %d calls in %.1f s on %s, seed %d, wrong results %d, calls beyond 50 ms %d" % (calls, total_s, dev, seed, wrong, len(slow)))
print("%-40s %8s %9s %9s %9s %9s" % ("call (wall-clock through the Python Engine)", "calls", "median us", "p99 us", "p99.9 us", "max us"))
for kind in sorted(lat):
    v = sorted(lat[kind]); k = len(v)
    print("%-40s %8d %9.1f %9.1f %9.1f %9.1f" % (kind, k, v[k // 2] * 1e6, v[min(k - 1, int(k * 0.99))] * 1e6, v[min(k - 1, int(k * 0.999))] * 1e6, v[-1] * 1e6))
for kind, n, dt in slow[:20]:
    print("slow: %s n=%d %.1f ms" % (kind, n, dt * 1e3))
tot = list(closed)
for c in ctxs:
    for w in range(3):
        tot[w] += c.counter(w)
print("counters of the %d contexts (3 long-lived at a time, %d fresh ones over the run): directly published calls %d, publications that outlasted the 2 ms spin (host blocked on the stream) %d, LOST publications %d"
      % (len(ctxs) + nfresh, nfresh, tot[2], tot[0], tot[1]))
sys.exit(1 if wrong else 0)
