"""The device code built with the limb-bound asserts (make debug: -DC25519_CHECK_BOUNDS, the counterpart of the reference's
debug_assert!s on limb magnitudes, u64/field.rs:162-166): every bound class annotated in fe26.h is CHECKED on the GPU while
the real workloads run -- a violated bound traps the kernel and the call fails.  Each scenario runs in its own process
(the library is a per-process singleton, and a trap poisons the HIP context)."""
import os
import subprocess
import sys

import pytest
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "curve25519-dalek_amd", "lib", "libc25519hip_dbg.so")

WORKLOADS = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import curve25519_dalek_amd as pkg
import util
E = pkg.engine
h = hashlib.sha256()
e = pkg.Engine(0)
ev = pkg.Engine(0, flags=E.FLAG_VARTIME_TABLES)
s = np.concatenate([util.edge_scalars(), util.rand_scalars(1, 5000)])
h.update(e.mul_base_batch(s).tobytes()); h.update(ev.mul_base_batch(s).tobytes())
pts = ev.mul_base_batch(s, out_fmt=2)
for n in (1, 100, 5000):
    st, r = e.msm_vartime(s[:n], pts[:n]); assert st == 0; h.update(r)
    st, r = e.msm_vartime(s[:n], ev.compress_batch(pts[:n]), in_fmt=0); assert st == 0; h.update(r)
k = util.rand_bytes(2, 2000); u = util.rand_bytes(3, 2000); u[:8] = 0xFF
h.update(e.x25519_batch(k, u).tobytes()); h.update(e.x25519_base_batch(k).tobytes())
out, ok = e.mul_batch(s[:500], pts[:500]); h.update(out.tobytes())
st, r = e.msm_consttime(s[:300], pts[:300]); h.update(r)
seeds = [bytes(x) for x in util.rand_bytes(4, 700)]; msgs = [bytes(x)[: (7 * i) %% 50] for i, x in enumerate(util.rand_bytes(5, 700, 64))]
pks, sigs = e.sign_batch(seeds, msgs); h.update(pks.tobytes()); h.update(sigs.tobytes())
P = [bytes(x) for x in pks]; S = [bytes(x) for x in sigs]
for z in (0, 1):
    assert e.verify_batch(msgs, S, P, z) == 0
assert not e.verify_each(msgs, S, P, True).any()
hp = e.precomp_create(pts[:300]); st, r = e.precomp_msm_vartime(hp, s[:300], s[300:320], pts[300:320]); h.update(r)
# the reference's overflow hunt (edwards.rs:2254-2261: 1000 chained scalar multiplications under debug assertions): the raw
# projective output of one variable-base multiplication is the input of the next, in both table modes, 64 chains wide
cur = pts[:64].copy(); cur_v = cur.copy(); prod = [1] * 64
for it in range(200):
    sc = util.rand_scalars(1000 + it, 64)
    cur, ok = e.mul_batch(sc, cur, out_fmt=2); assert ok.all()
    cur_v, ok = ev.mul_batch(sc, cur_v, out_fmt=2); assert ok.all()
    prod = [(p * int.from_bytes(x.tobytes(), "little")) %% util.L for p, x in zip(prod, sc)]
enc = e.compress_batch(cur); assert np.array_equal(enc, ev.compress_batch(cur_v))
want, ok = e.mul_batch(np.frombuffer(b"".join(p.to_bytes(32, "little") for p in prod), np.uint8).reshape(64, 32), pts[:64]); assert np.array_equal(enc, want)
h.update(enc.tobytes())
# a multi-pass MSM with small passes (records prepared ahead, 64 points per inversion) and affine + projective inputs mixed
big = util.rand_scalars(77, 300000); bp = ev.mul_base_batch(big, out_fmt=2); bp[::5] = e.decompress_batch(e.compress_batch(bp[::5]))[1]
st, r = e.msm_vartime(big, bp); assert st == 0; h.update(r)
rng = np.random.default_rng(9)
a = rng.integers(0, 1 << 26, size=(4096, 10), dtype=np.uint64).astype(np.uint32); a[:, 1::2] >>= 1
for chain in (0, 1):
    for op in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
        h.update(e.selftest_field(op, a, a[::-1].copy(), chain).tobytes())
print("DIGEST", h.hexdigest())
'''

VIOLATION = r'''
import sys
import numpy as np
sys.path.insert(0, %(root)r)
import curve25519_dalek_amd as pkg
e = pkg.Engine(0)
a = np.full((64, 10), 1 << 25, dtype=np.uint32)
b = a.copy()
b[:, 0] = 1 << 30                       # a "loose" operand far beyond its bound (1.52 * 2^27)
try:
    e.selftest_field(0, a, b, %(chain)d)
    print("NO-TRAP")
except pkg.EngineError as ex:
    print("TRAPPED", ex)
'''


def _run(code, lib):
    env = dict(os.environ)
    env["C25519_MSM_PASS_LOG2"] = "16"                                       # many small passes
    if lib:
        env["C25519_HIP_LIB"] = lib
    else:
        env.pop("C25519_HIP_LIB", None)
    return subprocess.run(util.child_argv(code), capture_output=True, text=True, timeout=900, env=env)


def test_workloads_run_clean_with_device_bound_checks():
    assert os.path.exists(DBG), "run __graft_entry__.build() (make debug)"
    code = WORKLOADS % {"root": ROOT}
    dbg = _run(code, DBG)
    assert dbg.returncode == 0 and "DIGEST" in dbg.stdout, (dbg.stdout[-2000:], dbg.stderr[-2000:])
    ref = _run(code, None)
    assert ref.returncode == 0, ref.stderr[-2000:]
    assert dbg.stdout.split("DIGEST")[1].strip() == ref.stdout.split("DIGEST")[1].strip()      # and the results are the release build's


@pytest.mark.parametrize("chain", [0, 1])
def test_a_violated_bound_traps_on_the_device(chain):
    """negative control: the asserts are really compiled into the device code"""
    assert os.path.exists(DBG)
    r = _run(VIOLATION % {"root": ROOT, "chain": chain}, DBG)
    assert "TRAPPED" in r.stdout or r.returncode != 0, (r.stdout, r.stderr[-1000:])
    assert "NO-TRAP" not in r.stdout
    ok = _run(VIOLATION % {"root": ROOT, "chain": chain}, None)        # the release build computes garbage silently, as documented
    assert "NO-TRAP" in ok.stdout
