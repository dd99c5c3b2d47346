"""The SCALAR layer AS COMPILED FOR THE GPU against Python big integers: c25519_selftest_scalar runs one csrc/sc28.h operation
per row on the device (release and, via C25519_HIP_LIB, the bound-checking debug library), on raw words / raw 28-bit limbs,
so the extremes the SHA-512 outputs of real signatures never reach -- 2^512 - 1, k l, k l +- 1, every limb at 2^28 - 1 -- go
through the device code generation directly (one v_mad_u64_u32 per partial product, 64-bit columns without carry tracking),
not only through a host compile of the same header (tests/test_fe26_host.py).  Reference: u64/scalar.rs:89-121 (from_bytes_wide),
:161-207 (add / sub), :222-320 (mul_internal / montgomery_reduce / mul), scalar.rs:248-263."""
import os
import subprocess
import sys

import numpy as np
import pytest
import util

pytestmark = pytest.mark.gpu
L = 2**252 + 27742317777372353535851937790883648493
M28 = (1 << 28) - 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    import curve25519_dalek_amd as pkg
    return pkg.Engine(0)


def words(vals, nw):
    """Python ints -> (n, 16) uint32, the first nw words little-endian, the rest zero"""
    raw = b"".join(int(v).to_bytes(4 * nw, "little") for v in vals)
    a = np.zeros((len(vals), 16), dtype=np.uint32)
    a[:, :nw] = np.frombuffer(raw, dtype="<u4").reshape(-1, nw)
    return a


def limbs(vals, nl):
    """Python ints -> (n, 16) uint32 holding nl 28-bit limbs (the last limb takes whatever is left: it may exceed 28 bits only if
    the value does not fit nl * 28 bits, which the callers avoid)"""
    a = np.zeros((len(vals), 16), dtype=np.uint32)
    for r, v in enumerate(vals):
        for i in range(nl):
            a[r, i] = (v >> (28 * i)) & M28 if i < nl - 1 else v >> (28 * i)
    return a


def limb_value(a, nl):
    acc = np.zeros(a.shape[0], dtype=object)
    for i in range(nl):
        acc = acc + (a[:, i].astype(object) << (28 * i))
    return [int(v) for v in acc]


def enc(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32)


def rand_ints(rng, n, bits):
    raw = rng.integers(0, 256, size=(n, (bits + 7) // 8), dtype=np.uint8)
    mask = (1 << bits) - 1
    return [int.from_bytes(raw[i].tobytes(), "little") & mask for i in range(n)]


def test_from_wide_2p20(eng):
    """Scalar::from_bytes_mod_order_wide on the device: 2^20 random 512-bit values + the extremes of every fold"""
    rng = np.random.default_rng(2801)
    edge = [0, 1, L - 1, L, L + 1, 2 * L, 2**252 - 1, 2**252, 2**253 - 1, 2**256 - 1, 2**256, 2**260 - 1, 2**384, 2**392, 2**393 - 1, 2**511, 2**512 - 1]
    top = (2**512 - 1) // L
    for k in (1, 2, 3, 7, 2**125, 2**128 - 1, 2**200 + 12345, top - 1, top):
        edge += [k * L - 1, k * L, k * L + 1]
    for sh in range(0, 512, 28):                      # one limb saturated, the others zero / all saturated but one
        edge += [M28 << sh & (2**512 - 1), (2**512 - 1) ^ (M28 << sh & (2**512 - 1))]
    n = 1 << 20
    vals = edge + rand_ints(rng, n - len(edge), 512)
    got = eng.selftest_scalar(0, words(vals, 16))
    want = enc([v % L for v in vals])
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "from_wide differs at rows %s (first value %x)" % (bad[:5], vals[int(bad[0])])


def test_mul_add_neg_canonical_2p18(eng):
    rng = np.random.default_rng(2802)
    n = 1 << 18
    # op 1: 5 x 10 limbs, a b < 2^393 -- z_i (128 bits, also the full 140-bit limb range) times s_i / h_i (canonical, and any value < 2^256)
    a = [0, 1, 2**128 - 1, 2**127, 2**140 - 1, 2**140 - 1, 2**128 - 1, 2**128 - 1] + rand_ints(rng, n // 2 - 8, 128) + rand_ints(rng, n // 2, 140)
    b = [0, L - 1, L - 1, 2**253 - 1, 2**253 - 1, L, 2**256 - 1, 2**252] + [v % L for v in rand_ints(rng, n // 2 - 8, 256)] + rand_ints(rng, n // 2, 253)
    al, bl = limbs(a, 5), limbs(b, 10)
    assert all(x * y < 2**393 for x, y in zip(a[:8], b[:8]))
    assert np.array_equal(eng.selftest_scalar(1, al, bl), enc([x * y % L for x, y in zip(a, b)]))
    # all-maximal limbs inside the contract: a = 2^140 - 1, b = 2^253 - 1 -> every column at its largest
    # op 2: 10 x 10 limbs with a b < 2^512: unreduced 256-bit values (signing: k a with the clamped a; invert chains)
    a = [0, 1, L - 1, 2**256 - 1, 2**256 - 1, 2**255 - 1, 2**252 - 1, (1 << 255) | M28] + rand_ints(rng, n - 8, 256)
    b = [0, L - 1, L - 1, 2**256 - 1, L, 2**255 - 19, 2**252 - 1, 2**256 - 1] + rand_ints(rng, n - 8, 256)
    assert np.array_equal(eng.selftest_scalar(2, limbs(a, 10), limbs(b, 10)), enc([x * y % L for x, y in zip(a, b)]))
    # ops 3, 4: canonical operands, the borrow chains at their ends
    a = [0, 0, L - 1, L - 1, 1, 2**252, 2**252 - 1, L - 2**252] + [v % L for v in rand_ints(rng, n - 8, 256)]
    b = [0, L - 1, L - 1, 1, L - 1, 2**252 - 1, 1, 2**252] + [v % L for v in rand_ints(rng, n - 8, 256)]
    assert np.array_equal(eng.selftest_scalar(3, words(a, 8), words(b, 8)), enc([(x + y) % L for x, y in zip(a, b)]))
    assert np.array_equal(eng.selftest_scalar(4, words(a, 8)), enc([(-x) % L for x in a]))
    # op 5: s < l word by word (from_canonical_bytes), every word boundary of l, bit 255
    c = [0, L - 1, L, L + 1, 2**252, 2**252 - 1, 2**255, 2**255 + 5, 2**256 - 1, L + 2**32, L - 2**32, L + 2**96, L - 2**96, L ^ 1, 2**253] + rand_ints(rng, 4096, 256) + rand_ints(rng, 4096, 253)
    got = eng.selftest_scalar(5, words(c, 8))
    assert [int(x) for x in got[:, 0]] == [int(v < L) for v in c]
    # op 6: words -> limbs -> words
    r = [0, 2**256 - 1, L, 2**255] + rand_ints(rng, 4096, 256)
    assert np.array_equal(eng.selftest_scalar(6, words(r, 8)), enc(r))
    # op 7: S = r + k a as ed25519_sign chains it (signing.rs:899): k from a 512-bit hash, a clamped and unreduced, r canonical
    m = 1 << 16
    h = [2**512 - 1, 0, L, 2**512 - 1] + rand_ints(rng, m - 4, 512)
    av = [2**255 - 8, 2**254, 2**255 - 8, (2**254) | (2**254 - 8)] + [(v & ~7 & (2**255 - 1)) | 2**254 for v in rand_ints(rng, m - 4, 256)]
    rv = [L - 1, 0, L - 1, 1] + [v % L for v in rand_ints(rng, m - 4, 256)]
    bw = np.zeros((m, 16), dtype=np.uint32)
    bw[:, :8] = words(av, 8)[:, :8]; bw[:, 8:] = words(rv, 8)[:, :8]
    assert np.array_equal(eng.selftest_scalar(7, words(h, 16), bw), enc([(r_ + (k % L) * a_) % L for k, a_, r_ in zip(h, av, rv)]))


def test_scalar_selftest_in_debug_library():
    """the same extremes through lib/libc25519hip_dbg.so (device limb-bound asserts on): a fresh process, because the
    library is chosen at load time"""
    dbg = os.path.join(ROOT, "curve25519-dalek_amd", "lib", "libc25519hip_dbg.so")
    if not os.path.exists(dbg):
        pytest.skip("debug library not built")
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_scalar as T
import curve25519_dalek_amd as pkg
e = pkg.Engine(0)
L = T.L
rng = np.random.default_rng(5)
v = [2**512 - 1, L, L - 1, L + 1, 0, ((2**512 - 1) // L) * L] + T.rand_ints(rng, 4090, 512)
assert np.array_equal(e.selftest_scalar(0, T.words(v, 16)), T.enc([x %% L for x in v]))
a = [2**140 - 1, 2**128 - 1] + T.rand_ints(rng, 4094, 128); b = [2**253 - 1, 2**256 - 1] + T.rand_ints(rng, 4094, 253)
assert np.array_equal(e.selftest_scalar(1, T.limbs(a, 5), T.limbs(b, 10)), T.enc([x * y %% L for x, y in zip(a, b)]))
a = [2**256 - 1] + T.rand_ints(rng, 4095, 256); b = [2**256 - 1] + T.rand_ints(rng, 4095, 256)
assert np.array_equal(e.selftest_scalar(2, T.limbs(a, 10), T.limbs(b, 10)), T.enc([x * y %% L for x, y in zip(a, b)]))
print("dbg scalar selftest ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, C25519_HIP_LIB=dbg)
    r = subprocess.run(util.child_argv(code), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "dbg scalar selftest ok" in r.stdout, r.stdout + r.stderr
