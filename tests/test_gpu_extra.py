"""GPU parity tests for the callers / batch conversions either side of the MSM:
VartimePrecomputedStraus (precomputed_straus.rs), MultiscalarMul::multiscalar_mul (straus.rs:103-144),
RistrettoPoint::double_and_compress_batch (ristretto.rs:564-648), Scalar::invert_batch_alloc
(scalar.rs:802-856) -- each against the oracle / big-int arithmetic, mirroring the reference's own
consistency tests (edwards.rs:2364-2411, ristretto.rs:1497-1546, scalar.rs:1996-2013)."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
L = util.L


@pytest.fixture(scope="module")
def eng():
    import curve25519_dalek_amd as pkg
    return pkg.Engine(0)


def rows(a):
    return [a[i].tobytes() for i in range(a.shape[0])]


def i2b(x):
    return int(x).to_bytes(32, "little")


def test_precomputed_vs_nonprecomputed(eng, orc):
    """edwards.rs:2364-2411: precomputed (static + dynamic) == plain vartime MSM == oracle"""
    ns, nd = 300, 500
    ss, ds = util.rand_scalars(91, ns), util.rand_scalars(92, nd)
    sp = eng.mul_base_batch(util.rand_scalars(93, ns), out_fmt=2)
    dp = eng.mul_base_batch(util.rand_scalars(94, nd), out_fmt=2)
    h = eng.precomp_create(sp, in_fmt=2)
    assert eng.precomp_len(h) == ns
    want = orc.ed_compress(orc.ed_msm(rows(ss) + rows(ds), rows(sp) + rows(dp)))
    st, got = eng.precomp_msm_vartime(h, ss, ds, dp, in_fmt=2)
    assert st == 0 and got == want
    st, plain = eng.msm_vartime(np.concatenate([ss, ds]), np.concatenate([sp, dp]), in_fmt=2)
    assert plain == want
    # fewer static scalars than static points uses the first ones (precomputed_straus.rs:86)
    st, got = eng.precomp_msm_vartime(h, ss[:100], ds[:7], dp[:7], in_fmt=2)
    assert st == 0 and got == orc.ed_compress(orc.ed_msm(rows(ss[:100]) + rows(ds[:7]), rows(sp[:100]) + rows(dp[:7])))
    # empty dynamic part / empty static part / both empty (ristretto.rs:1599-1740)
    e32, e160 = np.zeros((0, 32), np.uint8), np.zeros((0, 160), np.uint8)
    st, got = eng.precomp_msm_vartime(h, ss, e32, e160, in_fmt=2)
    assert st == 0 and got == orc.ed_compress(orc.ed_msm(rows(ss), rows(sp)))
    st, got = eng.precomp_msm_vartime(h, e32, ds, dp, in_fmt=2)
    assert st == 0 and got == orc.ed_compress(orc.ed_msm(rows(ds), rows(dp)))
    st, got = eng.precomp_msm_vartime(h, e32, e32, e160, in_fmt=2)
    assert st == 0 and got == i2b(1)
    # compressed dynamic points, one of them invalid -> None
    enc = eng.compress_batch(dp[:50]); bad = enc.copy(); bad[3] = np.frombuffer(i2b(2), np.uint8)
    st, got = eng.precomp_msm_vartime(h, ss[:10], ds[:50], enc, in_fmt=0)
    assert st == 0 and got == orc.ed_compress(orc.ed_msm(rows(ss[:10]) + rows(ds[:50]), rows(sp[:10]) + rows(dp[:50])))
    st, _ = eng.precomp_msm_vartime(h, ss[:10], ds[:50], bad, in_fmt=0)
    assert st == 1
    import curve25519_dalek_amd as pkg
    with pytest.raises(pkg.EngineError):
        eng.precomp_msm_vartime(h, np.concatenate([ss, ss]), e32, e160, in_fmt=2)   # more static scalars than points
    eng.precomp_destroy(h)


@pytest.mark.parametrize("ns", [1, 2, 17, 130, 500, 720, 1000, 4096, 70000])      # (130 / 500 / 720: 17 ns falls into the SMALL path's term range -- the merged layout must keep its own widths)
def test_precomputed_tables_all_window_layouts(eng, orc, ns):
    """precomputed_straus.rs:57-127 through the merged-window tables (2^(c k) P_i for every window k): the table layout
    changes with the number of static points (c = 6 .. 16), every layout must give the oracle's result -- edge scalars
    (0, 1, l - 1, unreduced up to 2^255 - 1: every digit pattern incl. the carry into the top window) and random ones,
    Edwards and Ristretto encodings of the static points."""
    e = util.edge_scalars()
    ss = np.concatenate([e, util.rand_scalars(700 + ns, max(0, ns - e.shape[0]))])[:ns]
    sp = eng.mul_base_batch(util.rand_scalars(701 + ns, ns), out_fmt=2)
    total = orc.ed_msm(rows(ss), rows(sp))
    e32, e160 = np.zeros((0, 32), np.uint8), np.zeros((0, 160), np.uint8)
    for fmt, enc in ((2, sp), (0, eng.compress_batch(sp)), (1, eng.compress_batch(sp, out_fmt=1))):
        h = eng.precomp_create(enc, in_fmt=fmt)
        # a CompressedRistretto decodes to SOME representative of the coset: compare as Ristretto points (ristretto.rs:822-829)
        out_fmt, want = (1, orc.ris_compress(total)) if fmt == 1 else (0, orc.ed_compress(total))
        st, got = eng.precomp_msm_vartime(h, ss, e32, e160 if fmt == 2 else e32, in_fmt=fmt, out_fmt=out_fmt)
        assert st == 0 and got == want, (ns, fmt)
        if fmt == 2 and ns >= 17:
            st, got = eng.precomp_msm_vartime(h, ss[:ns // 2], e32, e160, in_fmt=2)          # a prefix of the static points
            assert st == 0 and got == orc.ed_compress(orc.ed_msm(rows(ss[:ns // 2]), rows(sp[:ns // 2])))
        eng.precomp_destroy(h)
    import curve25519_dalek_amd as pkg
    bad = eng.compress_batch(sp); bad[ns // 2] = np.frombuffer(i2b(2), np.uint8)
    with pytest.raises(pkg.EngineError):
        eng.precomp_create(bad, in_fmt=0)                                                        # a static point that does not decode


@pytest.mark.parametrize("n", [0, 1, 5, 64, 65, 300, 5000])
def test_msm_consttime_vs_vartime(eng, orc, n):
    """edwards.rs:2276-2335: constant-time and variable-time multiscalar agree (and equal (sum x^2) B)"""
    x = util.rand_scalars(110 + n, n)
    raw = eng.mul_base_batch(x, out_fmt=2) if n else np.zeros((0, 160), np.uint8)
    want = orc.ed_compress(orc.ed_mul_base(i2b(sum(int.from_bytes(r.tobytes(), "little") ** 2 for r in x) % L)))
    st, got = eng.msm_consttime(x, raw, in_fmt=2)
    assert st == 0 and got == want
    st, vt = eng.msm_vartime(x, raw, in_fmt=2)
    assert vt == got
    if n:
        enc = eng.compress_batch(raw)
        st, got2 = eng.msm_consttime(x, enc, in_fmt=0, out_fmt=2)
        assert st == 0 and orc.ed_compress(got2) == want


def test_double_and_compress_batch(eng, orc):
    """ristretto.rs:1497-1512: batch == compress(P + P) one at a time; the identity and 4-torsion included"""
    n = 3000
    s = util.rand_scalars(120, n)
    s[0] = 0                                           # identity: e*g*f*h = 0 -> the zero-skipping branch
    raw = eng.mul_base_batch(s, out_fmt=2)
    t4 = orc.ed_decompress(i2b(0))                     # order-4 point (y = 0)
    raw[1] = np.frombuffer(t4, np.uint8)
    raw[2] = np.frombuffer(orc.ed_add(raw[5].tobytes(), t4), np.uint8)
    got = eng.double_and_compress_batch(raw)
    for i in list(range(40)) + list(range(40, n, 53)):
        want = orc.ris_compress(orc.ed_double(raw[i].tobytes()))
        assert got[i].tobytes() == want, i
    assert not got[0].any() and not got[1].any()       # 2*identity and 2*(4-torsion) encode as the identity
    assert eng.double_and_compress_batch(np.zeros((0, 160), np.uint8)).shape == (0, 32)


def test_scalar_invert_batch(eng):
    """scalar.rs:1996-2013 batch_invert_consistency / :1688 invert"""
    n = 5000
    s = util.rand_scalars(130, n)
    vals = [int.from_bytes(r.tobytes(), "little") % L or 1 for r in s]
    vals[:4] = [1, 2, L - 1, L - 2]
    arr = np.frombuffer(b"".join(i2b(v) for v in vals), np.uint8).reshape(-1, 32)
    inv, prod = eng.scalar_invert_batch(arr)
    want_prod = 1
    for i, v in enumerate(vals):
        w = pow(v, -1, L)
        want_prod = want_prod * w % L
        if i < 64 or i % 37 == 0:
            assert int.from_bytes(inv[i].tobytes(), "little") == w, i
    assert int.from_bytes(prod, "little") == want_prod
    e, p = eng.scalar_invert_batch(np.zeros((0, 32), np.uint8))
    assert e.shape == (0, 32) and int.from_bytes(p, "little") == 1
