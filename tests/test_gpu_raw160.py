"""The raw160 wire format at its CONTRACT (include/c25519_hip.h, point format 2): the 5 x u64 radix-2^51 limbs of a raw point
may be ANY u64 values and denote sum_i l_i 2^(51 i) mod p (u64/field.rs:43-52; the reference's own operations keep limbs
below 2^52 and debug_assert! that, field.rs:125-166).  Every raw point the other tests feed the GPU comes out of the engine
(canonical limbs) or the oracle (weakly reduced limbs); here the same points are re-expressed with limbs in [2^51, 2^52) -- what
the contract of round 3 promised -- and with limbs up to 2^64 - 1, and every entry point that reads raw points must return
byte for byte what it returns for the canonical limbs: nothing is truncated, no magnitude changes a result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 2**255 - 19


@pytest.fixture(scope="module")
def eng():
    import curve25519_dalek_amd as pkg
    return pkg.Engine(0)


def relimb(raw, rng, cmax):
    """(n, 160) uint8 raw points with canonical limbs -> the same field elements with limb i = l_i + c_i 2^51 - c_(i-1)
    (c_(-1) = 19 c_4: 2^255 = 19 mod p), c_i uniform in [1, cmax]: cmax = 1 puts every limb into [2^51 - 19, 2^52), cmax = 8191
    reaches 2^64 - 1."""
    l = raw.view("<u8").reshape(-1, 4, 5).astype(np.uint64).copy()
    assert (l < (1 << 51)).all()
    c = rng.integers(1, cmax + 1, size=l.shape, dtype=np.uint64)
    out = l + (c << np.uint64(51))
    out[:, :, 1:] -= c[:, :, :-1]
    out[:, :, 0] -= np.uint64(19) * c[:, :, 4]
    return np.ascontiguousarray(out.reshape(-1, 20)).view(np.uint8).reshape(-1, 160)


def fe_values(raw):
    l = raw.view("<u8").reshape(-1, 4, 5)
    return [[sum(int(l[r, c, i]) << (51 * i) for i in range(5)) % P for c in range(4)] for r in range(l.shape[0])]


@pytest.mark.parametrize("cmax", [1, 8191])
def test_raw_points_with_unreduced_limbs(eng, cmax):
    from curve25519_dalek_amd import engine as E
    rng = np.random.default_rng(3100 + cmax)
    n = 70001                                              # ragged against every chunk / wave / block size
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); s[:, 31] &= 0x0F
    pts = eng.mul_base_batch(s, out_fmt=E.FMT_RAW160)
    big = relimb(pts, rng, cmax)
    lim = big.view("<u8")
    if cmax == 1:
        assert (lim >= (1 << 51) - 19).all() and (lim < (1 << 52)).all()
    else:
        assert (lim >= (1 << 58)).any()                    # the range round 3 truncated silently
        big[0] = relimb(pts[:1], np.random.default_rng(1), 1)          # the classes mixed within one wave
    # the re-expressed limbs denote the same elements (sanity of the test itself)
    assert fe_values(big[:64]) == fe_values(pts[:64])
    # (de)serialisation paths
    assert np.array_equal(eng.compress_batch(big), eng.compress_batch(pts))
    assert np.array_equal(eng.compress_batch(big, out_fmt=E.FMT_RISTRETTO), eng.compress_batch(pts, out_fmt=E.FMT_RISTRETTO))
    assert np.array_equal(eng.to_montgomery_batch(big), eng.to_montgomery_batch(pts))
    assert np.array_equal(eng.double_and_compress_batch(big[:4096]), eng.double_and_compress_batch(pts[:4096]))
    # variable base, double base, MSM (the normaliser k_prep_raw2 reads X, Y, Z of every point), constant-time MSM
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    m = 20000
    assert np.array_equal(eng.mul_batch(x[:m], big[:m])[0], eng.mul_batch(x[:m], pts[:m])[0])
    assert np.array_equal(eng.double_base_batch(x[:m], big[:m], s[:m])[0], eng.double_base_batch(x[:m], pts[:m], s[:m])[0])
    for k in (1, 3, 189, 190, 4097, n):
        st1, r1 = eng.msm_vartime(x[:k], big[:k]); st0, r0 = eng.msm_vartime(x[:k], pts[:k])
        assert st1 == st0 == 0 and r1 == r0, k
    st1, r1 = eng.msm_consttime(x[:300], big[:300]); st0, r0 = eng.msm_consttime(x[:300], pts[:300])
    assert st1 == st0 == 0 and r1 == r0
    # order checks and the host-side fold of raw partial sums (host_from_raw160)
    assert np.array_equal(eng.point_order_checks(big[:4096], in_fmt=E.FMT_RAW160), eng.point_order_checks(pts[:4096], in_fmt=E.FMT_RAW160))
    assert eng.fold_partials([big[i].tobytes() for i in range(9)]) == eng.fold_partials([pts[i].tobytes() for i in range(9)])


def test_raw_point_with_affine_z_and_big_limbs(eng):
    """Z stored as exactly (1, 0, 0, 0, 0) takes the normaliser's no-inversion route (devio.h raw160_z_is_one); the same Z = 1
    written as (1 + 2^51 - 19 ..) must not: both give the canonical-limb result"""
    from curve25519_dalek_amd import engine as E
    rng = np.random.default_rng(3200)
    n = 4099
    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); s[:, 31] &= 0x0F
    enc = eng.mul_base_batch(s)
    aff = eng.decompress_batch(enc)[1]                     # Z = 1 exactly (edwards.rs:256)
    assert (aff.view("<u8").reshape(-1, 20)[:, 10:15] == np.array([1, 0, 0, 0, 0], dtype=np.uint64)).all()
    big = relimb(aff, rng, 8191)
    mixed = aff.copy(); mixed[::3] = big[::3]
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    want = eng.msm_vartime(x, aff)
    assert eng.msm_vartime(x, big) == want and eng.msm_vartime(x, mixed) == want


def test_msm_on_unreduced_limbs_against_the_oracle(eng):
    """The metamorphic test above compares engine(big limbs) with engine(canonical limbs); here the ORACLE is the judge: its fe51 arithmetic accepts limbs
    in [2^51, 2^52) (u64/field.rs:125-166 bounds), so the same unreduced raw points go to the oracle's Pippenger / Straus and to the HIP MSM, and the
    encodings must agree -- on the small path, the digit-matrix range and the chunk-local sort, with points outside the prime-order subgroup."""
    from curve25519_dalek_amd import engine as E
    from oracle import orc
    import util
    rng = np.random.default_rng(5150)
    rnd = util.rand_bytes(5151, 9000)
    rnd = rnd[orc.ed_decompress_ok_batch(rnd) == 1][:3000]
    st, pts, ok = eng.decompress_batch(rnd)
    assert st == 0 and ok.all()
    big = relimb(pts, rng, 1)
    lim = big.view("<u8")
    assert (lim >= (1 << 51) - 19).all() and (lim < (1 << 52)).all()
    s = util.rand_scalars(5152, big.shape[0])
    for n in (1, 7, 190, 1000, 3000):
        want = orc.ed_compress(orc.ed_msm([s[i].tobytes() for i in range(n)], [big[i].tobytes() for i in range(n)]))
        assert eng.msm_vartime(s[:n], big[:n], in_fmt=E.FMT_RAW160, out_fmt=E.FMT_EDWARDS_Y) == (0, want), n
    # the bucket pipeline (4096 .. 2^16: digit-matrix sort; above: chunk-local sort): the sum-of-squares identity on unreduced limbs, expected point by the oracle
    n = 70001
    x = util.rand_scalars(5153, n)
    pts2 = relimb(eng.mul_base_batch(x, out_fmt=E.FMT_RAW160), rng, 1)
    for m in (5000, n):
        tot = sum(int.from_bytes(x[i].tobytes(), "little") ** 2 for i in range(m)) % util.L
        want = orc.ed_compress(orc.ed_mul_base(tot.to_bytes(32, "little")))
        assert eng.msm_vartime(x[:m], pts2[:m], in_fmt=E.FMT_RAW160, out_fmt=E.FMT_EDWARDS_Y) == (0, want), m
