"""K1: the field layer AS COMPILED FOR THE GPU against Python big integers (reference tests
field.rs:552-642, limb-bound debug asserts u64/field.rs:162-166).  c25519_selftest_field runs one fe26.h operation per
element in either translation-unit flavour (chain = 1: chained-carry multiplication of kernels.hip, chain = 0: the
ten-column form of finish.hip / msm.hip), on raw limbs, so the extremes of every bound class (tight / loose / wide) reach
the device code -- asm pins, chained carries and all -- directly, not through composite outputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 2**255 - 19
POS = [0, 26, 51, 77, 102, 128, 153, 179, 204, 230]
T_EVEN, T_ODD = (1 << 26) + (1 << 19), (1 << 25) + (1 << 19)
L_EVEN, L_ODD = 204010946, 102005473
W_EVEN, W_ODD = 1 << 29, 1 << 28


@pytest.fixture(scope="module")
def eng():
    import curve25519_dalek_amd as pkg
    return pkg.Engine(0)


def values(limbs):
    """(n, 10) uint32 limbs -> list of Python ints"""
    acc = np.zeros(limbs.shape[0], dtype=object)
    for i in range(10):
        acc = acc + (limbs[:, i].astype(object) << POS[i])
    return [int(v) for v in acc]


def enc(vals):
    return np.frombuffer(b"".join(int(v % P).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32)


def rand_limbs(rng, n, even_max, odd_max):
    a = np.empty((n, 10), dtype=np.uint32)
    for i in range(10):
        a[:, i] = rng.integers(0, (odd_max if i & 1 else even_max) + 1, size=n, dtype=np.uint64).astype(np.uint32)
    return a


def edge_limbs(even_max, odd_max):
    """every limb at 0 / at its bound, alone and together; p, p - 1, 2p - 1 in reduced limbs; alternating patterns"""
    rows = [[0] * 10, [(odd_max if i & 1 else even_max) for i in range(10)]]
    for i in range(10):
        r = [0] * 10; r[i] = odd_max if i & 1 else even_max; rows.append(r)
        r = [(odd_max if j & 1 else even_max) for j in range(10)]; r[i] = 0; rows.append(r)
    for v in (P, P - 1, P + 1, 1, 2, 19, 2**255 - 1, 2**255 - 20, 2**254, (1 << 230) - 1):
        rows.append([(v >> POS[i]) & ((1 << (25 if i & 1 else 26)) - 1) for i in range(10)])
    rows.append([(even_max if i % 4 == 0 else 0) for i in range(10)])
    rows.append([(odd_max if i & 1 else 0) for i in range(10)])
    return np.array(rows, dtype=np.uint32)


def with_edges(rng, n, even_max, odd_max):
    e = edge_limbs(even_max, odd_max)
    return np.concatenate([e, rand_limbs(rng, n - e.shape[0], even_max, odd_max)])


@pytest.mark.parametrize("chain", [0, 1])
def test_fe_mul_sq_to_bytes_2p20(eng, chain):
    """2^20 random + bound-extreme operands through fe_mul (wide x loose), fe_sq (loose), the canonical encoder and the
    weak reduction, in both carry forms"""
    rng = np.random.default_rng(2600 + chain)
    n = 1 << 20
    a = with_edges(rng, n, W_EVEN, W_ODD); b = with_edges(rng, n, L_EVEN, L_ODD)
    b[:64] = b[:64][::-1].copy()                          # the edge rows meet each other in both orders
    va, vb = values(a), values(b)
    assert np.array_equal(eng.selftest_field(0, a, b, chain), enc([x * y for x, y in zip(va, vb)]))
    # all-maximal operands: the largest column sums the bound classes admit (fe26.h: 124.5 * 2^29 * 1.52 * 2^27 < 2^64)
    amax = np.tile(np.array([[W_EVEN if i % 2 == 0 else W_ODD for i in range(10)]], np.uint32), (256, 1))
    bmax = np.tile(np.array([[L_EVEN if i % 2 == 0 else L_ODD for i in range(10)]], np.uint32), (256, 1))
    assert np.array_equal(eng.selftest_field(0, amax, bmax, chain), enc([x * y for x, y in zip(values(amax), values(bmax))]))
    l = with_edges(rng, n, L_EVEN, L_ODD)
    vl = values(l)
    assert np.array_equal(eng.selftest_field(1, l, None, chain), enc([x * x for x in vl]))
    assert np.array_equal(eng.selftest_field(3, a, None, chain), enc(va))                       # to_bytes of wide limbs
    anyl = rng.integers(0, 1 << 32, size=(n, 10), dtype=np.uint64).astype(np.uint32)            # fe_carry takes ANY u32 limbs
    assert np.array_equal(eng.selftest_field(6, anyl, None, chain), enc(values(anyl)))


@pytest.mark.parametrize("chain", [0, 1])
def test_lockstep_products_2p20(eng, chain):
    """fe_mul_chain_n<3> / <4> (fe26x.h: the seven products of a bucket addition, chained carries, issued column by
    column): every slot of both group sizes against big integers, wide x loose operands with the bound extremes"""
    rng = np.random.default_rng(2700 + chain)
    n = 1 << 20
    a = with_edges(rng, n, W_EVEN, W_ODD); b = with_edges(rng, n, L_EVEN, L_ODD)
    b[:64] = b[:64][::-1].copy()
    a[64:320] = np.array([[W_EVEN if i % 2 == 0 else W_ODD for i in range(10)]], np.uint32)     # all-maximal operands
    b[64:320] = np.array([[L_EVEN if i % 2 == 0 else L_ODD for i in range(10)]], np.uint32)
    va, vb = values(a), values(b)
    ab = enc([x * y for x, y in zip(va, vb)]); bb = enc([y * y for y in vb])
    assert np.array_equal(eng.selftest_field(8, a, b, chain), ab)
    assert np.array_equal(eng.selftest_field(9, a, b, chain), bb)
    assert np.array_equal(eng.selftest_field(10, a, b, chain), ab)
    assert np.array_equal(eng.selftest_field(11, a, b, chain), bb)


@pytest.mark.parametrize("chain", [0, 1])
def test_fe_invert_pow_sub_chains(eng, chain):
    rng = np.random.default_rng(2700 + chain)
    n = 1 << 14
    t = with_edges(rng, n, T_EVEN, T_ODD)
    vt = values(t)
    assert np.array_equal(eng.selftest_field(2, t, None, chain), enc([pow(x, P - 2, P) for x in vt]))           # 0 -> 0 included
    assert np.array_equal(eng.selftest_field(4, t, None, chain), enc([pow(x, (P - 5) // 8, P) for x in vt]))
    n = 1 << 18
    a = with_edges(rng, n, L_EVEN, L_ODD); b = with_edges(rng, n, L_EVEN, L_ODD)
    b[:64] = b[:64][::-1].copy()
    va, vb = values(a), values(b)
    assert np.array_equal(eng.selftest_field(5, a, b, chain), enc([x - y for x, y in zip(va, vb)]))               # loose - loose -> wide
    ta = with_edges(rng, n, T_EVEN, T_ODD); tb = with_edges(rng, n, T_EVEN, T_ODD)
    tb[:64] = tb[:64][::-1].copy()
    xa, xb = values(ta), values(tb)
    assert np.array_equal(eng.selftest_field(7, ta, tb, chain), enc([(x - y) * (x + y) for x, y in zip(xa, xb)]))   # (T - T) * (T + T)
