"""CPU fuzz of the DEVICE arithmetic headers (curve25519-dalek_amd/csrc/fe26.h, ge26.h).

The headers are __host__ __device__; tests/host/fe26_host.cpp builds them for the host with
C25519_CHECK_BOUNDS, so every limb-bound assumption documented in fe26.h aborts the process if it
is violated.  Each operation is compared with the oracle (same inputs) and with Python big-ints, at
random points AND at the extreme limb values each bound class allows.  No GPU needed.
"""
import ctypes as C
import os
import random
import subprocess

import pytest

import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
POS = [0, 26, 51, 77, 102, 128, 153, 179, 204, 230]
T_EVEN, T_ODD = (1 << 26) + (1 << 19), (1 << 25) + (1 << 19)
L_EVEN, L_ODD = 204010946, 102005473
W_EVEN, W_ODD = 1 << 29, 1 << 28


def i2b(x):
    return int(x).to_bytes(32, "little")


def b2i(b):
    return int.from_bytes(b, "little")


@pytest.fixture(scope="module", params=[0, 1], ids=["columns", "chained"])
def host(request):
    """Both forms of fe_mul / fe_sq (fe26.h C25519_CHAIN: independent column sums, chained carries)."""
    src = os.path.join(ROOT, "tests", "host", "fe26_host.cpp")
    so = os.path.join(ROOT, "tests", "host", "libfe26host%d.so" % request.param)
    deps = [src] + [os.path.join(ROOT, "curve25519-dalek_amd", "csrc", f) for f in ("fe26.h", "fe9_probe.h", "ge26.h", "sc_sha.h", "sc28.h", "transcript_host.h", "constants_gen.h", "host51.h", "blake2b.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DC25519_CHAIN=%d" % request.param, "-DC25519_KECCAK_MAX_FORM=%d" % (1 + request.param), "-o", so, src])      # (the "columns" build also keeps the host transcript on its scalar Keccak-f forms)
    return C.CDLL(so)


def call(host, name, *args, out=32):
    o = C.create_string_buffer(out)
    getattr(host, name)(*args, o)
    return o.raw


def limbs_val(l):
    return sum(v << POS[i] for i, v in enumerate(l))


def rand_limbs(rng, emax, omax, extreme):
    out = []
    for i in range(10):
        m = omax if i & 1 else emax
        out.append(m if (extreme and rng.random() < 0.7) else rng.randrange(m + 1))
    return out


def test_field_ops_vs_bigint_and_oracle(host, orc):
    rng = random.Random(100)
    edge = [0, 1, 2, 19, P - 1, P, P + 1, 2**255 - 1, 2**255 - 19 + 18, 2**256 - 1, (1 << 255) - 20]
    vals = edge + [rng.getrandbits(256) for _ in range(300)]
    for a in vals:
        ab = i2b(a); am = (a & (2**255 - 1)) % P
        assert b2i(call(host, "h_fe_canon", ab)) == am
        assert b2i(call(host, "h_fe_sq", ab)) == am * am % P
        for b in (rng.choice(vals), rng.getrandbits(256)):
            bb = i2b(b); bm = (b & (2**255 - 1)) % P
            assert b2i(call(host, "h_fe_mul", ab, bb)) == am * bm % P
            assert call(host, "h_fe_mul", ab, bb) == orc.fe_mul(ab, bb)
            assert b2i(call(host, "h_fe_add", ab, bb)) == (am + bm) % P
            assert b2i(call(host, "h_fe_sub", ab, bb)) == (am - bm) % P
        assert b2i(call(host, "h_fe_mul_small", ab, C.c_uint32(121666))) == am * 121666 % P
    for a in edge + [rng.getrandbits(255) for _ in range(20)]:
        am = (a & (2**255 - 1)) % P
        assert b2i(call(host, "h_fe_invert", i2b(a))) == pow(am, P - 2, P)
        assert call(host, "h_fe_pow_p58", i2b(a)) == orc.fe_pow_p58(i2b(a))


def test_mul_sq_at_bound_extremes(host):
    """fe_mul(f wide, g loose), fe_sq(loose), fe_carry / to_bytes of anything < 2^32."""
    rng = random.Random(101)
    A10 = C.c_uint32 * 10
    for it in range(400):
        extreme = it % 2 == 0
        f = rand_limbs(rng, W_EVEN, W_ODD, extreme); g = rand_limbs(rng, L_EVEN, L_ODD, extreme)
        assert b2i(call(host, "h_fe_mul_limbs", A10(*f), A10(*g))) == limbs_val(f) * limbs_val(g) % P
        s = rand_limbs(rng, L_EVEN, L_ODD, extreme)
        assert b2i(call(host, "h_fe_sq_limbs", A10(*s))) == limbs_val(s) ** 2 % P
        x = rand_limbs(rng, 2**32 - 1, 2**32 - 1, extreme)
        assert b2i(call(host, "h_fe_carry_limbs", A10(*x))) == limbs_val(x) % P
        assert b2i(call(host, "h_fe_tobytes_limbs", A10(*x))) == limbs_val(x) % P
    allmax_f = [W_ODD if i & 1 else W_EVEN for i in range(10)]
    allmax_g = [L_ODD if i & 1 else L_EVEN for i in range(10)]
    assert b2i(call(host, "h_fe_mul_limbs", A10(*allmax_f), A10(*allmax_g))) == limbs_val(allmax_f) * limbs_val(allmax_g) % P
    assert b2i(call(host, "h_fe_sq_limbs", A10(*allmax_g))) == limbs_val(allmax_g) ** 2 % P
    # values just around p and 2p in limb form
    for v in (P - 1, P, P + 1, 2 * P - 1, 2 * P, 2 * P + 1, 2**255 - 1, 2**255, 2**256 - 38):
        l = [(v >> POS[i]) & ((1 << (25 if i & 1 else 26)) - 1) for i in range(9)] + [v >> 230]
        assert b2i(call(host, "h_fe_tobytes_limbs", A10(*l))) == v % P


def test_sqrt_ratio_and_decompress_vs_oracle(host, orc):
    rng = random.Random(102)
    cases = [(0, 0), (1, 0), (2, 1), (4, 1), (1, 4), (0, 5)] + [(rng.getrandbits(255), rng.getrandbits(255)) for _ in range(40)]
    for u, v in cases:
        o = C.create_string_buffer(32)
        ok = host.h_fe_sqrt_ratio_i(i2b(u), i2b(v), o)
        ook, oval = orc.fe_sqrt_ratio_i(i2b(u), i2b(v))
        assert bool(ok) == ook and o.raw == oval
    encs = [i2b(1), i2b(0), i2b(P + 1), i2b(2**255 + 1), i2b(2**256 - 1), i2b(2)] + [i2b(rng.getrandbits(256)) for _ in range(120)]
    for e in encs:
        o = C.create_string_buffer(128)
        ok = host.h_ge_decompress(e, o)
        want = orc.ed_decompress(e)
        assert bool(ok) == (want is not None)
        if ok:
            assert call(host, "h_ge_compress", o.raw) == orc.ed_compress(want)


def test_point_formulas_vs_oracle(host, orc):
    rng = random.Random(103)

    def to_host(p160):  # oracle 160-byte limb blob -> 4 x 32 canonical bytes
        X, Y, Z, T = (p160[40 * i:40 * i + 40] for i in range(4))
        def canon(l40):
            limbs = [int.from_bytes(l40[8 * i:8 * i + 8], "little") for i in range(5)]
            return i2b(sum(v << (51 * i) for i, v in enumerate(limbs)) % P)
        return canon(X) + canon(Y) + canon(Z) + canon(T)

    B = call(host, "h_ge_basepoint", out=128)
    assert call(host, "h_ge_compress", B) == orc.ed_compress(orc.ed_basepoint())
    assert host.h_ge_is_identity(call(host, "h_ge_identity", out=128))
    pts = []
    for _ in range(12):
        k = rng.randrange(L)
        po = orc.ed_mul_base(i2b(k))
        pts.append((po, to_host(po)))
    for (ao, ah) in pts[:6]:
        for (bo, bh) in pts[6:]:
            assert call(host, "h_ge_compress", call(host, "h_ge_add", ah, bh, out=128)) == orc.ed_compress(orc.ed_add(ao, bo))
        assert call(host, "h_ge_compress", call(host, "h_ge_dbl", ah, out=128)) == orc.ed_compress(orc.ed_double(ao))
        assert call(host, "h_ge_compress", call(host, "h_ge_neg", ah, out=128)) == orc.ed_compress(orc.ed_neg(ao))
        assert call(host, "h_ge_compress", call(host, "h_ge_mul_by_pow_2", ah, C.c_int(6), out=128)) == orc.ed_compress(orc.ed_mul_by_pow_2(ao, 6))
        # exceptional-looking operands go through the same complete formulas
        assert host.h_ge_is_identity(call(host, "h_ge_add", ah, call(host, "h_ge_neg", ah, out=128), out=128))
        assert call(host, "h_ge_compress", call(host, "h_ge_add", ah, ah, out=128)) == orc.ed_compress(orc.ed_double(ao))
        assert host.h_ge_eq(call(host, "h_ge_add", ah, call(host, "h_ge_identity", out=128), out=128), ah)
    # mixed add/sub with an affine (Z=1) operand
    for (ao, ah) in pts[:5]:
        q = orc.ed_decompress(orc.ed_compress(pts[7][0]))  # affine copy, Z = 1
        qh = to_host(q)
        assert call(host, "h_ge_compress", call(host, "h_ge_madd", ah, qh, C.c_int(0), out=128)) == orc.ed_compress(orc.ed_add(ao, q))
        assert call(host, "h_ge_compress", call(host, "h_ge_madd", ah, qh, C.c_int(1), out=128)) == orc.ed_compress(orc.ed_sub(ao, q))
        assert call(host, "h_ge_compress", call(host, "h_ge_madd_signed", ah, qh, C.c_int(0), out=128)) == orc.ed_compress(orc.ed_add(ao, q))
        assert call(host, "h_ge_compress", call(host, "h_ge_madd_signed", ah, qh, C.c_int(1), out=128)) == orc.ed_compress(orc.ed_sub(ao, q))
        # the first addition of a chain: +/- q straight from the affine Niels form, then used as an accumulator
        for neg in (0, 1):
            first = call(host, "h_ge_from_aniels_signed", qh, C.c_int(neg), out=128)
            assert call(host, "h_ge_compress", first, out=32) == orc.ed_compress(orc.ed_neg(q) if neg else q)
            nxt = call(host, "h_ge_madd_signed", first, to_host(orc.ed_decompress(orc.ed_compress(pts[3][0]))), C.c_int(0), out=128)
            assert call(host, "h_ge_compress", nxt, out=32) == orc.ed_compress(orc.ed_add(orc.ed_neg(q) if neg else q, pts[3][0]))
        # chained: the output of the signed form feeds the next addition (bounds of the next fe_mul operands)
        acc = ah
        for i in range(20):
            acc = call(host, "h_ge_madd_signed", acc, qh, C.c_int(i & 1), out=128)
        assert call(host, "h_ge_compress", acc, out=32) == orc.ed_compress(ao)
        for (bo, bh) in pts[8:11]:
            assert call(host, "h_ge_compress", call(host, "h_ge_add_cached_signed", ah, bh, C.c_int(0), out=128)) == orc.ed_compress(orc.ed_add(ao, bo))
            assert call(host, "h_ge_compress", call(host, "h_ge_add_cached_signed", ah, bh, C.c_int(1), out=128)) == orc.ed_compress(orc.ed_sub(ao, bo))
    # madd with the identity as the affine operand, both signs (xy2d = 0 -> p - 0 = p, non-canonical zero)
    ident = to_host(orc.ed_identity())
    for neg in (0, 1):
        assert call(host, "h_ge_compress", call(host, "h_ge_madd", pts[0][1], ident, C.c_int(neg), out=128)) == orc.ed_compress(pts[0][0])
        assert call(host, "h_ge_compress", call(host, "h_ge_madd_signed", pts[0][1], ident, C.c_int(neg), out=128)) == orc.ed_compress(pts[0][0])
        assert host.h_ge_is_identity(call(host, "h_ge_from_aniels_signed", ident, C.c_int(neg), out=128))
    # a long chain (the reference's overflow hunt, edwards.rs:2254-2261): 300 chained doublings+adds
    acc_h, acc_o = pts[0][1], pts[0][0]
    for i in range(300):
        acc_h = call(host, "h_ge_dbl", acc_h, out=128); acc_o = orc.ed_double(acc_o)
        acc_h = call(host, "h_ge_add", acc_h, pts[i % 12][1], out=128); acc_o = orc.ed_add(acc_o, pts[i % 12][0])
    assert call(host, "h_ge_compress", acc_h) == orc.ed_compress(acc_o)


def test_ladder_vs_oracle(host, orc, golden):
    rng = random.Random(104)
    cases = [(golden.bytes("x25519_tests.rs", "input_scalar", fn="rfc7748_ladder_test1_vectorset1"),
              golden.bytes("x25519_tests.rs", "input_point", fn="rfc7748_ladder_test1_vectorset1"))]
    cases += [(i2b(rng.getrandbits(256)), i2b(rng.getrandbits(256))) for _ in range(25)]
    cases += [(i2b(rng.getrandbits(256)), golden.bytes("src/constants.rs", "X25519_LOW_ORDER_POINTS", i)) for i in range(7)]
    for k, u in cases:
        s = orc.sc_clamp(k)
        assert call(host, "h_x25519_ladder", s, u) == orc.x25519(k, u) == pyref.x25519(k, u)


def test_scalar_sha_transcript_host_vs_oracle(host, orc, request):
    """the verify_batch pipeline's scalar arithmetic (sc_sha.h), SHA-512 and host transcript"""
    import hashlib
    rng = random.Random(105)
    for _ in range(200):
        a, b = rng.randrange(L), rng.randrange(L)
        assert b2i(call(host, "h_sc_mul", i2b(a), i2b(b))) == a * b % L
        assert b2i(call(host, "h_sc_add", i2b(a), i2b(b))) == (a + b) % L
        assert b2i(call(host, "h_sc_neg", i2b(a))) == (-a) % L
        w = rng.getrandbits(512)
        assert b2i(call(host, "h_sc_from_wide", w.to_bytes(64, "little"))) == w % L
        z = rng.getrandbits(128)
        assert b2i(call(host, "h_sc_mul", i2b(z), i2b(b))) == z * b % L
    for v, want in [(0, 1), (L - 1, 1), (L, 0), (L + 1, 0), (2**255, 0), (2**252, 1), (2**256 - 1, 0)]:
        assert host.h_sc_is_canonical(i2b(v)) == want
    for n in [0, 1, 7, 8, 63, 64, 65, 95, 96, 111, 112, 113, 119, 120, 127, 128, 129, 200, 255, 256, 300]:
        m = bytes(rng.getrandbits(8) for _ in range(n))
        assert call(host, "h_sha512", m, C.c_size_t(n), out=64) == hashlib.sha512(m).digest(), n
        # the kernels' message path (sha512_stream::put_bytes): every block position it can start at, aligned and unaligned sources
        for pre in (0, 1, 3, 8, 13, 64, 121, 127):
            for shift in (0, 1, 2):
                buf = bytes(shift) + m                                           # move the message off its 4-byte alignment
                src = (C.c_char * len(buf)).from_buffer_copy(buf)
                out = C.create_string_buffer(64)
                host.h_sha512_put_bytes(C.byref(src, shift), C.c_size_t(n), C.c_size_t(pre), out)
                assert out.raw == hashlib.sha512(m).digest(), (n, pre, shift)
    # Keccak-f[1600] in every form this host can run (csrc/transcript_host.h: generic, BMI2, AVX-512VL lane pairs) against the spec-level permutation of tests/pyref.py
    import pyref
    ran = set()
    for it in range(40):
        st = bytes(200) if it == 0 else bytes([0xFF]) * 200 if it == 1 else bytes(rng.getrandbits(8) for _ in range(200))
        want = bytearray(st); pyref.keccak_f1600(want)
        for which in (0, 1, 2):
            o = C.create_string_buffer(200)
            if host.h_keccak_form(st, which, o):
                ran.add(which)
                assert o.raw == bytes(want), (it, which)
    assert 0 in ran
    host.h_keccak_impl.restype = C.c_char_p
    assert host.h_keccak_impl() in (b"generic", b"bmi2", b"avx512vl-pairs")
    if request.node.callspec.params["host"] == 0:
        assert host.h_keccak_impl() != b"avx512vl-pairs"           # this build's transcripts below take the scalar forms and the written-out whole-lane squeeze
    # n = 333: 76- and 45-byte framed messages start at every (even / any) position of the 166-byte block, so both the in-block fast path of append_message and
    # the boundary-crossing duplex calls are taken at every offset
    for n in [1, 2, 7, 40, 333]:
        hr = [bytes(rng.getrandbits(8) for _ in range(64)) for _ in range(n)]
        sg = [bytes(rng.getrandbits(8) for _ in range(64)) for _ in range(n)]
        got = call(host, "h_transcript_zs", b"".join(hr), b"".join(sg), C.c_uint64(n), out=16 * n)
        want = b"".join(orc.batch_transcript_zs(hr, [s[32:] for s in sg]))
        assert got == want


def test_sc28_device_scalar_arithmetic_vs_bigint(host):
    """csrc/sc28.h -- the scalar arithmetic the verify_batch / sign / verify kernels run (radix 2^28, folding with l = 2^252 + c) --
    against Python integers: wide reduction (Scalar::from_bytes_mod_order_wide, scalar.rs:248), 128-bit x scalar and scalar x
    unreduced-255-bit products (batch.rs:213-233, signing.rs:899), add / neg, the canonical check (scalar.rs:259-263), at random
    values and at the extremes where a fold's bound is tight"""
    rng = random.Random(28)
    C_ = L - 2**252
    wides = [0, 1, L - 1, L, L + 1, 2**252 - 1, 2**252, 2**512 - 1, 2**512 - L, (2**260 - 1) << 252, ((2**260 - 1) << 252) + 2**252 - 1,
             2**511, 2**384, 2**385 - 1, 2**266, 2**253, 2**254 - 1, 7 * L, (2**259) * L % 2**512, L * (2**259 - 1)]
    wides += [rng.getrandbits(512) for _ in range(3000)] + [rng.getrandbits(rng.randrange(1, 513)) for _ in range(1000)]
    for w in wides:
        assert b2i(call(host, "h_sc28_from_wide", w.to_bytes(64, "little"))) == w % L, hex(w)
    zs = [0, 1, 2**128 - 1, 2**127, 2**127 - 1, 2**64] + [rng.getrandbits(128) for _ in range(2000)]
    bs = [0, 1, L - 1, L - 2, 2**252, 2**252 - 1, C_, L // 2] + [rng.randrange(L) for _ in range(60)]
    for z in zs[:40]:
        for b in bs:
            assert b2i(call(host, "h_sc28_mul_5x10", z.to_bytes(16, "little"), i2b(b))) == z * b % L, (z, b)
    for z in zs[40:]:
        b = rng.randrange(L)
        assert b2i(call(host, "h_sc28_mul_5x10", z.to_bytes(16, "little"), i2b(b))) == z * b % L
    full = [0, 1, L - 1, 2**255 - 1, 2**255 - 8, 2**254 + 2**253, 2**252 + 1] + [rng.getrandbits(255) for _ in range(40)]
    for a in full:
        for b in full[:12] + [rng.randrange(L) for _ in range(20)]:
            assert b2i(call(host, "h_sc28_mul", i2b(a), i2b(b))) == a * b % L, (a, b)
    for _ in range(2000):
        a, b = rng.randrange(L), rng.randrange(L)
        assert b2i(call(host, "h_sc28_add", i2b(a), i2b(b))) == (a + b) % L
        assert b2i(call(host, "h_sc28_neg", i2b(a))) == (-a) % L
        assert b2i(call(host, "h_sc28_roundtrip", i2b(a))) == a
    for a, b in ((0, 0), (L - 1, L - 1), (L - 1, 1), (1, L - 1), (0, L - 1)):
        assert b2i(call(host, "h_sc28_add", i2b(a), i2b(b))) == (a + b) % L
    assert b2i(call(host, "h_sc28_neg", i2b(0))) == 0
    assert b2i(call(host, "h_sc28_roundtrip", i2b(2**256 - 1))) == 2**256 - 1
    for v, want in [(0, 1), (1, 1), (L - 1, 1), (L, 0), (L + 1, 0), (2**255 - 1, 0), (2**255, 0), (2**256 - 1, 0), (2**252, 1), (2**252 + C_ - 1, 1)] + \
                   [(rng.getrandbits(256), None) for _ in range(300)] + [(L - d, None) for d in range(-40, 40)]:
        assert host.h_sc28_canonical(i2b(v)) == (want if want is not None else int(v < L)), hex(v)


def test_nine_limb_probe_products_vs_bigint(host):
    """csrc/fe9_probe.h (the 9 x 28.33-bit representation priced by c25519_microbench 70 / 71): product and square equal the big-integer
    product mod p for random limbs and for limbs at their extreme -- every limb at 2^width - 1 + the slack a carried output may have
    (limb 1 takes the final x19 carry) -- and the outputs are carried again (so products can be chained)."""
    pos = [(85 * i + 2) // 3 for i in range(10)]
    wid = [pos[i + 1] - pos[i] for i in range(9)]
    assert pos[:9] == [0, 29, 57, 85, 114, 142, 170, 199, 227] and pos[9] == 255
    rng = random.Random(909)
    val = lambda l: sum(int(v) << pos[i] for i, v in enumerate(l))
    def run(op, a, b):
        A = (C.c_uint32 * 9)(*a); B = (C.c_uint32 * 9)(*b); O = (C.c_uint32 * 9)()
        host.h_fe9(op, A, B, O)
        return list(O)
    for t in range(600):
        extreme = t % 3 == 0
        slack = 1 << 20                                      # what a carried output may exceed its width by
        mk = lambda: [((1 << wid[i]) - 1 + (slack if i == 1 else 0)) if (extreme and rng.random() < 0.8) else rng.randrange(1 << wid[i]) for i in range(9)]
        a, b = mk(), mk()
        r = run(0, a, b)
        assert val(r) % P == val(a) * val(b) % P
        assert all(r[i] < (1 << wid[i]) + (slack if i == 1 else 0) for i in range(9)), r
        q = run(1, a, a)
        assert val(q) % P == val(a) ** 2 % P
        r2 = run(0, r, q)                                    # outputs are valid inputs
        assert val(r2) % P == val(r) * val(q) % P


def test_ifma_doubling_chain(host):
    """(r6) The Horner fold's doubling chain on AVX-512 IFMA (csrc/host51.h hp3_mul_by_pow_2_ifma: four field operations per vector instruction, radix 2^51 in the
    lanes of 256-bit vectors) against the scalar 64 x 64 -> 128 form and against Python big integers (2^k P on the curve by the affine doubling law): random projective
    points, limbs at the extremes the inputs may have (2^51 + 2^19 after a carry pass, all-ones 51-bit limbs), the identity, points of small order, 1 .. 40 and 253
    doublings.  Skipped where the CPU has no avx512ifma (the library then runs the scalar form, which every other test exercises)."""
    if not host.h_has_ifma():
        pytest.skip("no avx512ifma on this CPU")
    rng = random.Random(99)
    d = (-121665 * pow(121666, P - 2, P)) % P

    def dbl(x, y):                                   # affine doubling on -x^2 + y^2 = 1 + d x^2 y^2
        x3 = (2 * x * y) * pow((1 + d * x * x * y * y) % P, P - 2, P) % P
        y3 = (y * y + x * x) * pow((1 - d * x * x * y * y) % P, P - 2, P) % P
        return x3, y3

    def point():
        while True:
            y = rng.randrange(P)
            u, v = (y * y - 1) % P, (d * y * y + 1) % P
            x2 = u * pow(v, P - 2, P) % P
            x = pow(x2, (P + 3) // 8, P)
            if (x * x - x2) % P:
                x = x * pow(2, (P - 1) // 4, P) % P
            if (x * x - x2) % P == 0:
                return x, y

    def limbs(v, style):
        v %= P
        l = [(v >> (51 * i)) & ((1 << 51) - 1) for i in range(5)]
        if style == 1:                                # the same element plus p, limb by limb: limbs up to 2^52 - 20 (an un-carried sum of two reduced elements)
            pl = [(1 << 51) - 19] + [(1 << 51) - 1] * 4
            l = [a + b for a, b in zip(l, pl)]
        return l

    cases = [(0, 1)] + [point() for _ in range(40)]
    cases.append((0, P - 1))                          # order 2
    cases.append((pow(2, (P - 1) // 4, P), 0))        # order 4
    for idx, (x, y) in enumerate(cases):
        for k in ([1, 2, 3, 5, 13, 16, 40, 253] if idx < 6 else [rng.randrange(1, 20)]):
            z = rng.randrange(1, P) if idx else 1
            xyz = limbs(x * z, idx & 1) + limbs(y * z, 0) + limbs(z, idx & 1)
            arr = (C.c_uint64 * 15)(*xyz)
            a, b = C.create_string_buffer(96), C.create_string_buffer(96)
            host.h_pow2_scalar(arr, k, a); host.h_pow2_ifma(arr, k, b)
            assert a.raw == b.raw, (idx, k)
            ex, ey = x, y
            for _ in range(k):
                ex, ey = dbl(ex, ey)
            assert b.raw[:32] == i2b(ex) and b.raw[32:64] == i2b(ey) and b.raw[64:] == i2b(ex * ey % P), (idx, k)


def test_ifma_horner_fold(host):
    """(r6) The whole Horner fold with the running total in lane form (csrc/host51.h hp3_horner: p4_pow2 / p4_cached / p4_add on AVX-512 IFMA) against the scalar
    fold and against Python big integers (affine addition law): 1 .. 56 columns -- random projective points, the identity, points of order 2 and 4, equal neighbours
    (the addition must be complete) -- with the shifts of real window layouts (5 .. 17 bits).  On a CPU without avx512ifma both calls run the scalar code."""
    rng = random.Random(123)
    d = (-121665 * pow(121666, P - 2, P)) % P

    def add(p, q):
        (x1, y1), (x2, y2) = p, q
        k = d * x1 * x2 * y1 * y2 % P
        return ((x1 * y2 + y1 * x2) * pow(1 + k, P - 2, P) % P, (y1 * y2 + x1 * x2) * pow(1 - k, P - 2, P) % P)

    def point():
        while True:
            y = rng.randrange(P)
            u, v = (y * y - 1) % P, (d * y * y + 1) % P
            x2 = u * pow(v, P - 2, P) % P
            x = pow(x2, (P + 3) // 8, P)
            if (x * x - x2) % P:
                x = x * pow(2, (P - 1) // 4, P) % P
            if (x * x - x2) % P == 0:
                return x, y

    def limbs(v):
        v %= P
        return [(v >> (51 * i)) & ((1 << 51) - 1) for i in range(5)]

    special = [(0, 1), (0, P - 1), (pow(2, (P - 1) // 4, P), 0)]
    for trial in range(30):
        n = rng.choice([1, 2, 3, 16, 22, 43, 51, 56])
        c = rng.randrange(5, 18)
        pts = []
        for k in range(n):
            r = rng.random()
            pts.append(rng.choice(special) if r < 0.15 else (pts[-1] if r < 0.25 and pts else point()))
        shifts = [0] + [rng.choice([c, c - 1, c - 2, 1, 4]) for _ in range(n - 1)]
        flat = []
        for (x, y) in pts:
            z = rng.randrange(1, P)
            flat += limbs(x * z) + limbs(y * z) + limbs(z) + limbs(x * y * z)
        arr = (C.c_uint64 * len(flat))(*flat); sh = (C.c_int * n)(*shifts)
        a, b = C.create_string_buffer(96), C.create_string_buffer(96)
        host.h_horner(arr, sh, n, 0, a); host.h_horner(arr, sh, n, 1, b)
        assert a.raw == b.raw, (trial, n)
        tot = (0, 1)
        for k in range(n):
            for _ in range(shifts[k] if k else 0):
                tot = add(tot, tot)
            tot = add(tot, pts[k])
        assert b.raw[:32] == i2b(tot[0]) and b.raw[32:64] == i2b(tot[1]) and b.raw[64:] == i2b(tot[0] * tot[1] % P), (trial, n)


def test_blake2b_compression_function_vs_hashlib(host):
    """csrc/blake2b.h (the hash of the device z-mode's tree, verify.hip) driven as the plain unkeyed BLAKE2b of RFC 7693: the RFC's "abc" vector (appendix A) and
    hashlib.blake2b on random messages around every block boundary, digest lengths 32 and 64."""
    import hashlib
    abc = ("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
           "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")
    o = C.create_string_buffer(64)
    host.h_blake2b(b"abc", C.c_uint64(3), C.c_uint32(64), o)
    assert o.raw.hex() == abc == hashlib.blake2b(b"abc").hexdigest()
    rng = random.Random(77)
    for n in [0, 1, 39, 40, 41, 111, 127, 128, 129, 255, 256, 257, 383, 384, 385, 1000]:
        for outlen in (32, 64):
            msg = bytes(rng.randrange(256) for _ in range(n))
            o = C.create_string_buffer(outlen)
            host.h_blake2b(msg, C.c_uint64(n), C.c_uint32(outlen), o)
            assert o.raw == hashlib.blake2b(msg, digest_size=outlen).digest(), (n, outlen)
