"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs -- bit-exact on canonical encodings / flags.
Nothing here reads /root/reference."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import curve25519_dalek_amd as pkg
    return pkg.Engine(0)


@pytest.fixture(scope="module")
def torch():
    import torch
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_microbench_reports(eng):
    names = ["v_mad_u64_u32", "fe_mul(2^25.5)", "fe_sq", "fe_mul(5x51,u128)", "v_add_u32", "v_mul_lo_u32"]
    for i, nm in enumerate(names):
        r = eng.microbench(i, 4000)
        print("MICROBENCH %-20s %10.1f Gop/s" % (nm, r))
        assert r > 1.0


@pytest.mark.parametrize("window", ["ct", 0, 4, 5, 6, 9, 10, 11, 13, 20])
def test_mul_base_vs_oracle(orc, window):
    """every fixed-base algorithm: the constant-time full-scan tables (the default for mul_base: the reference's is
    constant-time) and, on a VARTIME_TABLES context, each of the fast table variants"""
    import curve25519_dalek_amd as pkg
    e = pkg.Engine(0) if window == "ct" else pkg.Engine(0, window=window, flags=pkg.engine.FLAG_VARTIME_TABLES)
    s = np.concatenate([util.edge_scalars(), util.rand_scalars(11, 3000), util.rand_bytes(12, 500) & np.uint8(0xFF)])
    s[7:40, 0] &= 0xFE        # plenty of even scalars (the comb's parity correction)
    s[-500:, 31] &= 0x7F   # unreduced but < 2^255
    got = e.mul_base_batch(s)
    want = orc.mul_base_compress_batch(s, threads=8)
    assert np.array_equal(got, want)
    # raw 160-byte output: compare through the reference's compress
    raw = e.mul_base_batch(s[:300], out_fmt=2)
    assert [orc.ed_compress(raw[i].tobytes()) for i in range(300)] == [want[i].tobytes() for i in range(300)]
    # empty batch
    assert e.mul_base_batch(np.zeros((0, 32), np.uint8)).shape == (0, 32)
    e.close()


def test_mul_base_full_size_2p20(eng, orc, torch):
    """BASELINE configs[1] at full size: ALL 2^20 outputs bit-exact against the oracle, for the three independent
    algorithms (constant-time scan over radix-2^5 LDS tables, radix-2^16 tables in HBM, signed comb in LDS)."""
    import os
    import curve25519_dalek_amd as pkg
    n = 1 << 20
    s = util.rand_scalars(21, n)
    ds = dev(torch, s)
    want = orc.mul_base_compress_batch(s, threads=os.cpu_count() or 8)
    assert np.array_equal(eng.mul_base_batch_t(ds).cpu().numpy(), want)                  # constant-time default
    assert np.array_equal(eng.mul_base_batch_vartime_t(ds).cpu().numpy(), want)          # radix-2^16 tables
    e5 = pkg.Engine(0, window=9, flags=pkg.engine.FLAG_VARTIME_TABLES)
    assert np.array_equal(e5.mul_base_batch_t(ds).cpu().numpy(), want)                   # comb
    e5.close()


def test_x25519_vs_oracle(eng, orc, golden):
    f = "x25519_tests.rs"
    ks = [golden.bytes(f, "input_scalar", fn="rfc7748_ladder_test1_vectorset1"), golden.bytes(f, "input_scalar", fn="rfc7748_ladder_test1_vectorset2"),
          golden.bytes(f, "ALICE_PRIVATE_KEY"), golden.bytes(f, "BOB_PRIVATE_KEY")]
    us = [golden.bytes(f, "input_point", fn="rfc7748_ladder_test1_vectorset1"), golden.bytes(f, "input_point", fn="rfc7748_ladder_test1_vectorset2"),
          golden.bytes(f, "BOB_PUBLIC_KEY"), golden.bytes(f, "ALICE_PUBLIC_KEY")]
    exp = [golden.bytes(f, "expected", fn="rfc7748_ladder_test1_vectorset1"), golden.bytes(f, "expected", fn="rfc7748_ladder_test1_vectorset2"),
           golden.bytes(f, "SHARED_SECRET"), golden.bytes(f, "SHARED_SECRET")]
    got = eng.x25519_batch(np.frombuffer(b"".join(ks), np.uint8).reshape(-1, 32), np.frombuffer(b"".join(us), np.uint8).reshape(-1, 32))
    assert [got[i].tobytes() for i in range(4)] == exp
    # the 7 low-order points -> all-zero output (constants.rs:208-222)
    lo = np.frombuffer(b"".join(golden.bytes("src/constants.rs", "X25519_LOW_ORDER_POINTS", i) for i in range(7)), np.uint8).reshape(7, 32)
    k7 = util.rand_bytes(31, 7)
    assert not eng.x25519_batch(k7, lo).any()
    # random k, u (about half of the u on the twist), u >= p and bit 255 set included
    n = 3000
    k = util.rand_bytes(32, n); u = util.rand_bytes(33, n)
    u[:8] = 0xFF
    got = eng.x25519_batch(k, u)
    want = orc.x25519_batch(k, u, threads=8)
    assert np.array_equal(got, want)
    assert eng.x25519_batch(np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8)).shape == (0, 32)


def test_x25519_full_size_2p20_diffie_hellman(eng, orc, torch):
    """BASELINE configs[4] at full size: 2^20 independent ladders.  Size-independent property: Diffie-Hellman
    commutes, x25519(a, x25519(b, 9)) == x25519(b, x25519(a, 9)) for every pair (x25519_tests.rs:13-31 does this
    for one pair); and every one of the 2^20 outputs against the oracle."""
    n = 1 << 20
    g = torch.Generator(device="cuda"); g.manual_seed(7748)
    da = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    db = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    nine = torch.zeros((n, 32), dtype=torch.uint8, device="cuda"); nine[:, 0] = 9
    pa, pb = eng.x25519_batch_t(da, nine), eng.x25519_batch_t(db, nine)
    sab, sba = eng.x25519_batch_t(da, pb), eng.x25519_batch_t(db, pa)
    assert torch.equal(sab, sba)
    assert int((sab != 0).any(dim=1).sum()) == n                       # no contributory failure on honest keys
    import os
    want = orc.x25519_batch(da.cpu().numpy(), pb.cpu().numpy(), threads=os.cpu_count() or 8)    # ALL 2^20 ladders
    assert np.array_equal(sab.cpu().numpy(), want)


def test_x25519_public_keys_through_the_fixed_base_path(eng, orc, golden, torch):
    """x25519-dalek derives public keys as EdwardsPoint::mul_base_clamped(secret).to_montgomery() (x25519.rs:105-109):
    c25519_x25519_base_batch must equal the ladder on u = 9 -- RFC 7748 6.1 keys, edge secrets, 2^16 random ones
    against the engine's own ladder and a slice against the oracle."""
    f = "x25519_tests.rs"
    ks = [golden.bytes(f, "ALICE_PRIVATE_KEY"), golden.bytes(f, "BOB_PRIVATE_KEY")]
    want = [golden.bytes(f, "ALICE_PUBLIC_KEY"), golden.bytes(f, "BOB_PUBLIC_KEY")]
    got = eng.x25519_base_batch(np.frombuffer(b"".join(ks), np.uint8).reshape(-1, 32))
    assert [got[i].tobytes() for i in range(2)] == want
    edge = np.zeros((4, 32), np.uint8); edge[1] = 0xFF; edge[2, 0] = 7; edge[3, 31] = 0x80      # clamping decides all of these
    nine = np.zeros((4, 32), np.uint8); nine[:, 0] = 9
    assert np.array_equal(eng.x25519_base_batch(edge), orc.x25519_batch(edge, nine, threads=2))
    n = 1 << 16
    g = torch.Generator(device="cuda"); g.manual_seed(90125)
    dk = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    d9 = torch.zeros((n, 32), dtype=torch.uint8, device="cuda"); d9[:, 0] = 9
    pub = eng.x25519_base_batch_t(dk)
    assert torch.equal(pub, eng.x25519_batch_t(dk, d9))
    idx = torch.arange(0, n, 97, device="cuda")
    assert np.array_equal(pub[idx].cpu().numpy(), orc.x25519_batch(dk[idx].cpu().numpy(), d9[idx].cpu().numpy(), threads=8))
    assert eng.x25519_base_batch(np.zeros((0, 32), np.uint8)).shape == (0, 32)


def test_decompress_compress_vs_oracle(eng, orc):
    n = 4000
    enc = util.rand_bytes(41, n)
    enc[0] = 0; enc[0, 0] = 1                      # identity
    enc[1] = np.frombuffer((2**255 - 19 + 1).to_bytes(32, "little"), np.uint8)   # non-canonical y = p+1
    enc[2] = enc[0]; enc[2, 31] |= 0x80            # "negative zero" x
    st, pts, ok = eng.decompress_batch(enc)
    want_ok = orc.ed_decompress_ok_batch(enc, threads=8)
    assert np.array_equal(ok, want_ok)
    assert st == (1 if (want_ok == 0).any() else 0)
    good = np.nonzero(ok)[0]
    assert 1000 < len(good) < 3000
    # decompressed points re-compress to what the oracle's decompress->compress gives
    comp = eng.compress_batch(pts[good])
    for j in list(range(50)) + [len(good) - 1]:
        i = good[j]
        assert comp[j].tobytes() == orc.ed_compress(orc.ed_decompress(enc[i].tobytes()))
    # batched (n >= 4096) and per-lane (n < 4096) compress kernels agree
    big = np.concatenate([pts[good]] * 4)[:6000]
    assert np.array_equal(eng.compress_batch(big)[:len(good)], comp[:min(len(good), 6000)])
    # all-valid batch -> status OK
    valid = orc.mul_base_compress_batch(util.rand_scalars(42, 300))
    st, pts2, ok2 = eng.decompress_batch(valid)
    assert st == 0 and ok2.all()
    assert np.array_equal(eng.compress_batch(pts2), valid)


def test_ristretto_vs_oracle(eng, orc, golden):
    f = "src/ristretto.rs"
    encs = np.frombuffer(b"".join(golden.bytes(f, "compressed", i) for i in range(16)), np.uint8).reshape(16, 32)
    st, pts, ok = eng.decompress_batch(encs, in_fmt=1)
    assert st == 0 and ok.all()
    assert np.array_equal(eng.compress_batch(pts, out_fmt=1), encs)
    # i*B computed by the engine, compressed as Ristretto, equals the sage table (ristretto.rs:1387)
    sc = np.zeros((16, 32), np.uint8); sc[:, 0] = np.arange(16)
    raw = eng.mul_base_batch(sc, out_fmt=2)
    assert np.array_equal(eng.compress_batch(raw, out_fmt=1), encs)
    # invalid encodings: negative s, non-canonical s, random junk -> flags match the oracle
    bad = util.rand_bytes(51, 500)
    bad[0] = np.frombuffer((2**255 - 20).to_bytes(32, "little"), np.uint8)
    bad[1] = np.frombuffer((2**255 - 19).to_bytes(32, "little"), np.uint8)
    st, _, ok = eng.decompress_batch(bad, in_fmt=1)
    want = np.array([orc.ris_decompress(bad[i].tobytes()) is not None for i in range(500)], dtype=np.uint8)
    assert np.array_equal(ok, want) and st == 1
    # random points: engine ristretto compress == oracle ristretto compress
    s = util.rand_scalars(52, 200)
    raw = eng.mul_base_batch(s, out_fmt=2)
    got = eng.compress_batch(raw, out_fmt=1)
    for i in range(200):
        assert got[i].tobytes() == orc.ris_compress(orc.ed_mul_base(s[i].tobytes()))
    # RistrettoBasepointTable * scalar in one call (out_fmt = 1): the sage table and the composed path
    assert np.array_equal(eng.mul_base_batch(sc, out_fmt=1), encs)
    assert np.array_equal(eng.mul_base_batch(s, out_fmt=1), got)


def test_to_montgomery_batch_vs_oracle(eng, orc):
    """edwards.rs:2339-2360 batch_to_montgomery: batched == one-at-a-time; identity -> 0"""
    s = util.rand_scalars(81, 5000)
    s[0] = 0                                   # the identity: Z - Y = 0, must give u = 0 without poisoning the batch
    s[17] = 0
    raw = eng.mul_base_batch(s, out_fmt=2)
    got = eng.to_montgomery_batch(raw)
    assert not got[0].any() and not got[17].any()
    for i in list(range(0, 5000, 97)) + [1, 16, 18, 4999]:
        assert got[i].tobytes() == orc.ed_to_montgomery(raw[i].tobytes()), i
    # and it agrees with the ladder from the basepoint u = 9 (montgomery.rs:628-641)
    nine = np.zeros((64, 32), np.uint8); nine[:, 0] = 9
    for i in range(1, 64):
        assert got[i].tobytes() == orc.mont_mul(nine[i].tobytes(), s[i].tobytes())
    assert eng.to_montgomery_batch(np.zeros((0, 160), np.uint8)).shape == (0, 32)
