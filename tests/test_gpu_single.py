"""GPU parity tests for the per-item paths:
variable-base batch (variable_base.rs), per-signature verify / verify_strict (verifying.rs:203-214,
:359-382 over vartime_double_base.rs) pinned by all 914 VALIDATIONVECTORS, and batched keygen/signing
(signing.rs:878-905) pinned byte-for-byte by the 128 deterministic TESTVECTORS signatures."""
import json
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = util.L


@pytest.fixture(scope="module")
def eng():
    import curve25519_dalek_amd as pkg
    return pkg.Engine(0)


def i2b(x):
    return int(x).to_bytes(32, "little")


def _testvectors():
    out = []
    with open(os.path.join(ROOT, "tests", "golden", "ed25519_testvectors.txt")) as fh:
        for line in fh:
            p = line.strip().split(":")
            if len(p) >= 4:
                out.append((bytes.fromhex(p[0])[:32], bytes.fromhex(p[1]), bytes.fromhex(p[2]), bytes.fromhex(p[3])[:64]))
    return out


def test_mul_batch_vs_oracle(eng, orc):
    n = 700
    s = util.rand_scalars(61, n)
    edge = util.edge_scalars()
    edge = edge[[int.from_bytes(e.tobytes(), "little") < 2**255 for e in edge]]
    s[:edge.shape[0]] = edge
    enc = util.rand_bytes(62, 4000)
    okmask = orc.ed_decompress_ok_batch(enc)
    enc = np.concatenate([enc[okmask == 1][:n - 5], enc[okmask == 0][:5]])      # last 5 do not decode
    got, ok = eng.mul_batch(s, enc, in_fmt=0, out_fmt=0)
    assert ok[:n - 5].all() and not ok[n - 5:].any()
    for i in list(range(60)) + list(range(n - 40, n - 5)):
        want = orc.ed_compress(orc.ed_mul(orc.ed_decompress(enc[i].tobytes()), s[i].tobytes()))
        assert got[i].tobytes() == want, i
    # raw in / raw out (points with Z != 1 from the fixed-base kernel)
    t = util.rand_scalars(63, 300)
    raw = eng.mul_base_batch(t, out_fmt=2)
    out, ok = eng.mul_batch(s[:300], raw, in_fmt=2, out_fmt=2)
    assert ok.all()
    for i in range(0, 300, 7):
        want = orc.ed_compress(orc.ed_mul(raw[i].tobytes(), s[i].tobytes()))
        assert orc.ed_compress(out[i].tobytes()) == want
    # s*(t*B) == (s*t mod l)*B for reduced s (the reference's own consistency check, edwards.rs:2146)
    st = np.frombuffer(b"".join(orc.sc_mul(orc.sc_reduce(s[i].tobytes()), t[i].tobytes()) for i in range(300)), np.uint8).reshape(-1, 32)
    red = np.frombuffer(b"".join(orc.sc_reduce(s[i].tobytes()) for i in range(300)), np.uint8).reshape(-1, 32)
    lhs, _ = eng.mul_batch(red, raw, in_fmt=2, out_fmt=0)
    assert np.array_equal(lhs, eng.mul_base_batch(st))
    assert eng.mul_batch(np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8), 0, 0)[0].shape == (0, 32)


def test_double_base_batch_vs_oracle(eng, orc, golden):
    """a*A + b*B against vartime_double_base.rs:23-72 (oracle restatement) and the reference's own
    A_TIMES_BASEPOINT / DOUBLE_SCALAR_MULT_RESULT constants (edwards.rs:1797-1818)."""
    import curve25519_dalek_amd.dalek as dalek
    n = 400
    a = util.rand_scalars(71, n); b = util.rand_scalars(72, n)
    edge = util.edge_scalars()
    edge = edge[[int.from_bytes(e.tobytes(), "little") < 2**255 for e in edge]]
    a[:edge.shape[0]] = edge; b[:edge.shape[0]] = edge[::-1]
    A = eng.mul_base_batch(util.rand_scalars(73, n), out_fmt=2)                 # raw points with Z != 1
    out, ok = eng.double_base_batch(a, A, b, in_fmt=2, out_fmt=0)
    assert ok.all()
    for i in list(range(0, 40)) + list(range(40, n, 9)):
        want = orc.ed_compress(orc.ed_double_scalar_mul_basepoint(a[i].tobytes(), A[i].tobytes(), b[i].tobytes()))
        assert out[i].tobytes() == want, i
    raw, _ = eng.double_base_batch(a[:64], A[:64], b[:64], in_fmt=2, out_fmt=2)
    assert [orc.ed_compress(raw[i].tobytes()) for i in range(64)] == [out[i].tobytes() for i in range(64)]
    # compressed inputs, one that does not decode
    enc = eng.compress_batch(A[:50])
    bad = util.rand_bytes(74, 4000); bad = bad[orc.ed_decompress_ok_batch(bad) == 0][:1]
    enc = np.concatenate([enc, bad])
    got, ok = eng.double_base_batch(a[:51], enc, b[:51], in_fmt=0, out_fmt=0)
    assert ok[:50].all() and not ok[50]
    assert [got[i].tobytes() for i in range(50)] == [out[i].tobytes() for i in range(50)]
    # reference constants: A_SCALAR * (A_TIMES_BASEPOINT) + B_SCALAR * B = DOUBLE_SCALAR_MULT_RESULT
    f = "curve25519-dalek/src/edwards.rs"
    asc = golden.bytes(f, "A_SCALAR"); bsc = golden.bytes(f, "B_SCALAR")
    atb = golden.bytes(f, "A_TIMES_BASEPOINT"); res = golden.bytes(f, "DOUBLE_SCALAR_MULT_RESULT")
    P = orc.ed_decompress(bytes(atb))
    assert dalek.EdwardsPoint.vartime_double_scalar_mul_basepoint([bytes(asc)], [P], [bytes(bsc)], engine=eng) == [bytes(res)]
    assert eng.double_base_batch(np.zeros((0, 32), np.uint8), np.zeros((0, 160), np.uint8), np.zeros((0, 32), np.uint8))[0].shape[0] == 0


def test_verify_each_testvectors(eng, orc):
    tv = _testvectors()
    pks, msgs, sigs = [t[1] for t in tv], [t[2] for t in tv], [t[3] for t in tv]
    for strict in (False, True):
        st = eng.verify_each(msgs, sigs, pks, strict)
        assert not st.any()
    bad = list(sigs)
    for j in (0, 17, 127):
        b = bytearray(bad[j]); b[2] ^= 0x40; bad[j] = bytes(b)
    bad[5] = sigs[5][:32] + i2b(int.from_bytes(sigs[5][32:], "little") + L)       # non-canonical s
    pk2 = list(pks); pk2[9] = i2b(2)                                               # key does not decode
    st = eng.verify_each(msgs, bad, pk2)
    want = np.array([orc.ed25519_verify(pk2[i], msgs[i], bad[i]) for i in range(len(tv))], dtype=np.uint8)
    assert np.array_equal(st, want)
    assert st[0] == 3 and st[17] == 3 and st[127] == 3 and st[5] == 2 and st[9] == 1 and st[1] == 0
    assert eng.verify_each([], [], []).shape == (0,)


def test_verify_each_validation_vectors(eng, orc):
    """validation_criteria.rs:134-170 on the GPU: all 914 C2SP vectors in ONE call per mode; accept iff
    the vector's flags are a subset of the reference's allowed sets (:8-23), and status == oracle."""
    allowed = {"low_order_A", "low_order_R", "non_canonical_A", "low_order_component_A", "low_order_component_R", "reencoded_k"}
    allowed_strict = {"low_order_component_A", "low_order_component_R"}
    with open(os.path.join(ROOT, "tests", "golden", "ed25519_validation.json")) as fh:
        vv = json.load(fh)
    pks = [bytes.fromhex(v["key"]) for v in vv]; sigs = [bytes.fromhex(v["sig"]) for v in vv]; msgs = [v["msg"].encode() for v in vv]
    st = eng.verify_each(msgs, sigs, pks, False)
    sts = eng.verify_each(msgs, sigs, pks, True)
    for i, v in enumerate(vv):
        flags = set(v["flags"])
        assert (st[i] == 0) == flags.issubset(allowed), (v["number"], flags, st[i])
        assert (sts[i] == 0) == flags.issubset(allowed_strict), (v["number"], flags, sts[i])
    want = np.array([orc.ed25519_verify(pks[i], msgs[i], sigs[i]) for i in range(len(vv))], dtype=np.uint8)
    wants = np.array([orc.ed25519_verify_strict(pks[i], msgs[i], sigs[i]) for i in range(len(vv))], dtype=np.uint8)
    assert np.array_equal(st, want) and np.array_equal(sts, wants)


def test_keygen_sign_testvectors(eng, orc):
    """deterministic RFC 8032 signatures: byte-equal to TESTVECTORS (tests/ed25519.rs:82-85)"""
    tv = _testvectors()
    seeds, pks, msgs, sigs = [t[0] for t in tv], [t[1] for t in tv], [t[2] for t in tv], [t[3] for t in tv]
    gpk, gsig = eng.sign_batch(seeds, msgs)
    assert [gpk[i].tobytes() for i in range(len(tv))] == pks
    assert [gsig[i].tobytes() for i in range(len(tv))] == sigs
    import torch
    dpk = eng.keygen_batch_t(torch.from_numpy(np.frombuffer(b"".join(seeds), np.uint8).reshape(-1, 32).copy()).cuda())
    assert [dpk[i].cpu().numpy().tobytes() for i in range(len(tv))] == pks


def test_sign_then_verify_roundtrip_large(eng, orc):
    import torch
    n = 1 << 16
    seeds = util.rand_bytes(71, n); msgs = util.rand_bytes(72, n, 59)
    dseed = torch.from_numpy(seeds).cuda(); dmsg = torch.from_numpy(msgs.reshape(-1)).cuda()
    doff = torch.arange(0, 59 * (n + 1), 59, dtype=torch.int64).cuda()
    dpk, dsig = eng.sign_batch_t(dseed, dmsg, doff)
    idx = np.random.default_rng(3).choice(n, 200, replace=False)
    pk_h, sig_h = dpk.cpu().numpy(), dsig.cpu().numpy()
    wpk, wsig = orc.ed25519_keygen_sign_batch(seeds[idx], msgs[idx], threads=os.cpu_count() or 1)
    assert np.array_equal(pk_h[idx], wpk) and np.array_equal(sig_h[idx], wsig)
    st = eng.verify_each_t(dmsg, doff, dsig, dpk, True)
    assert not st.any()
    print("verify_each 2^16 strict: %.3f ms (var-base %.3f ms)" % (eng.last_kernel_ms(), eng.phase_ms(0, 0)))
    assert eng.verify_batch_t(dmsg, doff, dsig, dpk, 1) == 0
    dsig[4242, 7] ^= 1
    st = eng.verify_each_t(dmsg, doff, dsig, dpk, False).cpu().numpy()
    assert st[4242] == 3 and st.sum() == 3            # exactly the tampered one is located
    assert eng.verify_batch_t(dmsg, doff, dsig, dpk, 1) == 3


# ---- Ed25519ph / Ed25519ctx (RFC 8032 5.1 with dom2): verify_prehashed[_strict] / sign_prehashed -----------------------------------
def test_ed25519ph_rfc8032_vector(eng, orc):
    """The reference's own prehash test (ed25519-dalek/tests/ed25519.rs:104-146, RFC 8032 7.3): sign_prehashed(SHA-512("abc"), None) reproduces the
    vector byte for byte, verify_prehashed and verify_prehashed_strict accept it, and the signature is bound to the context, the prehash and the
    variant (plain Ed25519 rejects it) -- through the host-pointer entry points, the device-pointer ones and the dalek front end."""
    import hashlib, torch
    from test_oracle_kat import ED25519PH_SK as SK, ED25519PH_PK as PK, ED25519PH_MSG as MSG, ED25519PH_SIG as SIG
    import curve25519_dalek_amd.dalek as dalek
    ph = hashlib.sha512(MSG).digest()
    pks, sigs = eng.sign_batch_prehashed([SK], [ph])
    assert pks[0].tobytes() == PK and sigs[0].tobytes() == SIG
    for strict in (False, True):
        st = eng.verify_each_prehashed([ph, ph, hashlib.sha512(b"abd").digest()], [SIG, SIG[:7] + bytes([SIG[7] ^ 1]) + SIG[8:], SIG], [PK, PK, PK], strict=strict)
        assert list(st) == [0, 3, 3]
        assert list(eng.verify_each_prehashed([ph], [SIG], [PK], context=b"edtest", strict=strict)) == [3]        # context mismatch
    assert list(eng.verify_each([MSG, ph], [SIG, SIG], [PK, PK])) == [3, 3]                                       # not a plain Ed25519 signature
    # device-pointer forms
    dph = torch.from_numpy(np.frombuffer(ph, np.uint8).copy()).cuda(); dsk = torch.from_numpy(np.frombuffer(SK, np.uint8).copy()).cuda()
    dpk, dsig = eng.sign_batch_prehashed_t(dsk, dph)
    assert dpk.cpu().numpy().tobytes() == PK and dsig.cpu().numpy().tobytes() == SIG
    assert eng.verify_each_prehashed_t(dph, dsig, dpk, strict=True).cpu().numpy().tolist() == [0]
    assert eng.verify_each_prehashed_t(dph, dsig, dpk, context=b"x").cpu().numpy().tolist() == [3]
    # the front end takes the digest STATE, as the reference does
    vk, sg = dalek.sign_batch_prehashed([SK], [hashlib.sha512(MSG)], engine=eng)
    assert vk == [PK] and sg == [SIG]
    assert dalek.verify_each_prehashed([hashlib.sha512(MSG)], [SIG], [PK], strict=True, engine=eng) == [None]
    err = dalek.verify_each_prehashed([hashlib.sha512(MSG)], [SIG], [PK], context=b"edtest", engine=eng)[0]
    assert isinstance(err, dalek.SignatureError)
    # a context beyond 255 octets: InternalError::PrehashedContextLength (signing.rs:931-933), on every entry point
    assert eng.sign_batch_prehashed([SK], [ph], context=bytes(256)) == 5 and eng.verify_each_prehashed([ph], [SIG], [PK], context=bytes(256)) == 5
    with pytest.raises(dalek.SignatureError):
        dalek.sign_batch_prehashed([SK], [ph], context=bytes(300), engine=eng)


def test_ed25519ph_contexts_and_batches_vs_oracle(eng, orc):
    """Random keys, prehashes and contexts of every alignment class (the dom2 prefix is 34 + len(ctx) bytes: R, A and the prehash are absorbed at
    any byte position of the SHA-512 block, across one, two and three block boundaries): signatures byte-equal to the oracle's, verdicts equal to the
    oracle's for valid, tampered and cross-context pairs, plain and strict; a batch large enough for several blocks of the grid."""
    import hashlib
    rng = np.random.default_rng(2024)
    for ctx in (b"", b"e", b"edtest", bytes(range(30)), bytes(range(94)), bytes(range(95)), bytes(range(200)), bytes(range(255))):
        n = 37
        sks = [rng.bytes(32) for _ in range(n)]
        phs = [hashlib.sha512(rng.bytes(int(rng.integers(0, 90)))).digest() for _ in range(n)]
        pks, sigs = eng.sign_batch_prehashed(sks, phs, context=ctx)
        for i in range(n):
            st, want = orc.ed25519_sign_prehashed(sks[i], phs[i], ctx)
            assert st == 0 and sigs[i].tobytes() == want and pks[i].tobytes() == orc.ed25519_pubkey(sks[i]), (len(ctx), i)
        S = [sigs[i].tobytes() for i in range(n)]; P = [pks[i].tobytes() for i in range(n)]
        bad = list(S); bad[5] = bad[5][:40] + bytes([bad[5][40] ^ 4]) + bad[5][41:]; bad[9] = S[10]
        for strict in (False, True):
            assert not eng.verify_each_prehashed(phs, S, P, context=ctx, strict=strict).any()
            got = eng.verify_each_prehashed(phs, bad, P, context=ctx, strict=strict)
            want = [orc.ed25519_verify_prehashed(P[i], phs[i], bad[i], context=ctx, strict=strict) for i in range(n)]
            assert list(got) == want and got[5] != 0 and got[9] != 0
            other = ctx[:-1] + b"\xff" if ctx else b"\x00"
            assert eng.verify_each_prehashed(phs, S, P, context=other, strict=strict).all()
    n = 5000
    sks = util.rand_bytes(81, n); phs = util.rand_bytes(82, n, 64)
    pks, sigs = eng.sign_batch_prehashed([r.tobytes() for r in sks], [r.tobytes() for r in phs], context=b"batch")
    for i in (0, 1, 255, 256, 4095, 4999):
        assert sigs[i].tobytes() == orc.ed25519_sign_prehashed(sks[i].tobytes(), phs[i].tobytes(), b"batch")[1]
    st = eng.verify_each_prehashed([r.tobytes() for r in phs], [r.tobytes() for r in sigs], [r.tobytes() for r in pks], context=b"batch", strict=True)
    assert not st.any()


def test_ed25519ph_repudiation_weak_key(eng, orc):
    """ed25519-dalek/tests/ed25519.rs:248-292 (repudiation_prehash): for the order-2 public key (EIGHT_TORSION[4]) a signature R = sB - A, s can be found
    that verify_prehashed accepts for TWO messages under the context "edtest", and verify_prehashed_strict rejects both (small-order key)."""
    import hashlib
    A = bytes([236] + [255] * 30 + [127])
    m1, m2 = hashlib.sha512(b"Send 100 USD to Alice").digest(), hashlib.sha512(b"Send 100000 USD to Alice").digest()
    ctx = b"edtest"
    Apt = orc.ed_decompress(A)
    assert Apt is not None
    rng = np.random.default_rng(77)
    found = None
    for _ in range(400):
        s = (int.from_bytes(rng.bytes(32), "little") % (util.L - 1) + 1).to_bytes(32, "little")
        R = orc.ed_compress(orc.ed_add(orc.ed_mul_base(s), orc.ed_neg(Apt)))
        sig = R + s
        if orc.ed25519_verify_prehashed(A, m1, sig, context=ctx) == 0 and orc.ed25519_verify_prehashed(A, m2, sig, context=ctx) == 0:
            found = sig
            break
    assert found is not None
    assert list(eng.verify_each_prehashed([m1, m2], [found, found], [A, A], context=ctx)) == [0, 0]
    assert list(eng.verify_each_prehashed([m1, m2], [found, found], [A, A], context=ctx, strict=True)) == [3, 3]
    assert [orc.ed25519_verify_prehashed(A, m, found, context=ctx, strict=True) for m in (m1, m2)] == [3, 3]
