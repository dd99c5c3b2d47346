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
