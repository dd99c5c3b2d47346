"""GPU parity tests of the host-pointer (FFI) entry points -- the interface a Rust caller binds -- and of the entry points
added in round 3: chunked / overlapped staging must give byte for byte what the device-pointer calls give at the full
BASELINE sizes (2^20), for ragged sizes around the chunk boundaries, with caller buffers from c25519_host_alloc, after
c25519_ctx_trim; constant-time fixed-base tables for a caller's point (EdwardsBasepointTable::create, edwards.rs:1131-1141;
RistrettoBasepointTable::create, ristretto.rs:1080-1110); mul_clamped / mul_base_clamped (edwards.rs:932-956);
SharedSecret::was_contributory (x25519.rs:335); the ONE transcript of the multi-context strict z-mode."""
import ctypes as C

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


def clamp(b):
    a = bytearray(b); a[0] &= 248; a[31] &= 127; a[31] |= 64
    return bytes(a)


@pytest.mark.parametrize("n", [0, 1, 1023, (1 << 18) + 1, (1 << 20), (1 << 20) + 12345])
def test_mul_base_host_twin_equals_device_call(eng, n):
    """c25519_mul_base_batch through the chunked pipeline == c25519_mul_base_batch_dev on the whole batch (all three output formats)"""
    import torch
    s = util.rand_scalars(400 + (n % 97), n)
    d = torch.from_numpy(s).cuda() if n else torch.zeros((0, 32), dtype=torch.uint8, device="cuda")
    for fmt in (0, 1, 2):
        want = eng.mul_base_batch_t(d, fmt).cpu().numpy()
        got = eng.mul_base_batch(s, fmt)
        assert got.shape == want.shape and np.array_equal(got, want), (n, fmt)
    ms, up, down = eng.last_ffi()
    assert up == n * 32 and down == n * 160 and (n == 0 or ms > 0)


def test_x25519_host_twin_full_size_and_contributory(eng, orc):
    import torch
    n = (1 << 20) + 77
    k = util.rand_bytes(511, n); u = util.rand_bytes(512, n)
    low = [0, 1, 325606250916557431795983626356110631294008115727848805560023387167927233504, 39382357235489614581723060781553021112529911719440698176882885853963445705823,
           2**255 - 19 - 1, 2**255 - 19, 2**255 - 19 + 1]                      # the 7 low-order u of constants.rs:98-123
    for j, v in enumerate(low):
        u[j * 1000] = np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
    want = eng.x25519_batch_t(torch.from_numpy(k).cuda(), torch.from_numpy(u).cuda()).cpu().numpy()
    out, fl = eng.x25519_contributory_batch(k, u)
    assert np.array_equal(out, want) and np.array_equal(eng.x25519_batch(k, u), want)
    idx = np.random.default_rng(1).integers(0, n, 200)
    assert np.array_equal(out[idx], orc.x25519_batch(k[idx], u[idx], threads=2))
    zero = ~out.any(axis=1)
    assert np.array_equal(fl, (~zero).astype(np.uint8))
    assert all(zero[j * 1000] for j in range(len(low))) and zero.sum() == len(low)      # was_contributory is false exactly there


def test_host_buffers_from_host_alloc_and_reuse(eng):
    """input and output buffers in page-locked memory from c25519_host_alloc; the same output buffer reused across calls"""
    lib = eng.lib
    n = (1 << 19) + 5
    pin = lib.c25519_host_alloc(n * 32); pout = lib.c25519_host_alloc(n * 32)
    assert pin and pout
    try:
        a_in = np.ctypeslib.as_array(C.cast(pin, C.POINTER(C.c_uint8)), shape=(n, 32))
        a_out = np.ctypeslib.as_array(C.cast(pout, C.POINTER(C.c_uint8)), shape=(n, 32))
        a_in[:] = util.rand_scalars(77, n)
        want = eng.mul_base_batch(a_in.copy())
        for _ in range(3):
            a_out[:] = 0
            got = eng.mul_base_batch(a_in, out=a_out)
            assert got is a_out and np.array_equal(a_out, want)
    finally:
        lib.c25519_host_free(pin); lib.c25519_host_free(pout)


def test_codec_host_twins_ragged(eng, orc):
    """decompress / compress / to_montgomery / double_and_compress / mul_batch / double_base through the pipeline on a size that is
    not a multiple of the chunk, with an invalid encoding in the last chunk"""
    n = (1 << 18) + 4099
    s = util.rand_scalars(31, n)
    enc = eng.mul_base_batch(s, 0)
    enc[n - 2] = np.frombuffer((2).to_bytes(32, "little"), np.uint8)                # y = 2: not on the curve
    st, pts, ok = eng.decompress_batch(enc)
    assert st == 1 and ok[n - 2] == 0 and ok.sum() == n - 1
    good = np.ones(n, bool); good[n - 2] = False
    idx = np.random.default_rng(3).integers(0, n - 2, 64)
    assert np.array_equal(eng.compress_batch(pts)[good], enc[good])
    for i in idx:
        assert eng.to_montgomery_batch(pts[i:i + 1])[0].tobytes() == orc.ed_to_montgomery(pts[i].tobytes())
    mont = eng.to_montgomery_batch(pts)
    assert all(mont[i].tobytes() == orc.ed_to_montgomery(pts[i].tobytes()) for i in idx[:16])
    dc = eng.double_and_compress_batch(pts[good])
    gi = np.flatnonzero(good)
    for j in np.random.default_rng(4).integers(0, len(gi), 16):
        assert dc[j].tobytes() == orc.ris_compress(orc.ed_double(pts[gi[j]].tobytes()))
    t = util.rand_scalars(32, n)
    prod, ok2 = eng.mul_batch(t, pts, in_fmt=2, out_fmt=0)
    for i in idx[:16]:
        assert prod[i].tobytes() == orc.ed_compress(orc.ed_mul(pts[i].tobytes(), t[i].tobytes()))
    db, ok3 = eng.double_base_batch(t, pts, s, in_fmt=2, out_fmt=0)
    for i in idx[:8]:
        assert db[i].tobytes() == orc.ed_compress(orc.ed_double_scalar_mul_basepoint(t[i].tobytes(), pts[i].tobytes(), s[i].tobytes()))


def test_msm_host_twin_many_passes(eng, orc):
    """c25519_msm_vartime on host pointers cuts n >= 2^20 raw terms into 2^19-term passes whose inputs travel while the previous
    pass computes: same point as the device-pointer call and as (sum x_i y_i) B; compressed points, an invalid encoding -> NONE"""
    import torch
    n = (1 << 21) + 333
    x = util.rand_scalars(81, n); y = util.rand_scalars(82, n)
    dy = torch.from_numpy(y).cuda()
    dpts = eng.mul_base_batch_vartime_t(dy, 2)
    pts = dpts.cpu().numpy()
    st_d, want = eng.msm_vartime_t(torch.from_numpy(x).cuda(), dpts, 2, 0)
    st, got = eng.msm_vartime(x, pts, in_fmt=2, out_fmt=0)
    xi = [int.from_bytes(r.tobytes(), "little") for r in x[:5000]]
    assert st == 0 and st_d == 0 and got == want
    acc = sum(int.from_bytes(x[i].tobytes(), "little") * int.from_bytes(y[i].tobytes(), "little") for i in range(n)) % L
    assert got == orc.ed_compress(orc.ed_mul_base(i2b(acc)))
    ms, up, down = eng.last_ffi()
    assert up == n * 192
    enc = eng.compress_batch_t(dpts).cpu().numpy()
    st, got_c = eng.msm_vartime(x, enc, in_fmt=0, out_fmt=0)
    assert st == 0 and got_c == want
    enc[n - 7] = np.frombuffer((2).to_bytes(32, "little"), np.uint8)
    st, none = eng.msm_vartime(x, enc, in_fmt=0, out_fmt=0)
    assert st == 1


@pytest.mark.parametrize("n", [3000, 9001, 16384, 16385])
def test_verify_batch_host_twin_both_modes(eng, orc, n):
    """ed25519_verify_batch[_keys] on host pointers (staged uploads in the device z-mode; everything up front in the strict
    mode): honest / forged / non-canonical s, with and without the keys' points, single pass and several passes.  Sizes: the one-copy staged upload with the
    small path's MSM range of rounds 5 - 6 (3000), with the mid path behind it (9001, 16 384: staged since late round 6), and the first size of the general route (16 385)"""
    import curve25519_dalek_amd as pkg
    E = pkg.engine
    seeds = util.rand_bytes(901, n); msgs = [bytes(util.rand_bytes(902 + i, 1, 1 + (i % 50))[0]) for i in range(n)]
    pks, sigs = eng.sign_batch(rows(seeds), msgs)
    P, S = rows(pks), rows(sigs)
    _, pkpts, ok = eng.decompress_batch(pks)
    assert ok.all()
    forged = list(S); b = bytearray(forged[n - 5]); b[3] ^= 1; forged[n - 5] = bytes(b)
    nonc = list(forged); b = bytearray(nonc[7]); b[63] |= 0xE0; nonc[7] = bytes(b)
    for z_mode in (E.Z_TRANSCRIPT, E.Z_DEVICE):
        for pp in (None, pkpts):
            assert eng.verify_batch(msgs, S, P, z_mode, pk_points=pp) == 0
            assert eng.verify_batch(msgs, forged, P, z_mode, pk_points=pp) == 3
            assert eng.verify_batch(msgs, nonc, P, z_mode, pk_points=pp) == 2
    assert orc.ed25519_verify_batch(msgs, S, P) == 0 and orc.ed25519_verify_batch(msgs, forged, P) == 3


def test_verify_batch_host_twin_multi_pass_fresh_process():
    """2^15-signature passes (C25519_VERIFY_PASS_LOG2, read once per process): the staged uploads slice the message blob and the
    offsets per pass; verdicts as in one pass"""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import util, curve25519_dalek_amd as pkg
        E = pkg.engine
        eng = pkg.Engine(0)
        n = 3 * 32768 + 4321
        seeds = util.rand_bytes(5, n); blob = util.rand_bytes(6, n, 9)
        msgs = [blob[i, :1 + (i %% 9)].tobytes() for i in range(n)]
        pks, sigs = eng.sign_batch([seeds[i].tobytes() for i in range(n)], msgs)
        P = [pks[i].tobytes() for i in range(n)]; S = [sigs[i].tobytes() for i in range(n)]
        bad = list(S); b = bytearray(bad[n - 9]); b[40] ^= 2; bad[n - 9] = bytes(b)
        for z in (E.Z_DEVICE, E.Z_TRANSCRIPT):
            assert eng.verify_batch(msgs, S, P, z) == 0, z
            assert eng.verify_batch(msgs, bad, P, z) == 3, z
        print("ok")
    """) % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    e = util.tune_env(C25519_VERIFY_PASS_LOG2="15")
    r = subprocess.run(util.child_argv(code), env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-3000:])


def test_basetable_for_caller_points(eng, orc):
    """EdwardsBasepointTable::create(&P) / RistrettoBasepointTable::create for three caller points -- a random one, one with a
    torsion component (P + T8) and the basepoint itself -- then 2^16 secret scalars each (incl. 0, 1, l - 1, unreduced
    2^255 - 1) against the oracle's variable-base ed_mul; Edwards, Ristretto and raw outputs; device and host twins"""
    import torch
    n = 1 << 16
    s = util.rand_scalars(61, n)
    edge = util.edge_scalars()
    s[:edge.shape[0]] = edge
    P1 = orc.ed_mul_base(i2b(0x1234567890abcdef ** 3 % L))
    T8 = orc.ed_decompress(bytes.fromhex("26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc05"))      # a point of order 8
    assert T8 is not None and orc.ed_is_small_order(T8)
    P2 = orc.ed_add(P1, T8)
    assert not orc.ed_is_torsion_free(P2)
    idx = np.concatenate([np.arange(edge.shape[0]), np.random.default_rng(9).integers(0, n, 96)])
    for P, fmt_in in ((P1, 2), (P2, 2), (orc.ed_compress(P1), 0), (orc.ed_basepoint(), 2)):
        h = eng.basetable_create(P, fmt_in)
        Praw = P if fmt_in == 2 else orc.ed_decompress(P)
        got = eng.mul_table_batch(h, s, 0)
        got_t = eng.mul_table_batch_t(h, torch.from_numpy(s).cuda(), 0).cpu().numpy()
        raw = eng.mul_table_batch(h, s, 2)
        ris = eng.mul_table_batch(h, s, 1)
        assert np.array_equal(got, got_t)
        for i in idx:
            want = orc.ed_mul(Praw, s[i].tobytes())
            assert got[i].tobytes() == orc.ed_compress(want), i
            assert orc.ed_compress(raw[i].tobytes()) == orc.ed_compress(want)
            assert ris[i].tobytes() == orc.ris_compress(want)
        eng.basetable_destroy(h)
    hb = eng.basetable_create(orc.ed_basepoint(), 2)
    assert np.array_equal(eng.mul_table_batch(hb, s, 0), eng.mul_base_batch(s, 0))              # the basepoint's table = mul_base
    eng.basetable_destroy(hb)
    ris_enc = orc.ris_compress(P1)
    hr = eng.basetable_create(ris_enc, 1)                                                        # RistrettoBasepointTable::create
    out = eng.mul_table_batch(hr, s[:64], 1)
    Pd = orc.ris_decompress(ris_enc)
    assert all(out[i].tobytes() == orc.ris_compress(orc.ed_mul(Pd, s[i].tobytes())) for i in range(64))
    eng.basetable_destroy(hr)
    import curve25519_dalek_amd as pkg
    with pytest.raises(pkg.EngineError):
        eng.basetable_create((2).to_bytes(32, "little"), 0)                                      # does not decode


def test_mul_clamped_entry_points(eng, orc):
    """EdwardsPoint::mul_base_clamped / mul_clamped (edwards.rs:932-956): clamp_integer(bytes), NOT reduced mod l"""
    n = 4099
    raw = util.rand_bytes(71, n)
    raw[0] = 0xFF; raw[1] = 0
    got = eng.mul_base_clamped_batch(raw, 0)
    pts = eng.mul_base_batch(util.rand_scalars(72, n), 2)
    prod, ok = eng.mul_clamped_batch(raw, pts, in_fmt=2, out_fmt=0)
    assert ok.all()
    for i in list(range(8)) + [n - 1]:
        c = clamp(raw[i].tobytes())
        assert got[i].tobytes() == orc.ed_compress(orc.ed_mul_base(c))
        assert prod[i].tobytes() == orc.ed_compress(orc.ed_mul(pts[i].tobytes(), c))
    # x25519 public keys are the Montgomery form of mul_base_clamped (x25519.rs:105-109)
    assert np.array_equal(eng.x25519_base_batch(raw), eng.to_montgomery_batch(eng.mul_base_clamped_batch(raw, 2)))


def test_verify_batch_multi_one_transcript(eng, orc):
    """ed25519_verify_batch_multi over three contexts on one GPU: strict z-mode = ONE transcript over the whole batch and one
    equation (verdicts as the single context; a forgery that needs z_1 = z_2 across the shard boundary is rejected), device
    z-mode = shard verdicts in precedence order"""
    import curve25519_dalek_amd as pkg
    E = pkg.engine
    engs = [eng, pkg.Engine(0), pkg.Engine(0)]
    n = 1000
    seeds = util.rand_bytes(31, n); msgs = [bytes(util.rand_bytes(32 + i, 1, 5 + (i % 7))[0]) for i in range(n)]
    pks, sigs = eng.sign_batch(rows(seeds), msgs)
    P, S = rows(pks), rows(sigs)
    bad = list(S); b = bytearray(bad[n // 3 + 1]); b[1] ^= 8; bad[n // 3 + 1] = bytes(b)
    nonc = list(bad); b = bytearray(nonc[n - 1]); b[63] |= 0x80; nonc[n - 1] = bytes(b)
    for z in (E.Z_TRANSCRIPT, E.Z_DEVICE):
        assert E.verify_batch_multi(engs, msgs, S, P, z) == 0
        assert E.verify_batch_multi(engs, msgs, bad, P, z) == 3
        assert E.verify_batch_multi(engs, msgs, nonc, P, z) == 2
        assert E.verify_batch_multi(engs[:1], msgs, S, P, z) == 0
    # cancellation forgery across the shard boundary (R_a + T, R_b - T): passes iff z_a == z_b -- shards with their own
    # transcripts would still reject it, but so must the single transcript; what distinguishes the single transcript is that
    # the verdict equals the reference's on the WHOLE batch: compare with the oracle's verify_batch on all n signatures
    a, bq = n // 3 - 1, n // 3                                   # last signature of shard 0, first of shard 1
    T = orc.ed_mul_base(i2b(987654321))
    Ra = orc.ed_decompress(S[a][:32]); Rb = orc.ed_decompress(S[bq][:32])
    fs = list(S)
    fs[a] = orc.ed_compress(orc.ed_add(Ra, T)) + S[a][32:]
    fs[bq] = orc.ed_compress(orc.ed_sub(Rb, T)) + S[bq][32:]
    assert orc.ed25519_verify_batch(msgs, fs, P) == 3
    assert E.verify_batch_multi(engs, msgs, fs, P, E.Z_TRANSCRIPT) == 3
    for e in engs[1:]:
        e.close()


def test_trim_then_reuse(eng):
    s = util.rand_scalars(5, 5000)
    a = eng.mul_base_batch(s)
    eng.trim()
    assert np.array_equal(eng.mul_base_batch(s), a)
    pts = eng.mul_base_batch(s, 2)
    st, r1 = eng.msm_vartime(s, pts, 2, 0)
    eng.trim()
    st2, r2 = eng.msm_vartime(s, pts, 2, 0)
    assert st == 0 and st2 == 0 and r1 == r2


@pytest.mark.parametrize("n", [1, 127, 129, 32768, 32769, 65536, 65537, 131072, 131073, 200000])
def test_constant_time_fixed_base_small_batches(eng, orc, n):
    """every launch shape of the constant-time fixed-base path (k_mul_base_ct_split with 256 / 512 / 1024-thread blocks for small
    batches: a scalar's windows split between two threads; k_mul_base<5, CT> beyond) against the radix-2^16 tables (an independent
    algorithm) on all outputs and against the oracle on the edge scalars, in the three output formats"""
    import torch
    s = util.rand_scalars(1000 + n % 977, n)
    edge = util.edge_scalars()
    k = min(n, edge.shape[0])
    s[:k] = edge[:k]
    d = torch.from_numpy(s).cuda()
    for fmt in (0, 2, 1):
        got = eng.mul_base_batch_t(d, fmt).cpu().numpy()
        want = eng.mul_base_batch_vartime_t(d, fmt).cpu().numpy()
        if fmt == 2:
            assert np.array_equal(eng.compress_batch(got[:4096]), eng.compress_batch(want[:4096]))
        else:
            assert np.array_equal(got, want), (n, fmt)
    got = eng.mul_base_batch_t(d, 0).cpu().numpy()
    for i in list(range(k)) + [n - 1]:
        assert got[i].tobytes() == orc.ed_compress(orc.ed_mul_base(s[i].tobytes())), i


def test_reference_named_front_end_round3(eng, orc):
    """dalek.py names added in round 3: EdwardsBasepointTable::create / RistrettoBasepointTable::create, mul_base_clamped,
    mul_clamped, diffie_hellman + SharedSecret::was_contributory -- against the oracle"""
    from curve25519_dalek_amd import dalek
    P = orc.ed_mul_base(i2b(424242))
    s = [util.rand_scalars(5, 9)[i].tobytes() for i in range(9)]
    t = dalek.EdwardsBasepointTable.create(orc.ed_compress(P), engine=eng)
    assert t.mul_base(s) == [orc.ed_compress(orc.ed_mul(P, x)) for x in s] and t.basepoint() == orc.ed_compress(P)
    t.close()
    r = dalek.RistrettoBasepointTable.create(orc.ris_compress(P), engine=eng)
    Pd = orc.ris_decompress(orc.ris_compress(P))
    assert r.mul_base(s) == [orc.ris_compress(orc.ed_mul(Pd, x)) for x in s]
    r.close()
    raw = [util.rand_bytes(6, 9)[i].tobytes() for i in range(9)]
    assert dalek.EdwardsPoint.mul_base_clamped(raw, engine=eng) == [orc.ed_compress(orc.ed_mul_base(clamp(b))) for b in raw]
    pts = [orc.ed_compress(orc.ed_mul_base(x)) for x in s]
    got = dalek.EdwardsPoint.mul_clamped(pts + [(2).to_bytes(32, "little")], raw + [raw[0]], engine=eng)
    assert got[:9] == [orc.ed_compress(orc.ed_mul(orc.ed_decompress(p), clamp(b))) for p, b in zip(pts, raw)] and got[9] is None
    ss = dalek.diffie_hellman(raw, [orc.x25519(b, (9).to_bytes(32, "little")) for b in raw[:8]] + [bytes(32)], engine=eng)
    assert [x.was_contributory() for x in ss] == [True] * 8 + [False] and ss[8].as_bytes() == bytes(32)
    assert ss[0].as_bytes() == orc.x25519(raw[0], orc.x25519(raw[0], (9).to_bytes(32, "little")))


def test_point_order_checks(eng, orc, golden):
    """c25519_point_order_checks_batch[_dev]: EdwardsPoint::is_small_order / is_torsion_free (edwards.rs:1405 / :1435) and
    VerifyingKey::is_weak on the eight torsion points (multiples of the reference's EIGHT_TORSION generator), prime-order points,
    mixed-order points (P + T), the identity and an encoding that does not decode -- against the oracle; compressed and raw input;
    device and host entry points agree."""
    import torch
    import curve25519_dalek_amd as pkg
    from curve25519_dalek_amd import dalek
    E = pkg.engine
    t1 = orc.ed_decompress(golden.bytes("ed25519.rs", "EIGHT_TORSION_4"))          # a torsion point the reference's own tests hold
    pts = [orc.ed_identity(), t1, orc.ed_double(t1), orc.ed_add(t1, orc.ed_double(t1))]   # and its multiples
    prime = [orc.ed_mul_base(util.rand_scalars(70 + i, 1)[0].tobytes()) for i in range(5)]
    pts += prime
    pts += [orc.ed_add(p, t1) for p in prime[:3]]                                     # mixed order
    pts += [orc.ed_add(prime[3], orc.ed_double(t1))]
    enc = [orc.ed_compress(p) for p in pts] + [(2).to_bytes(32, "little")]            # y = 2 is not on the curve
    want = []
    for p in pts:
        want.append(E.POINT_DECODES | (E.POINT_SMALL_ORDER if orc.ed_is_small_order(p) else 0) | (E.POINT_TORSION_FREE if orc.ed_is_torsion_free(p) else 0))
    want.append(0)
    got = eng.point_order_checks(np.frombuffer(b"".join(enc), np.uint8).reshape(-1, 32))
    assert got.tolist() == want
    assert want[1] & E.POINT_SMALL_ORDER and not want[1] & E.POINT_TORSION_FREE and want[0] == 7 and want[4] == (E.POINT_DECODES | E.POINT_TORSION_FREE)
    assert want[9] == E.POINT_DECODES                                                 # P + T: neither small nor torsion-free
    # single checks leave the other bit clear
    so = eng.point_order_checks(np.frombuffer(b"".join(enc), np.uint8).reshape(-1, 32), which=E.POINT_SMALL_ORDER)
    tf = eng.point_order_checks(np.frombuffer(b"".join(enc), np.uint8).reshape(-1, 32), which=E.POINT_TORSION_FREE)
    assert so.tolist() == [w & ~E.POINT_TORSION_FREE for w in want] and tf.tolist() == [w & ~E.POINT_SMALL_ORDER for w in want]
    # raw input, device entry point, a batch large enough for several blocks
    raw = np.frombuffer(b"".join(pts), np.uint8).reshape(-1, 160)
    reps = 300
    d = torch.from_numpy(np.tile(raw, (reps, 1))).cuda()
    gd = eng.point_order_checks_t(d, E.FMT_RAW160).cpu().numpy()
    assert gd.tolist() == want[:-1] * reps
    # the reference-named front end
    assert dalek.is_small_order(enc) == [bool(w & E.POINT_SMALL_ORDER) if w else None for w in want]
    assert dalek.is_torsion_free(enc) == [bool(w & E.POINT_TORSION_FREE) if w else None for w in want]
    keys = dalek.VerifyingKey.from_bytes(enc[:-1], engine=eng)
    assert dalek.VerifyingKey.is_weak(keys, engine=eng) == [bool(w & E.POINT_SMALL_ORDER) for w in want[:-1]]
