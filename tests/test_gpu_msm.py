"""GPU parity tests for the variable-time MSM (edwards.rs:1002-1031 / pippenger.rs) through the C ABI.
MSM beyond two terms has no embedded answers in the reference; it is pinned the way the reference
pins it (edwards.rs:2276-2296, pippenger.rs:169-198): sum x_i (x_i B) = (sum x_i^2) B, and against
the oracle's own Straus/Pippenger on arbitrary points at sizes the oracle finishes in seconds."""
import hashlib
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu
L = util.L
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    import curve25519_dalek_amd as pkg
    return pkg.Engine(0)


def i2b(x):
    return int(x).to_bytes(32, "little")


def rows(a):
    return [a[i].tobytes() for i in range(a.shape[0])]


def sumsq(x):
    return sum(int.from_bytes(r.tobytes(), "little") ** 2 for r in x) % L


@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 100, 189, 190, 500, 1000, 4097, 70000])
def test_msm_sum_of_squares_identity(eng, orc, n):
    x = util.rand_scalars(100 + n, n)
    pts_raw = eng.mul_base_batch(x, out_fmt=2) if n else np.zeros((0, 160), np.uint8)
    want = orc.ed_compress(orc.ed_mul_base(i2b(sumsq(x))))
    st, got = eng.msm_vartime(x, pts_raw, in_fmt=2, out_fmt=0)
    assert st == 0 and got == want
    # same points given as CompressedEdwardsY
    enc = eng.compress_batch(pts_raw) if n else np.zeros((0, 32), np.uint8)
    st, got = eng.msm_vartime(x, enc, in_fmt=0, out_fmt=0)
    assert st == 0 and got == want
    # raw output decodes to the same point
    st, raw = eng.msm_vartime(x, enc, in_fmt=0, out_fmt=2)
    assert st == 0 and orc.ed_compress(raw) == want


def test_msm_config1_ristretto_1k(eng, orc):
    """BASELINE configs[0] shape on the GPU: 1000-term RistrettoPoint::vartime_multiscalar_mul."""
    seed = b"c25519-hip/cfg1"
    xs = [orc.sc_reduce_wide(hashlib.sha512(seed + i.to_bytes(8, "little")).digest()) for i in range(1000)]
    x = np.frombuffer(b"".join(xs), np.uint8).reshape(-1, 32)
    raw = eng.mul_base_batch(x, out_fmt=2)
    enc = eng.compress_batch(raw, out_fmt=1)
    st, got = eng.msm_vartime(x, enc, in_fmt=1, out_fmt=1)
    want_pt = orc.ed_mul_base(i2b(sum(int.from_bytes(v, "little") ** 2 for v in xs) % L))
    assert st == 0 and got == orc.ris_compress(want_pt)
    # and against the oracle's own dispatch (Pippenger, w = 8) on the same inputs
    opts = [orc.ris_decompress(e) for e in rows(enc)]
    assert got == orc.ris_compress(orc.ed_msm(xs, opts))
    # the dalek-style mirror API
    import curve25519_dalek_amd as pkg
    assert pkg.dalek.RistrettoPoint.vartime_multiscalar_mul(xs, rows(enc), engine=eng) == got
    with pytest.raises(AssertionError):
        pkg.dalek.RistrettoPoint.vartime_multiscalar_mul(xs[:-1], rows(enc), engine=eng)


def test_msm_vs_oracle_arbitrary_points_and_edge_scalars(eng, orc):
    """points NOT in the prime-order subgroup (decompressed random encodings), edge scalars, repeated /
    opposite / identity points, one hot bucket."""
    enc = util.rand_bytes(77, 3000)
    ok = orc.ed_decompress_ok_batch(enc)
    enc = enc[ok == 1][:600]
    n = enc.shape[0]
    s = util.rand_scalars(78, n)
    edge = util.edge_scalars()
    edge = edge[[int.from_bytes(e.tobytes(), "little") < 2**255 for e in edge]]
    s[:edge.shape[0]] = edge
    s[100:160] = s[100]                     # identical scalars: one hot bucket per window
    enc[200:230] = enc[200]                 # identical points
    ident = np.zeros(32, np.uint8); ident[0] = 1
    enc[300] = ident                        # the identity as an operand
    neg = enc[301].copy(); neg[31] ^= 0x80
    enc[302] = neg; s[302] = s[301]         # P and -P with the same scalar cancel
    opts = [orc.ed_decompress(e) for e in rows(enc)]
    want = orc.ed_compress(orc.ed_msm(rows(s), opts))
    st, got = eng.msm_vartime(s, enc, in_fmt=0, out_fmt=0)
    assert st == 0 and got == want
    raw = np.frombuffer(b"".join(opts), np.uint8).reshape(-1, 160)
    st, got = eng.msm_vartime(s, raw, in_fmt=2, out_fmt=0)
    assert st == 0 and got == want
    # small sizes through the same path vs the oracle's Straus
    for m in (1, 2, 5, 31):
        st, got = eng.msm_vartime(s[:m], enc[:m], in_fmt=0, out_fmt=0)
        assert st == 0 and got == orc.ed_compress(orc.ed_msm(rows(s[:m]), opts[:m], which=1))


def test_msm_invalid_point_is_none(eng, orc):
    x = util.rand_scalars(5, 300)
    enc = eng.mul_base_batch(x)
    bad = enc.copy()
    bad[123] = np.frombuffer(i2b(2), np.uint8)      # y = 2 is not on the curve
    assert orc.ed_decompress(i2b(2)) is None
    st, _ = eng.msm_vartime(x, bad, in_fmt=0, out_fmt=0)
    assert st == 1
    import curve25519_dalek_amd as pkg
    assert pkg.dalek.EdwardsPoint.vartime_multiscalar_mul(rows(x), rows(bad), engine=eng) is None
    assert pkg.dalek.EdwardsPoint.vartime_multiscalar_mul(rows(x), rows(enc), engine=eng) is not None


def test_msm_partials_fold(eng, orc):
    """the multi-GPU decomposition on one GPU: shard, partial sums, fold."""
    import torch
    n = 30000
    x = util.rand_scalars(9, n)
    raw = eng.mul_base_batch(x, out_fmt=2)
    want = orc.ed_compress(orc.ed_mul_base(i2b(sumsq(x))))
    dx, dr = torch.from_numpy(x).cuda(), torch.from_numpy(raw).cuda()
    parts = []
    for lo, hi in [(0, 10000), (10000, 10001), (10001, 30000), (30000, 30000)]:
        st, p = eng.msm_partial_t(dx[lo:hi].contiguous(), dr[lo:hi].contiguous(), in_fmt=2)
        assert st == 0
        parts.append(p)
    assert eng.fold_partials(parts, out_fmt=0) == want


def test_msm_full_size_2p21(eng, orc):
    """per-GPU share of BASELINE configs[3] (2^24 terms over 8 GPUs = 2^21 per GPU): points x_i*B are
    generated on the device; expected = (sum x_i^2 mod l) * B."""
    import torch
    n = 1 << 21
    x = util.rand_scalars(2024, n)
    dx = torch.from_numpy(x).cuda()
    draw = eng.mul_base_batch_t(dx, out_fmt=2)
    st, got = eng.msm_vartime_t(dx, draw, in_fmt=2, out_fmt=0)
    want = orc.ed_compress(orc.ed_mul_base(i2b(sumsq(x))))
    assert st == 0 and got == want
    print("MSM 2^21 raw-in: last call %.3f ms (accumulate %.3f ms)" % (eng.last_kernel_ms(), eng.phase_ms(0, 0)))
    denc = eng.compress_batch_t(draw)
    st, got = eng.msm_vartime_t(dx, denc, in_fmt=0, out_fmt=0)
    assert st == 0 and got == want
    print("MSM 2^21 compressed-in: last call %.3f ms (accumulate %.3f ms)" % (eng.last_kernel_ms(), eng.phase_ms(0, 0)))


@pytest.mark.parametrize("n", [(1 << 21) + 5, 300001, 20000])
def test_msm_scalars_up_to_2p255(eng, orc, n):
    """Scalars anywhere below 2^255 (Scalar invariant #1 is all the MSM may assume, scalar.rs:199-225): bits 252 .. 254 fill the overflow
    window (digits up to 7 plus the recoding carry) and the unsigned window below it to its last bucket -- the 17-bit layout, the latency-oriented
    mid-range layout and the digit-matrix sort.  Sum-of-squares identity (x_i B is the same point for x_i and x_i mod l)."""
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(255 + n)
    dx = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    dx[:, 31] &= 0x7F
    dx[::7, 31] |= 0x70                                      # every seventh scalar: bits 252 .. 254 all set
    dx[1::11, :] = 0xFF; dx[1::11, 31] = 0x7F                # and some 2^255 - 1
    draw = eng.mul_base_batch_t(dx, out_fmt=2)
    st, got = eng.msm_vartime_t(dx, draw, in_fmt=2, out_fmt=0)
    assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(i2b(_sumsq_device(dx))))


def test_msm_random_sizes_and_formats(eng, orc):
    """Seeded sweep over sizes drawn log-uniformly from 1 to 2^21 (every path: small tables, digit-matrix sort, chunk-local sort with 16- and 17-bit
    windows, the mid-range layouts in between) in all three input encodings: sum-of-squares identity against the oracle's fixed-base multiplication."""
    import torch
    rng = np.random.default_rng(20260924)
    sizes = sorted(set(int(2 ** rng.uniform(0, 21)) for _ in range(48)) | {1023, 1024, 4095, 4096, 6143, 6144, 8191, 8192, 12287, 12288, 131071, 1 << 19, (1 << 20) + 1})
    for i, n in enumerate(sizes):
        g = torch.Generator(device="cuda"); g.manual_seed(9000 + n)
        dx = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        dx[:, 31] &= 0x0F
        fmt = i % 3
        dpts = eng.mul_base_batch_t(dx, out_fmt=fmt)
        # (a Ristretto encoding decodes to a representative of its coset: the sum is compared as a Ristretto element)
        st, got = eng.msm_vartime_t(dx, dpts, in_fmt=fmt, out_fmt=1 if fmt == 1 else 0)
        want = orc.ed_mul_base(i2b(_sumsq_device(dx)))
        assert st == 0 and got == (orc.ris_compress(want) if fmt == 1 else orc.ed_compress(want)), (n, fmt)


def _sumsq_device(dx):
    """sum x_i^2 mod l for an (n, 32) uint8 CUDA tensor: exact 16-bit-limb Gram matrix on the device
    (every entry < 2^24 * 2^32), recombined with Python integers."""
    import torch
    a = dx.view(torch.int16).to(torch.int64) & 0xFFFF                   # (n, 16) little-endian 16-bit limbs
    tot = 0
    for j in range(16):
        col = (a[:, j:j + 1] * a).sum(0).cpu().tolist()
        for k in range(16):
            tot += int(col[k]) << (16 * (j + k))
    return tot % L


def test_msm_config4_2p24_terms(eng, orc):
    """BASELINE configs[3] at its full size on ONE GPU: 2^24 terms as 8 shards of 2^21 (the decomposition the
    8-GPU run uses, SURVEY.md 8e: per-shard partial sums, 160-byte partials, one fold) and as a single call.
    Points x_i*B are generated on the device; expected = (sum x_i^2 mod l) * B."""
    import torch
    n, shards = 1 << 24, 8
    g = torch.Generator(device="cuda"); g.manual_seed(424242)
    dx = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    dx[:, 31] &= 0x0F
    assert _sumsq_device(dx[:1000]) == sumsq(dx[:1000].cpu().numpy())   # the checker itself
    want = orc.ed_compress(orc.ed_mul_base(i2b(_sumsq_device(dx))))
    draw = eng.mul_base_batch_t(dx, out_fmt=2)
    parts = []
    per = n // shards
    for r in range(shards):
        st, p = eng.msm_partial_t(dx[r * per:(r + 1) * per], draw[r * per:(r + 1) * per], in_fmt=2)
        assert st == 0
        parts.append(p)
    assert eng.fold_partials(parts, out_fmt=0) == want
    st, got = eng.msm_vartime_t(dx, draw, in_fmt=2, out_fmt=0)
    assert st == 0 and got == want
    print("MSM 2^24 single call: %.3f ms" % eng.last_kernel_ms())


def test_msm_affine_and_projective_lanes_mixed(eng, orc):
    """k_prep_raw takes a direct path for a lane whose sixteen points all have Z = 1 (points straight from a
    decompression): lanes of both kinds in one launch, and lanes with both kinds of points, give the MSM of the
    all-projective representation of the same points."""
    n = 1 << 14                                             # 1024 lanes x 16 points, lane t owns t, t + 1024, ...
    x = util.rand_scalars(91, n)
    proj = eng.mul_base_batch(util.rand_scalars(92, n), out_fmt=2)
    st, aff, ok = eng.decompress_batch(eng.compress_batch(proj))
    assert st == 0 and ok.all()
    one = np.zeros(40, np.uint8); one[0] = 1
    assert all(aff[i, 80:120].tobytes() == one.tobytes() for i in (0, 777, n - 1))       # Z is literally 1
    assert any(proj[i, 80:120].tobytes() != one.tobytes() for i in (0, 777, n - 1))
    st, want = eng.msm_vartime(x, proj, in_fmt=2)
    assert st == 0
    idx = np.arange(n)
    for mask in ((idx % 1024) < 512,                        # whole lanes affine, whole lanes projective
                 idx < n // 2,                              # every lane: 8 affine then 8 projective points
                 np.ones(n, bool),                          # everything affine
                 (idx % 3) == 0):
        pts = np.where(mask[:, None], aff, proj)
        st, got = eng.msm_vartime(x, pts, in_fmt=2)
        assert st == 0 and got == want


@pytest.mark.parametrize("env", [{}, {"C25519_MSM_PASS_LOG2": "16"}, {"C25519_MSM_PASS_LOG2": "16", "C25519_PASS_LANES": "1"},
                                 {"C25519_MSM_PASS_LOG2": "16", "C25519_PASS_LANES": "3"},
                                 {"C25519_SORT_SMALL": "1"}, {"C25519_SORT_SMALL": "1", "C25519_MSM_PASS_LOG2": "16"},      # the A/B arms of round 4 stay bit-exact
                                 {"C25519_REDUCE_COOP": "0"},
                                 {"C25519_SORT_CHUNK_LOCAL_MIN": "2048"},                   # the chunk-local sort at its smallest sizes (one to a few chunks per window)
                                 # how far the sort of a continuing pass runs ahead of its predecessor's accumulation (0 = not at all, 1 = the partition
                                 # half, 2 = all of it on a second copy of the lists), with the chunk-local sort on many short passes
                                 {"C25519_SWEEP_EARLY": "0", "C25519_MSM_PASS_LOG2": "16", "C25519_SORT_CHUNK_LOCAL_MIN": "2048"},
                                 {"C25519_SWEEP_EARLY": "1", "C25519_MSM_PASS_LOG2": "16", "C25519_SORT_CHUNK_LOCAL_MIN": "2048"},
                                 {"C25519_SWEEP_EARLY": "2", "C25519_MSM_PASS_LOG2": "16", "C25519_SORT_CHUNK_LOCAL_MIN": "2048"},
                                 {"C25519_SWEEP_EARLY": "2", "C25519_MSM_PASS_LOG2": "16", "C25519_PASS_LANES": "3"},
                                 # the layouts of rounds 1-3 (window width log2 n - 4 throughout, at most 16 bits; 8 buckets per lane in the reduction)
                                 {"C25519_MSM_MIDRANGE_WINDOWS": "0", "C25519_MSM_CMAX": "16", "C25519_RED_LB_MIN": "3"},
                                 {"C25519_MSM_MIDRANGE_WINDOWS": "0", "C25519_SORT_CHUNK_LOCAL_MIN": "2048"},
                                 # every pass normalises its own points; the 512-thread partition
                                 {"C25519_PREP_AHEAD": "0", "C25519_MSM_PASS_LOG2": "16"}, {"C25519_SWEEP_THREADS": "512", "C25519_SORT_CHUNK_LOCAL_MIN": "2048"},
                                 # round 5: single-pass calls in two window groups (bucket order, accumulation and reduction group by group), the sort enqueued
                                 # ahead of the normaliser.  (Three / four groups, SORT_FIRST=2, PREP_SPLIT, 7-bit small tables lost on two boxes each and left the
                                 # tuning build in round 6: profiles/r05_ab_window_groups.txt, r05_ab_prep_split.txt, r05_ab_small_path_range.txt are the record.)
                                 {"C25519_ACC_GROUPS": "2"}, {"C25519_SORT_FIRST": "1"},
                                 {"C25519_REDUCE_MAIN": "0"}, {"C25519_SMALL_DIRECT": "0"},      # the reduction of a single-pass call on the second stream (rounds 3-4); small calls through their slot
                                 # the small path's range: round 4's (4095 terms), and 5-bit windows far beyond the default boundary
                                 {"C25519_MSM_SMALL_MAX": "4095"}, {"C25519_MSM_SMALL_MAX": "40000", "C25519_MSM_SMALL_C": "5"},
                                 # round 6, the mid path: off (the bucket pipeline from 12 288 terms, as it still serves everything beyond 2^17), the normaliser +
                                 # k_accumulate arm from 12 288 terms, many small sort slices, up to 2^18 terms; streaming normaliser / sort reads
                                 {"C25519_MSM_MID_MAX": "0"}, {"C25519_MID_PROJ_MAX": "0"}, {"C25519_MID_SORT_BLOCKS": "1024", "C25519_MSM_MID_MAX": "262144"},
                                 {"C25519_PREP_NT": "1", "C25519_SWEEP_NT": "1", "C25519_MSM_PASS_LOG2": "20"},
                                 # round 6, late: the cap on a bucket lane's list (the first mid path's rule; many lists above it at every size), the over-long
                                 # lists as a launch of their own, the cooperative gather from the first size of the path
                                 {"C25519_MID_LONG_TARGET": "0"}, {"C25519_MID_LONG_TARGET": "2048", "C25519_MID_LONG_TARGET_ALWAYS": "1"},
                                 {"C25519_MID_RAW_FUSED": "0"}, {"C25519_MID_COOP_MIN": "12288"}])
def test_msm_kernel_variants_in_a_fresh_process(orc, env):
    """The remaining knobs (pass size, number of stream sets) are read once per process: 2^16-term passes make a small input
    run many passes (more than the 16 result slots at the largest size: the slots are reused and the record is summed in
    batches), with records prepared ahead (two or three stream sets) or per pass (one); every split must give the same MSM
    on ragged sizes -- last wave partly out of range, last step of a lane out of range, fewer points than one block."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import util, curve25519_dalek_amd as pkg
        from oracle import orc
        eng = pkg.Engine(0)
        L = util.L
        for n in (1, 63, 65, 1500, 4097, 12000, 20001, 70001, 3 * 65536 + 5, 262144 + 64 * 37 + 1, 18 * 65536 + 77):
            x = util.rand_scalars(500 + n, n)
            pts = eng.mul_base_batch(x, out_fmt=2)
            pts[::3] = eng.decompress_batch(eng.compress_batch(pts[::3]))[1]          # a third of the points affine (Z = 1)
            want = orc.ed_compress(orc.ed_mul_base((sum(int.from_bytes(r.tobytes(), "little") ** 2 for r in x) %% L).to_bytes(32, "little")))
            st, got = eng.msm_vartime(x, pts, in_fmt=2, out_fmt=0)
            assert st == 0 and got == want, n
        print("ok")
    """) % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    e = util.tune_env(env) if env else dict(os.environ)          # ({}: the release library, which has no knobs)
    r = subprocess.run(util.child_argv(code), env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (env, r.stdout[-500:], r.stderr[-2000:])


def test_msm_two_passes_odd_size(eng, orc):
    """just past the single-pass limit (msm.hip MSM_PASS_MAX = 2 625 000 terms): two passes of unequal length, column sums added on the device."""
    import torch
    n = 2625000 + 17
    g = torch.Generator(device="cuda"); g.manual_seed(31337)
    dx = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    dx[:, 31] &= 0x0F
    draw = eng.mul_base_batch_t(dx, out_fmt=2)
    st, got = eng.msm_vartime_t(dx, draw, in_fmt=2, out_fmt=0)
    assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(i2b(_sumsq_device(dx))))


@pytest.mark.parametrize("log2n", [13, 17, 20, 21])
def test_msm_maximally_skewed_digits(eng, orc, log2n):
    """Every term carries the SAME scalar, so each window has one bucket holding all n entries: the oversize-bin branch
    of the partition sort, the long-bucket path and the one-pass sort of small inputs (2^21: the 17-bit layout).  Points are distinct
    (P_i = y_i B); expected = s * (sum y_i) * B.  A second batch repeats one point n times (n * s * P)."""
    import torch
    n = 1 << log2n
    s_int = (L - 12345) // 3
    s = np.frombuffer(i2b(s_int), np.uint8)
    g = torch.Generator(device="cuda"); g.manual_seed(4242 + log2n)
    dy = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    dy[:, 31] &= 0x0F
    dP = eng.mul_base_batch_t(dy, out_fmt=2)
    ds = torch.from_numpy(np.tile(s, (n, 1))).cuda()
    ysum = 0
    yb = dy.cpu().numpy()
    for i in range(n):
        ysum += int.from_bytes(yb[i].tobytes(), "little")
    want = orc.ed_compress(orc.ed_mul_base(i2b(s_int * ysum % L)))
    st, got = eng.msm_vartime_t(ds, dP, in_fmt=2, out_fmt=0)
    assert st == 0 and got == want
    same = dP[:1].repeat(n, 1).contiguous()
    want2 = orc.ed_compress(orc.ed_mul(dP[0].cpu().numpy().tobytes(), i2b(s_int * n % L)))
    st, got = eng.msm_vartime_t(ds, same, in_fmt=2, out_fmt=0)
    assert st == 0 and got == want2


@pytest.mark.parametrize("n", [65535, 65536, 65537, (1 << 17) + 3, (1 << 18) + 5, (1 << 19) - 3, (1 << 20) - 1, 2625000, 2625001, (3 << 20) + 1])
def test_msm_window_and_path_boundaries(eng, orc, n):
    """Sizes at which the window width, the sort path (one-pass below 2^16 terms or c < 13, two-pass partition sort
    above) and the pass splitting (above 2 625 000 terms) change: sum-of-squares identity on device-generated points."""
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(77 + n)
    dx = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    dx[:, 31] &= 0x0F
    draw = eng.mul_base_batch_t(dx, out_fmt=2)
    st, got = eng.msm_vartime_t(dx, draw, in_fmt=2, out_fmt=0)
    assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(i2b(_sumsq_device(dx))))


def test_msm_and_verify_over_several_contexts_in_one_process(eng, orc):
    """c25519_msm_vartime_multi / ed25519_verify_batch_multi (the C-ABI multi-GPU path for a host without torch / RCCL):
    contexts on one device here; the results must equal the single-context calls for every shard count."""
    import curve25519_dalek_amd as pkg
    E = pkg.engine
    engs = [eng, pkg.Engine(0), pkg.Engine(0)]
    n = 70001
    x = util.rand_scalars(4100, n)
    pts = eng.mul_base_batch(x, out_fmt=2)
    want = orc.ed_compress(orc.ed_mul_base(i2b(sumsq(x))))
    for k in (1, 2, 3):
        st, got = E.msm_vartime_multi(engs[:k], x, pts, E.FMT_RAW160, E.FMT_EDWARDS_Y)
        assert st == 0 and got == want
    st, got = E.msm_vartime_multi(engs, x[:2], pts[:2], E.FMT_RAW160, E.FMT_EDWARDS_Y)          # fewer terms than contexts
    assert st == 0 and got == orc.ed_compress(orc.ed_msm(rows(x[:2]), rows(pts[:2])))
    enc = eng.compress_batch(pts[:1000]); enc[777] = np.frombuffer(i2b(2), np.uint8)
    st, _ = E.msm_vartime_multi(engs, x[:1000], enc, E.FMT_EDWARDS_Y, E.FMT_EDWARDS_Y)
    assert st == E.NONE
    m = 5000
    seeds = util.rand_bytes(4200, m); msgs = util.rand_bytes(4201, m, 23)
    pks, sigs = orc.ed25519_keygen_sign_batch(seeds, msgs, threads=8)
    M = [msgs[i].tobytes() for i in range(m)]; S = [sigs[i].tobytes() for i in range(m)]; P = [pks[i].tobytes() for i in range(m)]
    for z in (0, 1):
        assert E.verify_batch_multi(engs, M, S, P, z) == E.OK
        bad = list(S); b = bytearray(bad[m - 3]); b[1] ^= 8; bad[m - 3] = bytes(b)
        assert E.verify_batch_multi(engs, M, bad, P, z) == E.VERIFY
        b = bytearray(bad[2]); b[63] |= 0x20; bad[2] = bytes(b)
        assert E.verify_batch_multi(engs, M, bad, P, z) == E.SCALAR_FORMAT
    for e in engs[1:]:
        e.close()


def test_msm_long_runs_in_a_few_chunks(eng, orc):
    """The first 8000 of 2^20 terms share the raw bits of window 3: their entries form ONE run of ~8000 in chunk 0 of a (window, slice)
    bin that still fits LDS -- the pass-2 gather's "more pieces than a wave lists" branch (msm.hip k_part2g), and a long bucket."""
    import torch
    n = 1 << 20
    g = torch.Generator(device="cuda"); g.manual_seed(90210)
    dx = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    dx[:, 31] &= 0x0F
    dx[:8000, 6] = 0x55; dx[:8000, 7] = 0x05                     # bits 48 .. 63
    draw = eng.mul_base_batch_t(dx, out_fmt=2)
    st, got = eng.msm_vartime_t(dx, draw, in_fmt=2, out_fmt=0)
    assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(i2b(_sumsq_device(dx))))


def test_msm_small_sort_blocks_256_slices(orc):
    """C25519_SORT_SMALL=1 (round 4's A/B arm: 256-thread partition blocks) at a pass size whose windows have 256 slices -- as many as the block
    has threads (the slice-start copy-out must loop); fresh process: the knob is read once."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import test_gpu_msm as T, curve25519_dalek_amd as pkg
        from oracle import orc
        eng = pkg.Engine(0)
        for n in (2200001, 2625000):
            g = torch.Generator(device="cuda"); g.manual_seed(4321 + n)
            dx = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
            dx[:, 31] &= 0x0F
            draw = eng.mul_base_batch_vartime_t(dx, out_fmt=2)
            st, got = eng.msm_vartime_t(dx, draw, in_fmt=2, out_fmt=0)
            assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(T.i2b(T._sumsq_device(dx)))), n
        print("ok")
    """) % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run(util.child_argv(code), env=util.tune_env(C25519_SORT_SMALL="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("log2pass,n", [(21, 5898242), (22, 12386305)])
def test_msm_continuing_last_pass_one_sort_chunk_shorter(orc, log2pass, n):
    """ADVICE r3 (msm.hip workspace carve): with C25519_MSM_PASS_LOG2 = 21 / 22 the last of three passes is one sort chunk shorter
    than the others (1 966 080 against 1 966 081 terms: 30 against 31 chunks; 4 128 767 against 4 128 769: 63 against 64) and CONTINUES
    the bucket sums of pass 0 on its stream set -- which it must find at the same workspace offset.  Fresh process: the knob is read once."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import test_gpu_msm as T, curve25519_dalek_amd as pkg
        from oracle import orc
        eng = pkg.Engine(0)
        n = %d
        g = torch.Generator(device="cuda"); g.manual_seed(1234 + n)
        dx = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        dx[:, 31] &= 0x0F
        draw = eng.mul_base_batch_vartime_t(dx, out_fmt=2)
        st, got = eng.msm_vartime_t(dx, draw, in_fmt=2, out_fmt=0)
        assert eng.last_call_phase_ms(0)[1] == 3, eng.last_call_phase_ms(0)
        assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(T.i2b(T._sumsq_device(dx))))
        print("ok")
    """) % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n)
    e = util.tune_env(C25519_MSM_PASS_LOG2=str(log2pass))
    r = subprocess.run(util.child_argv(code), env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-2000:])


def test_msm_every_size_1_to_1024_and_the_small_path_boundaries(eng, orc):
    """The reference's benchmark shapes (dalek_benchmarks.rs:16 MULTISCALAR_SIZES = 1 .. 1024) and everything it hands to Straus
    (edwards.rs:1025): EVERY n in 1 .. 1024 through the small path (small.hip: tables by repeated addition, one lane per (window, term)),
    then its upper sizes (c = 6 from 1024 terms), both sides of its round-4 boundary at 4095 / 4096 terms and of its boundary now, 12287 / 12288, and the
    window-width steps of the digit-matrix sort below 2^16 terms.  Expected: (sum_{i < n} x_i^2) B for every prefix, from ONE fixed-base batch over the prefix sums."""
    nmax = 1024
    x = util.rand_scalars(7001, nmax)
    pts = eng.mul_base_batch(x, out_fmt=2)
    enc = eng.compress_batch(pts)
    ris = eng.compress_batch(pts, out_fmt=1)
    acc, pref = 0, []
    for i in range(nmax):
        acc = (acc + int.from_bytes(x[i].tobytes(), "little") ** 2) % L
        pref.append(acc)
    # (the expected encodings come from the ORACLE's fixed-base multiplication of the prefix sums -- round 4 took them from the engine's own
    #  fixed-base kernel, i.e. compared the product with the product)
    want = orc.mul_base_compress_batch(np.frombuffer(b"".join(i2b(v) for v in pref), np.uint8).reshape(-1, 32), threads=os.cpu_count() or 1)
    assert np.array_equal(want, eng.mul_base_batch(np.frombuffer(b"".join(i2b(v) for v in pref), np.uint8).reshape(-1, 32)))
    for n in range(1, nmax + 1):
        st, got = eng.msm_vartime(x[:n], pts[:n], in_fmt=2, out_fmt=0)
        assert st == 0 and got == want[n - 1].tobytes(), n
    for n in list(range(1, 70)) + [127, 128, 129, 255, 256, 511, 512, 513, 1023, 1024]:          # compressed input: decompression, then the same path
        st, got = eng.msm_vartime(x[:n], enc[:n], in_fmt=0, out_fmt=0)
        assert st == 0 and got == want[n - 1].tobytes(), n
    for n in (1, 5, 64, 190, 1000):                                                                # Ristretto in and out
        st, got = eng.msm_vartime(x[:n], ris[:n], in_fmt=1, out_fmt=1)
        assert st == 0 and got == orc.ris_compress(orc.ed_decompress(want[n - 1].tobytes())), n
    # against the oracle's own Straus / Pippenger on points OUTSIDE the prime-order subgroup (decompressed random encodings) with edge scalars
    rnd = util.rand_bytes(7002, 4000)
    rnd = rnd[orc.ed_decompress_ok_batch(rnd) == 1][:1500]
    s = util.rand_scalars(7003, rnd.shape[0])
    edge = util.edge_scalars()
    edge = edge[[int.from_bytes(e.tobytes(), "little") < 2**255 for e in edge]]
    s[:edge.shape[0]] = edge
    opts = [orc.ed_decompress(e) for e in rows(rnd)]
    raw = np.frombuffer(b"".join(opts), np.uint8).reshape(-1, 160)
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 31, 33, 100, 189, 190, 191, 257, 1023, 1024, 1025, 1500):
        w = orc.ed_compress(orc.ed_msm(rows(s[:n]), opts[:n]))
        assert eng.msm_vartime(s[:n], raw[:n], in_fmt=2, out_fmt=0) == (0, w), n
        assert eng.msm_vartime(s[:n], rnd[:n], in_fmt=0, out_fmt=0) == (0, w), n
    # an undecodable point and a scalar with bit 255 set, on the small path
    bad = enc[:100].copy(); bad[57] = np.frombuffer(i2b(2), np.uint8)
    assert eng.msm_vartime(x[:100], bad, in_fmt=0, out_fmt=0)[0] == 1
    hi = x[:100].copy(); hi[3, 31] |= 0x80
    import curve25519_dalek_amd as pkg
    with pytest.raises(pkg.engine.EngineError):
        eng.msm_vartime(hi, pts[:100], in_fmt=2, out_fmt=0)
    # the boundary of the small path and the narrow windows of the sort
    import torch
    for n in (2047, 2048, 2049, 3000, 4095, 4096, 4097, 6143, 6144, 6145, 8191, 8192, 12287, 12288, 12289, 16383, 16384, 32767, 32768, 50001):
        g = torch.Generator(device="cuda"); g.manual_seed(9000 + n)
        dx = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        dx[:, 31] &= 0x0F
        draw = eng.mul_base_batch_vartime_t(dx, out_fmt=2)
        st, got = eng.msm_vartime_t(dx, draw, in_fmt=2, out_fmt=0)
        assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(i2b(_sumsq_device(dx)))), n
        st, got = eng.msm_vartime_t(dx, eng.compress_batch_t(draw), in_fmt=0, out_fmt=0)
        assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(i2b(_sumsq_device(dx)))), n


# ---- (r6) the mid path (mid.hip): 12 288 .. 2^17 terms of raw points, up to 2^18 + 1 terms over prepared records -------------------------------------------------
def _sum_xy(x, y):
    return sum(int.from_bytes(a.tobytes(), "little") * int.from_bytes(b.tobytes(), "little") for a, b in zip(x, y)) % L


@pytest.mark.parametrize("n", [4096, 5001, 6143, 6144, 6145, 7001, 8191, 8192, 12288, 12289, 16383, 16384, 16391, 20000, 32768, 65535, 65536, 100003, 131072, 200003, 262144])
def test_mid_path_sizes_raw_points_and_encodings(eng, orc, n):
    """pippenger.rs:67-160 through the mid path at every kind of size it serves (rows of the digit matrix padded / not padded to eight terms, every window width 12 .. 15,
    both ends of the range): P_i = y_i B with independent x_i, so the expected point is (sum x_i y_i) B from the ORACLE's fixed-base multiplication; raw points with
    their projective Z (the fixed-base kernel leaves Z != 1), device and host pointers, then the same points as CompressedEdwardsY and as CompressedRistretto (the
    decompression's affine records: the other accumulation kernel)."""
    import torch
    x = util.rand_scalars(9000 + n, n); y = util.rand_scalars(9500 + n, n)
    e = util.edge_scalars(); e = e[[int.from_bytes(v.tobytes(), "little") < 2**255 for v in e]]
    x[:e.shape[0]] = e                                                   # unreduced scalars up to 2^255 - 1 among them: every digit pattern incl. the top window's carry
    raw = eng.mul_base_batch(y, out_fmt=2)
    want_pt = orc.ed_mul_base(i2b(_sum_xy(x, y)))
    want = orc.ed_compress(want_pt)
    dx, dr = torch.from_numpy(x).cuda(), torch.from_numpy(raw).cuda()
    st, got = eng.msm_vartime_t(dx, dr, in_fmt=2, out_fmt=0)
    assert st == 0 and got == want
    st, got = eng.msm_vartime(x, raw, in_fmt=2, out_fmt=0)
    assert st == 0 and got == want
    if n in (4096, 5001, 6143, 6144, 8191, 12288, 16391, 65536, 131072, 262144):      # (encoded points take the mid path from 4096 terms, raw ones from 6144)
        enc = eng.compress_batch(raw)
        st, got = eng.msm_vartime_t(dx, torch.from_numpy(enc).cuda(), in_fmt=0, out_fmt=0)
        assert st == 0 and got == want
        st, got = eng.msm_vartime(x, enc, in_fmt=0, out_fmt=0)
        assert st == 0 and got == want
        ris = eng.compress_batch(raw, out_fmt=1)
        st, got = eng.msm_vartime_t(dx, torch.from_numpy(ris).cuda(), in_fmt=1, out_fmt=1)
        assert st == 0 and got == orc.ris_compress(want_pt)
        bad = enc.copy(); bad[n // 3] = np.frombuffer(i2b(2), np.uint8)      # an encoding that does not decode: Option::None of the reference
        st, _ = eng.msm_vartime_t(dx, torch.from_numpy(bad).cuda(), in_fmt=0, out_fmt=0)
        assert st == 1


def test_mid_path_against_the_oracles_pippenger_on_arbitrary_points(eng, orc):
    """13 000 points that are NOT in the prime-order subgroup (decompressed random encodings, Z = 1: the other class of raw input) with edge scalars, judged by the
    oracle's own Pippenger (pippenger.rs:67-160 restated) -- not by the sum-of-products identity."""
    enc = util.rand_bytes(4242, 30000)
    ok = orc.ed_decompress_ok_batch(enc)
    enc = enc[ok == 1][:13000]
    n = enc.shape[0]
    assert n == 13000
    s = util.rand_scalars(4243, n)
    e = util.edge_scalars(); e = e[[int.from_bytes(v.tobytes(), "little") < 2**255 for v in e]]
    s[:e.shape[0]] = e
    _, raw, okd = eng.decompress_batch(enc)
    assert okd.all()
    want = orc.ed_compress(orc.ed_msm(rows(s), [orc.ed_decompress(b) for b in rows(enc)]))
    st, got = eng.msm_vartime(s, raw, in_fmt=2, out_fmt=0)
    assert st == 0 and got == want
    st, got = eng.msm_vartime(s, enc, in_fmt=0, out_fmt=0)
    assert st == 0 and got == want


@pytest.mark.parametrize("n", [6144, 12288, 40000])
def test_mid_path_skewed_digits_long_lists(eng, orc, n):
    """Digit distributions that put thousands of terms into ONE bucket of every window (equal scalars), into two buckets (s and l - s: the same buckets with opposite
    signs), or leave whole windows empty (small scalars) -- the over-long lists go through k_mid_long's segments and the last-finisher sum; a third of the terms random."""
    import torch
    y = util.rand_scalars(777 + n, n)
    raw = eng.mul_base_batch(y, out_fmt=2)
    x = util.rand_scalars(778 + n, n)
    s0 = int.from_bytes(x[0].tobytes(), "little") % L
    third = n // 3
    x[:third] = np.frombuffer(i2b(s0), np.uint8)
    x[third:third + third // 2] = np.frombuffer(i2b(L - s0), np.uint8)
    x[third + third // 2:2 * third] = np.frombuffer(i2b(5), np.uint8)
    want = orc.ed_compress(orc.ed_mul_base(i2b(_sum_xy(x, y))))
    st, got = eng.msm_vartime_t(torch.from_numpy(x).cuda(), torch.from_numpy(raw).cuda(), in_fmt=2, out_fmt=0)
    assert st == 0 and got == want
    enc = eng.compress_batch(raw)
    st, got = eng.msm_vartime(x, enc, in_fmt=0, out_fmt=0)
    assert st == 0 and got == want


def test_mid_path_flags_records_and_fold(eng, orc):
    """A scalar with bit 255 set fails the call (Scalar invariant #1) wherever it sits; the partial-result RECORD of the mid path (device-resident, header written by the
    reduction's last block) folds with records of the small path and of the bucket pipeline into the oracle's point; all-zero scalars give the identity."""
    import torch
    import curve25519_dalek_amd as pkg
    n = 20011
    x = util.rand_scalars(31, n); y = util.rand_scalars(32, n)
    raw = eng.mul_base_batch(y, out_fmt=2)
    dx, dr = torch.from_numpy(x).cuda(), torch.from_numpy(raw).cuda()
    for pos in (0, 255, 256, n - 1):
        bad = x.copy(); bad[pos, 31] |= 0x80
        with pytest.raises(pkg.EngineError, match="bit 255"):
            eng.msm_vartime_t(torch.from_numpy(bad).cuda(), dr, in_fmt=2, out_fmt=0)
    st, got = eng.msm_vartime_t(dx, dr, in_fmt=2, out_fmt=0)                   # the context is still good
    assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(i2b(_sum_xy(x, y))))
    z = torch.zeros_like(dx)
    st, got = eng.msm_vartime_t(z, dr, in_fmt=2, out_fmt=0)
    assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(i2b(0)))
    # records: shards of 300 (small path), 20 011 - 300 terms (the mid path) and 300 000 terms (bucket pipeline), folded once
    m = 300000
    x2 = util.rand_scalars(33, m); y2 = util.rand_scalars(34, m)
    raw2 = eng.mul_base_batch(y2, out_fmt=2)
    recs = [eng.msm_partial_record_t(dx[:300], dr[:300]), eng.msm_partial_record_t(dx[300:], dr[300:]),
            eng.msm_partial_record_t(torch.from_numpy(x2).cuda(), torch.from_numpy(raw2).cuda())]
    st, got = pkg.engine.fold_partial_records(torch.stack(recs).cpu().numpy(), out_fmt=0)
    assert st == 0 and got == orc.ed_compress(orc.ed_mul_base(i2b((_sum_xy(x, y) + _sum_xy(x2, y2)) % L)))


def test_mid_path_lost_publication_is_recovered(orc):
    """The mid path publishes its record like the small path does (k_reduce_b4pub -> page-locked host memory -> the sequence word the host polls): the forced loss of
    every third publication (tuning build) must be recovered through the copy path with the normal result."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, faulthandler
        faulthandler.dump_traceback_later(300, exit=True)
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np, torch
        import curve25519_dalek_amd as pkg, util
        from oracle import orc
        L = util.L
        eng = pkg.Engine(0)
        for n in (12288, 30000, 131072):
            x = util.rand_scalars(5 + n, n); y = util.rand_scalars(6 + n, n)
            raw = eng.mul_base_batch(y, out_fmt=2)
            tot = sum(int.from_bytes(a.tobytes(), "little") * int.from_bytes(b.tobytes(), "little") for a, b in zip(x, y)) %% L
            want = orc.ed_compress(orc.ed_mul_base(tot.to_bytes(32, "little")))
            dx, dr = torch.from_numpy(x).cuda(), torch.from_numpy(raw).cuda()
            for rep in range(4):
                st, got = eng.msm_vartime_t(dx, dr, in_fmt=2, out_fmt=0)
                assert st == 0 and got == want, (n, rep)
        assert eng.counter(2) == 12 and eng.counter(1) == 4, (eng.counter(2), eng.counter(1))
        print("ok")
    """) % (ROOT, ROOT)
    r = subprocess.run(util.child_argv(code), env=util.tune_env(C25519_FAULT_LOSE_PUBLICATION="3", C25519_PUBLISH_SPIN_US="1000"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])
