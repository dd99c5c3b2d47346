"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding + exchange + fold code of
curve25519-dalek_amd/multi.py (the per-rank partial sums, which need a GPU in production, are supplied
here by the oracle -- the test checks the decomposition, not the kernels)."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = 2**252 + 27742317777372353535851937790883648493


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch.distributed as dist
    import curve25519_dalek_amd as pkg
    from oracle import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(99)                         # same inputs on every rank
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    lo, hi = pkg.multi.shard_range(n, rank, world)
    xs = [x[i].tobytes() for i in range(lo, hi)]
    pts = [orc.ed_mul_base(s) for s in xs]
    partial = orc.ed_msm(xs, pts)                            # stand-in for c25519_msm_partial_dev
    out = pkg.multi.gather_fold(partial, pkg.engine.FMT_EDWARDS_Y)
    total = sum(int.from_bytes(x[i].tobytes(), "little") ** 2 for i in range(n)) % L
    want = orc.ed_compress(orc.ed_mul_base(total.to_bytes(32, "little")))
    # the sharded entry point itself with a stand-in engine (the record of a shard = its oracle partial sum, packed), and its per-step breakdown
    # (multi.StepTimes, what a --gpus N bench line reports): the three parts are non-negative and add up to the wall-clock of the calls
    import torch

    class OracleEngine:
        def msm_partial_record_t(self, scalars_t, points_t, in_fmt):
            sc = scalars_t.numpy(); pt = points_t.numpy()
            part = orc.ed_msm([sc[i].tobytes() for i in range(sc.shape[0])], [pt[i].tobytes() for i in range(pt.shape[0])])
            return torch.frombuffer(bytearray(pkg.engine.partial_record_pack(part)), dtype=torch.uint8)
    xs_t = torch.from_numpy(x[lo:hi].copy()); pts_t = torch.from_numpy(np.frombuffer(b"".join(pts), np.uint8).reshape(-1, 160).copy()) if hi > lo else torch.zeros((0, 160), dtype=torch.uint8)
    tm = pkg.multi.StepTimes()
    ok_t = True
    for _ in range(2):
        st_t, got_t = pkg.multi.msm_vartime_sharded(OracleEngine(), xs_t, pts_t, pkg.engine.FMT_RAW160, pkg.engine.FMT_EDWARDS_Y, times=tm)
        ok_t = ok_t and st_t == 0 and got_t == want
    bm = tm.mean()
    ok_t = ok_t and bm["calls"] == 2 and min(bm["shard_ms"], bm["collective_ms"], bm["d2h_fold_ms"]) >= 0 and bm["collective_ms"] > 0
    ok_t = ok_t and abs(bm["shard_ms"] + bm["collective_ms"] + bm["d2h_fold_ms"] - bm["sum_ms"]) <= 0.05 * bm["sum_ms"]
    q.put((rank, out == want and ok_t, (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1000, 7])
def test_sharded_msm_exchange_and_fold_gloo(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    ranges = sorted(r for _, _, r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == n and ranges[0][1] == ranges[1][0]


def test_shard_range_properties():
    import curve25519_dalek_amd as pkg
    for n in (0, 1, 7, 8, 1000, 2**24):
        for world in (1, 2, 3, 8):
            rs = [pkg.multi.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_fold_partials_without_gpu():
    import curve25519_dalek_amd as pkg
    from oracle import orc
    a = orc.ed_mul_base((5).to_bytes(32, "little")); b = orc.ed_mul_base((7).to_bytes(32, "little"))
    assert pkg.multi.fold_partials([a, b]) == orc.ed_compress(orc.ed_mul_base((12).to_bytes(32, "little")))
    assert pkg.multi.fold_partials([]) == (1).to_bytes(32, "little")          # empty sum = identity
    assert pkg.multi.fold_partials([a], pkg.engine.FMT_RISTRETTO) == orc.ris_compress(a)
    raw = pkg.multi.fold_partials([a, b], pkg.engine.FMT_RAW160)
    assert orc.ed_compress(raw) == orc.ed_compress(orc.ed_add(a, b))


def _verdict_worker(rank, world, port, statuses, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import curve25519_dalek_amd as pkg
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = [pkg.multi.combine_verdicts(case[rank]) for case in statuses]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_verify_verdict_precedence_gloo():
    """verify_batch over two ranks: the batch verdict is the worst shard verdict in the reference's order
    (batch.rs:208-211 ScalarFormat before :244-250 Verify), identical on every rank."""
    OK, NONE, SCALAR_FORMAT, VERIFY = 0, 1, 2, 3
    cases = [(OK, OK), (OK, VERIFY), (VERIFY, OK), (VERIFY, SCALAR_FORMAT), (SCALAR_FORMAT, VERIFY), (OK, SCALAR_FORMAT),
             (NONE, SCALAR_FORMAT), (VERIFY, NONE), (VERIFY, VERIFY)]
    want = [OK, VERIFY, VERIFY, SCALAR_FORMAT, SCALAR_FORMAT, SCALAR_FORMAT, NONE, NONE, VERIFY]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_verdict_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, out in res:
        assert out == want


class _FakeEngine:
    """Stands in for Engine in the CPU tests: what the GPU computes per rank -- the MSM partial record, H(R||A||M), the
    share of the batch equation for given z_i, the shard verdict of the device z-mode -- comes from the oracle; everything
    else -- sharding, the exchange of the records, the fold, the ONE transcript over the gathered hram / s, the verdict --
    is the production code of multi.py and of the library's host-side fold."""

    def __init__(self, orc, bad_point_rank=None, rank=0):
        self.orc, self.bad_point_rank, self.rank = orc, bad_point_rank, rank
        self.calls = 0

    def msm_partial_record_t(self, scalars_t, points_t, in_fmt):
        import torch
        import curve25519_dalek_amd as pkg
        self.calls += 1
        if self.bad_point_rank == self.rank:                         # NONE: a point of this shard does not decode
            rec = pkg.engine.partial_record_pack(self.orc.ed_identity(), pkg.engine.NONE)
        else:
            xs = [bytes(r) for r in scalars_t.numpy()]; ps = [bytes(r) for r in points_t.numpy()]
            rec = pkg.engine.partial_record_pack(self.orc.ed_msm(xs, ps) if xs else self.orc.ed_identity())
        return torch.frombuffer(bytearray(rec), dtype=torch.uint8)

    def verify_batch_t(self, msgs_t, msg_off_t, sigs_t, pks_t, z_mode, pk_points=None):
        self.calls += 1
        off = msg_off_t.numpy(); blob = bytes(msgs_t.numpy())
        n = sigs_t.shape[0]
        M = [blob[int(off[i]):int(off[i + 1])] for i in range(n)]
        return self.orc.ed25519_verify_batch(M, [bytes(r) for r in sigs_t.numpy()], [bytes(r) for r in pks_t.numpy()])

    def batch_hram_t(self, msgs_t, msg_off_t, sigs_t, pks_t):
        import hashlib
        import numpy as np
        import torch
        off = msg_off_t.numpy(); blob = bytes(msgs_t.numpy()); n = sigs_t.shape[0]
        out = bytearray(n * 64 + 64)
        bad_s = 0
        for i in range(n):
            sg = bytes(sigs_t[i].numpy())
            out[64 * i:64 * i + 64] = hashlib.sha512(sg[:32] + bytes(pks_t[i].numpy()) + blob[int(off[i]):int(off[i + 1])]).digest()
            bad_s += int(int.from_bytes(sg[32:], "little") >= L)
        out[n * 64:n * 64 + 4] = int(bad_s).to_bytes(4, "little")
        return torch.frombuffer(out, dtype=torch.uint8)

    def verify_batch_record_t(self, sigs_t, pks_t, hram_t, z_t, pk_points=None):
        """-sum z_i s_i B + sum z_i R_i + sum (z_i h_i) A_i (batch.rs:213-244) with the given z_i, as a record"""
        import torch
        import curve25519_dalek_amd as pkg
        orc = self.orc
        self.calls += 1
        n = sigs_t.shape[0]
        hr = bytes(hram_t.numpy()); zz = bytes(z_t.numpy())
        counters = [0] * 8
        counters[4] = int.from_bytes(hr[n * 64:n * 64 + 4], "little")
        scal, pts, bsum = [], [], 0
        for i in range(n):
            sg = bytes(sigs_t[i].numpy())
            z = int.from_bytes(zz[16 * i:16 * i + 16], "little")
            h = int.from_bytes(hr[64 * i:64 * i + 64], "little") % L
            R = orc.ed_decompress(sg[:32]); A = orc.ed_decompress(bytes(pks_t[i].numpy()))
            okR, okA = R is not None, A is not None
            counters[3] += int(not okR); counters[2] += int(not okA)
            bsum += z * (int.from_bytes(sg[32:], "little") % L)
            scal += [z.to_bytes(32, "little"), (z * h % L).to_bytes(32, "little")]
            pts += [R if okR else orc.ed_identity(), A if okA else orc.ed_identity()]
        scal.append(((-bsum) % L).to_bytes(32, "little")); pts.append(orc.ed_basepoint())
        rec = pkg.engine.partial_record_pack(orc.ed_msm(scal, pts), pkg.engine.OK, counters)
        return torch.frombuffer(bytearray(rec), dtype=torch.uint8)


def _sharded_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import torch.distributed as dist
    import curve25519_dalek_amd as pkg
    from oracle import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E = pkg.engine
    out = {}
    # ---- msm_vartime_sharded: this rank's shard in, the same total on every rank -----------------------------------
    n = 301
    rng = np.random.default_rng(4242)                       # same inputs on every rank
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    y = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); y[:, 31] &= 0x0F
    pts = np.frombuffer(b"".join(orc.ed_mul_base(y[i].tobytes()) for i in range(n)), np.uint8).reshape(n, 160)
    lo, hi = pkg.multi.shard_range(n, rank, world)
    eng = _FakeEngine(orc, rank=rank)
    st, got = pkg.multi.msm_vartime_sharded(eng, torch.from_numpy(x[lo:hi].copy()), torch.from_numpy(pts[lo:hi].copy()), E.FMT_RAW160, E.FMT_EDWARDS_Y)
    want = orc.ed_compress(orc.ed_msm([x[i].tobytes() for i in range(n)], [pts[i].tobytes() for i in range(n)]))
    out["msm"] = (st == 0 and got == want and eng.calls == 1)
    # a point that does not decode on ONE rank -> NONE on EVERY rank (Option::None of the reference), no result
    eng = _FakeEngine(orc, bad_point_rank=1, rank=rank)
    st, got = pkg.multi.msm_vartime_sharded(eng, torch.from_numpy(x[lo:hi].copy()), torch.from_numpy(pts[lo:hi].copy()), E.FMT_RAW160, E.FMT_EDWARDS_Y)
    out["msm_none"] = (st == E.NONE and got is None)
    # ---- verify_batch_sharded: both z-modes ---------------------------------------------------------------------------------
    m = 25                                                  # odd: the two shards differ in size
    seeds = rng.integers(0, 256, size=(m, 32), dtype=np.uint8); msgs = rng.integers(0, 256, size=(m, 19), dtype=np.uint8)
    pks, sigs = orc.ed25519_keygen_sign_batch(seeds, msgs, threads=1)
    res = {E.Z_TRANSCRIPT: [], E.Z_DEVICE: []}
    for z_mode in (E.Z_TRANSCRIPT, E.Z_DEVICE):
        for bad_at, kind in ((None, None), (3, "verify"), (m - 2, "verify"), (m - 2, "scalar"), (3, "scalar")):
            sg = sigs.copy()
            if kind == "verify":
                sg[bad_at, 5] ^= 1
            elif kind == "scalar":
                sg[bad_at, 63] |= 0x20
            if bad_at == 3 and kind == "scalar":
                sg[m - 2, 5] ^= 1                                # Verify on the other rank: ScalarFormat still wins
            lo, hi = pkg.multi.shard_range(m, rank, world)
            off = np.arange(0, 19 * (hi - lo + 1), 19, dtype=np.int64)
            eng = _FakeEngine(orc, rank=rank)
            v = pkg.multi.verify_batch_sharded(eng, torch.from_numpy(msgs[lo:hi].reshape(-1).copy()), torch.from_numpy(off), torch.from_numpy(sg[lo:hi].copy()),
                                              torch.from_numpy(pks[lo:hi].copy()), z_mode=z_mode)
            res[z_mode].append(v)
    out["verify_transcript"] = res[E.Z_TRANSCRIPT]
    out["verify_device"] = res[E.Z_DEVICE]
    # ---- the z_i of the sharded transcript mode are the ONE transcript's: equal to the oracle's over the whole batch --------------
    lo, hi = pkg.multi.shard_range(m, rank, world)
    off = np.arange(0, 19 * (hi - lo + 1), 19, dtype=np.int64)
    eng = _FakeEngine(orc, rank=rank)
    hram_t = eng.batch_hram_t(torch.from_numpy(msgs[lo:hi].reshape(-1).copy()), torch.from_numpy(off), torch.from_numpy(sigs[lo:hi].copy()), torch.from_numpy(pks[lo:hi].copy()))
    hram_all, sigs_all, mylo = pkg.multi.gather_transcript_inputs(hram_t, torch.from_numpy(sigs[lo:hi].copy()))
    z_all = E.batch_transcript_zs(hram_all, sigs_all)
    import hashlib
    hr = [hashlib.sha512(sigs[i, :32].tobytes() + pks[i].tobytes() + msgs[i].tobytes()).digest() for i in range(m)]
    want_z = orc.batch_transcript_zs(hr, [sigs[i, 32:].tobytes() for i in range(m)])
    out["one_transcript"] = (mylo == lo and z_all.tobytes() == b"".join(want_z) and sigs_all[:, 32:].tobytes() == sigs[:, 32:].tobytes())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_entry_points_control_flow_gloo():
    """world_size 2 through multi.msm_vartime_sharded / multi.verify_batch_sharded themselves (a fake engine supplies
    what the GPU computes): equal results on both ranks, NONE agreed on by all_reduce, verdict precedence."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    OK, SCALAR_FORMAT, VERIFY = 0, 2, 3
    for _, out in res:
        assert out["msm"] and out["msm_none"]
        assert out["verify_transcript"] == [OK, VERIFY, VERIFY, SCALAR_FORMAT, SCALAR_FORMAT]
        assert out["verify_device"] == [OK, VERIFY, VERIFY, SCALAR_FORMAT, SCALAR_FORMAT]
        assert out["one_transcript"]


def test_bench_gpus_flag_fails_loudly_without_the_gpus():
    """`bench.py --gpus N` must spawn N ranks or refuse: on a box with fewer than N GPUs it exits non-zero with a message
    (round 1 parsed the flag, ignored it and printed n_gpus = 1)."""
    import subprocess
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "--gpus 2" in (r.stderr + r.stdout) and "GPU" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout
