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
    q.put((rank, out == want, (lo, hi)))
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
