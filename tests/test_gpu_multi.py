"""The multi-rank path on REAL kernels and REAL collectives, as far as one GPU allows:

* two processes that share cuda:0 (gloo carries the exchange: RCCL refuses two ranks on one device), each running the
  real engine on its shard through multi.msm_vartime_sharded / multi.verify_batch_sharded -- the result must equal the
  single-context call, the oracle, and (transcript z-mode) the z_i of the ONE transcript over the whole batch;
* one process with backend "nccl" and world_size 1 and force_collective=True, so that all_gather_into_tensor /
  all_reduce really execute in RCCL on device uint8 / int32 / int64 tensors;
* bench.py launched by torch.distributed.run with --gpus 1, so that its `use_dist` branch (init_process_group("nccl"),
  barriers, the max-over-ranks all_reduce, the forced collective in every step) runs on hardware.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = 2**252 + 27742317777372353535851937790883648493


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


_COMMON = r'''
import os, sys, json, hashlib
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import curve25519_dalek_amd as pkg
from oracle import orc
E = pkg.engine
L = 2**252 + 27742317777372353535851937790883648493
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(os.environ["TEST_BACKEND"], rank=rank, world_size=world, **({"device_id": dev} if os.environ["TEST_BACKEND"] == "nccl" else {}))
force = os.environ.get("TEST_FORCE", "0") == "1"
eng = pkg.Engine(0)
out = {"rank": rank}

# ---- MSM: the same inputs on every rank (same seed), every rank takes its shard ------------------------------------------------
n = int(os.environ.get("TEST_N", str((1 << 20) + 37)))
g = torch.Generator(device=dev); g.manual_seed(777)
x = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g); x[:, 31] &= 0x0F
y = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g); y[:, 31] &= 0x0F
pts = eng.mul_base_batch_vartime_t(y, E.FMT_RAW160)
lo, hi = pkg.multi.shard_range(n, rank, world)
st, got = pkg.multi.msm_vartime_sharded(eng, x[lo:hi].contiguous(), pts[lo:hi].contiguous(), E.FMT_RAW160, E.FMT_EDWARDS_Y, force_collective=force)
st1, single = eng.msm_vartime_t(x, pts, E.FMT_RAW160, E.FMT_EDWARDS_Y)
xi = x.cpu().numpy(); yi = y.cpu().numpy()
acc = sum(int.from_bytes(xi[i].tobytes(), "little") * int.from_bytes(yi[i].tobytes(), "little") for i in range(n)) %% L
want = orc.ed_compress(orc.ed_mul_base(acc.to_bytes(32, "little")))
out["msm"] = (st == 0 and st1 == 0 and got == single == want)
# the per-step breakdown a --gpus N bench line reports (multi.StepTimes): same result, and the three parts add up to the wall-clock of the calls
tm = pkg.multi.StepTimes()
for _ in range(3):
    st_t, got_t = pkg.multi.msm_vartime_sharded(eng, x[lo:hi].contiguous(), pts[lo:hi].contiguous(), E.FMT_RAW160, E.FMT_EDWARDS_Y, force_collective=force, times=tm)
bm = tm.mean()
out["step_times"] = bm
out["step_times_ok"] = bool(st_t == 0 and got_t == want and bm["calls"] == 3 and all(bm[k] >= 0 for k in ("shard_ms", "collective_ms", "d2h_fold_ms"))
                            and abs(bm["shard_ms"] + bm["collective_ms"] + bm["d2h_fold_ms"] - bm["sum_ms"]) <= 0.05 * bm["sum_ms"] and bm["shard_ms"] > 0)
# Ristretto / raw outputs take the same exchange
st, got_r = pkg.multi.msm_vartime_sharded(eng, x[lo:hi].contiguous(), pts[lo:hi].contiguous(), E.FMT_RAW160, E.FMT_RISTRETTO, force_collective=force)
out["msm_ristretto"] = (st == 0 and got_r == orc.ris_compress(orc.ed_mul_base(acc.to_bytes(32, "little"))))
# a point that does not decode in ONE shard (the last term of the last rank) -> NONE on every rank
cp = eng.compress_batch_t(pts)
bad = cp.clone(); bad[n - 1] = torch.tensor(list((2).to_bytes(32, "little")), dtype=torch.uint8, device=dev)      # y = 2 is not on the curve
st_ok, got_c = pkg.multi.msm_vartime_sharded(eng, x[lo:hi].contiguous(), cp[lo:hi].contiguous(), E.FMT_EDWARDS_Y, E.FMT_EDWARDS_Y, force_collective=force)
st_bad, got_b = pkg.multi.msm_vartime_sharded(eng, x[lo:hi].contiguous(), bad[lo:hi].contiguous(), E.FMT_EDWARDS_Y, E.FMT_EDWARDS_Y, force_collective=force)
out["msm_compressed"] = (st_ok == 0 and got_c == want)
out["msm_none"] = (st_bad == E.NONE and got_b is None)
# an empty shard takes part in the exchange like any other
elo, ehi = (0, n) if rank == 0 else (n, n)
st, got_e = pkg.multi.msm_vartime_sharded(eng, x[elo:ehi].contiguous(), pts[elo:ehi].contiguous(), E.FMT_RAW160, E.FMT_EDWARDS_Y, force_collective=force)
out["msm_empty_shard"] = (st == 0 and got_e == want)

# ---- verify_batch: both z-modes, honest and forged, against the single-context verdict --------------------------------------
m = int(os.environ.get("TEST_M", "6001"))
seeds = torch.randint(0, 256, (m, 32), dtype=torch.uint8, device=dev, generator=g)
msgs = torch.randint(0, 256, (m * 24,), dtype=torch.uint8, device=dev, generator=g)
off = torch.arange(0, 24 * (m + 1), 24, dtype=torch.int64, device=dev)
pks, sigs = eng.sign_batch_t(seeds, msgs, off)
lo, hi = pkg.multi.shard_range(m, rank, world)
def shard(sg):
    return (msgs[24 * lo:24 * hi].contiguous(), (off[lo:hi + 1] - off[lo]).contiguous(), sg[lo:hi].contiguous(), pks[lo:hi].contiguous())
res = {}
for z_mode in (E.Z_TRANSCRIPT, E.Z_DEVICE):
    r = []
    forged = sigs.clone(); forged[m - 3, 7] ^= 4                 # R of a signature in the last shard
    noncanon = forged.clone(); noncanon[2, 63] |= 0xE0            # and a non-canonical s in the first shard: ScalarFormat wins
    for sg in (sigs, forged, noncanon):
        v = pkg.multi.verify_batch_sharded(eng, *shard(sg), z_mode=z_mode, force_collective=force)
        v1 = eng.verify_batch_t(msgs, off, sg, pks, z_mode)
        r.append((v, v1))
    res[z_mode] = r
out["verify"] = res
# transcript z-mode: the z_i every rank derived are the z_i of ONE transcript over the whole batch = the single-context
# engine's = the oracle's, byte for byte
hram_t = eng.batch_hram_t(*shard(sigs)[:2], shard(sigs)[2], shard(sigs)[3])
hram_all, sigs_all, mylo = pkg.multi.gather_transcript_inputs(hram_t, shard(sigs)[2], force_collective=force)
z_all = E.batch_transcript_zs(hram_all, sigs_all)
mh = msgs.cpu().numpy().reshape(m, 24); sh = sigs.cpu().numpy(); ph = pks.cpu().numpy()
M = [mh[i].tobytes() for i in range(m)]; S = [sh[i].tobytes() for i in range(m)]; P = [ph[i].tobytes() for i in range(m)]
hr = [hashlib.sha512(S[i][:32] + P[i] + M[i]).digest() for i in range(m)]
z_orc = b"".join(orc.batch_transcript_zs(hr, [s[32:] for s in S]))
limit = 3 << (int(os.environ.get("C25519_VERIFY_PASS_LOG2", "20")) - 1)       # c25519_debug_batch_zs serves one pass
z_single = eng.debug_batch_zs(M, S, P, E.Z_TRANSCRIPT).tobytes() if m <= limit else z_orc
out["one_transcript"] = (mylo == lo and z_all.tobytes() == z_single == z_orc)
out["verdict_collective"] = [pkg.multi.combine_verdicts(v, device=dev, force_collective=force) for v in (0, 3, 2, 1)]
print("RESULT " + json.dumps(out))
dist.barrier()
dist.destroy_process_group()
'''


def _launch(world, backend, force, extra_env=None, timeout=900):
    port = _free_port()
    procs = []
    code = _COMMON % {"root": ROOT}
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "TEST_BACKEND": backend,
                    "TEST_FORCE": "1" if force else "0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.update(extra_env or {})
        if any(k.startswith("C25519_") for k in (extra_env or {})):      # knobs exist only in the tuning build (csrc/msm_internal.h C25519_KNOB)
            env["C25519_HIP_LIB"] = os.path.join(ROOT, "curve25519-dalek_amd", "lib", "libc25519hip_tune.so")
        procs.append(subprocess.Popen(util.child_argv(code), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, (so[-2000:], se[-4000:])
        line = [ln for ln in so.splitlines() if ln.startswith("RESULT ")]
        assert line, (so[-2000:], se[-2000:])
        outs.append(json.loads(line[-1][7:]))
    return outs


def _check(outs):
    OK, NONE, SCALAR_FORMAT, VERIFY = 0, 1, 2, 3
    for o in outs:
        assert o["msm"] and o["msm_ristretto"] and o["msm_compressed"] and o["msm_none"] and o["msm_empty_shard"], o
        for z_mode in ("0", "1"):
            got = o["verify"][z_mode]
            assert [v for v, _ in got] == [OK, VERIFY, SCALAR_FORMAT], (z_mode, got)
            assert all(v == v1 for v, v1 in got), (z_mode, got)         # sharded verdict == single-context verdict
        assert o["one_transcript"], o
        assert o["step_times_ok"], o["step_times"]              # shard_ms + collective_ms + d2h_fold_ms = the step, within 5 %
        assert o["verdict_collective"] == [OK, VERIFY, SCALAR_FORMAT, NONE]


def test_two_ranks_on_one_gpu_real_engine_gloo():
    """world_size 2, both ranks on cuda:0, the real engine computes every shard; gloo carries the records."""
    outs = _launch(2, "gloo", False)
    assert sorted(o["rank"] for o in outs) == [0, 1]
    _check(outs)


def test_three_ranks_many_passes_gloo():
    """uneven shards (n and m not divisible by 3: the records' layouts differ, the fold takes the per-record path) and
    2^16-term passes, so that every shard runs several passes whose column sums are added on the device"""
    outs = _launch(3, "gloo", False, {"C25519_MSM_PASS_LOG2": "16", "C25519_VERIFY_PASS_LOG2": "15", "TEST_N": str((1 << 19) + 5), "TEST_M": "70001"})
    _check(outs)


@pytest.mark.parametrize("n,m", [(1003, 77), (5, 5)])
def test_eight_ranks_on_one_gpu_real_engine_gloo(n, m):
    """world_size 8 -- the size BASELINE configs[3] is quoted on -- with the real engine on every rank, all sharing cuda:0 over gloo:
    the index arithmetic of multi.shard_range / gather_transcript_inputs / the record fold at eight ranks.  (1003, 77): tiny RAGGED shards
    (125 / 126 terms, 9 / 10 signatures), the undecodable point on the last rank (NONE everywhere), seven EMPTY shards beside one full one;
    (5, 5): more ranks than units -- ranks 5..7 hold nothing in the MSM, in both verify_batch z-modes and in the ONE transcript."""
    outs = _launch(8, "gloo", False, {"TEST_N": str(n), "TEST_M": str(m)})
    assert sorted(o["rank"] for o in outs) == list(range(8))
    _check(outs)


def test_rccl_collectives_execute_world_1():
    """backend nccl (= RCCL), world_size 1, force_collective: all_gather_into_tensor on the device records (uint8), on the
    shard sizes (int64) and all_reduce(MAX) on the verdict (int32) really run in RCCL."""
    outs = _launch(1, "nccl", True)
    _check(outs)


def test_bench_under_torch_distributed_run_one_gpu():
    """bench.py the way the driver launches it for N > 1, with N = 1: RANK / WORLD_SIZE / MASTER_* in the environment, so the
    use_dist branch runs (RCCL process group, barrier, forced all_gather of the record in every step, max-over-ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--log2n", "21", "--no-sub", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["ranks_seen_by_rccl"] == 1 and j["collective_executed_per_step"] is True
    assert j["value"] > 1e8 and j["roofline"]["bound"] == "valu_int_mac" and 0 < j["roofline"]["frac"] < 1
    # the N > 1 line explains itself: shard / collective / read-back + fold per step (max over ranks), summing to the step within 5 %, the ranks RCCL
    # saw, the warm-up collectives outside the timed region and the model's prediction beside the measured step
    mg = j["multi_gpu"]
    for k in ("shard_ms", "collective_ms", "d2h_fold_ms", "sum_ms", "measured_step_ms", "ranks_seen_by_rccl", "rccl_warmup_collectives_outside_timed_region", "model", "min_over_ranks"):
        assert k in mg, k
    assert mg["ranks_seen_by_rccl"] == 1 and mg["rccl_warmup_collectives_outside_timed_region"] >= 1
    assert abs(mg["shard_ms"] + mg["collective_ms"] + mg["d2h_fold_ms"] - mg["sum_ms"]) <= 0.05 * mg["sum_ms"]
    assert abs(mg["sum_ms"] - mg["measured_step_ms"]) <= 0.10 * mg["measured_step_ms"]
    assert mg["collective_ms"] > 0 and mg["model"]["predicted_step_ms"] > mg["shard_ms"]
