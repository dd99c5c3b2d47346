"""Multi-GPU decomposition of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

* MSM is a sum over terms: rank r takes the contiguous term range shard_range(n, r, world) and runs the whole
  single-GPU pipeline on it, which leaves a fixed-size partial-result RECORD in device memory
  (c25519_msm_partial_record_dev: the window column sums of its terms + counters; nothing waits for the host).  Then
  ONE exchange step: an all_gather of the records (9 KB per rank, device to device over RCCL), one copy to the host and
  one fold (c25519_fold_partial_records: columns added rank by rank, ONE Horner fold) -- the same on every rank.
  RCCL has no elliptic-curve reduction operator, so the "all-reduce of partial sums" is all_gather + fold; the payload
  is latency-bound.
* fixed-base, X25519 and (de)compression are independent units: replicas only, the batch is split with
  shard_range and no collective is involved.
* verify_batch, Z_TRANSCRIPT (the reference's derivation): the z_i come from ONE Merlin transcript over the whole batch
  (batch.rs:168-222) however many ranks share it -- every rank hashes its shard (ed25519_batch_hram_dev), the hram_i and
  s_i are all-gathered (96 bytes per signature), the sequential transcript runs on the host (every rank derives the same
  z_i from the same bytes, which saves the scatter), every rank evaluates its share of the ONE batch equation with its
  z_i (ed25519_verify_batch_record_dev), and the records are all-gathered and folded into the reference's single
  identity check.  Same z_i, same equation, same verdict as the reference on one machine.
* verify_batch, Z_DEVICE: every rank checks its shard as an independent random linear combination (its own z_i); the
  verdict is the worst shard verdict in the reference's precedence, agreed on with a single 4-byte all_reduce(MAX).

force_collective=True makes a world of ONE rank go through the collectives as well (the tests use it to run the RCCL
calls on a single GPU).
"""
import ctypes as C
import threading

import numpy as np

from . import engine as _e


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) of rank among world ranks (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def fold_partials(partials, out_fmt=_e.FMT_EDWARDS_Y):
    """Fold 160-byte partial sums into one encoded point.  Host arithmetic in the C library; needs no GPU."""
    lib = _e.load_library()
    width = {_e.FMT_EDWARDS_Y: 32, _e.FMT_RISTRETTO: 32, _e.FMT_RAW160: 160}[out_fmt]
    out = C.create_string_buffer(width)
    st = lib.c25519_fold_partials(None, b"".join(partials), len(partials), out_fmt, out)
    if st != 0:
        raise _e.EngineError("fold_partials failed with status %d" % st)
    return out.raw


def _dist_state(group, force_collective):
    """-> (dist | None, world): dist is None when no collective has to run"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None, 1
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return None, 1
    return dist, world


_PINNED = {}                     # (numel, dtype) -> page-locked buffer; at most _PINNED_MAX entries of at most _PINNED_BYTES each
_PINNED_MAX, _PINNED_BYTES = 8, 1 << 20
_PINNED_LOCK = threading.Lock()


def _to_host(t):
    """device tensor -> numpy array through a cached page-locked buffer (one asynchronous copy + one stream synchronisation; a
    plain .cpu() stages through pageable memory, which costs tens of microseconds on a 9 KB record).  The cache is bounded (the record
    sizes of a job are few: world x 9 KB and the small int64 / int32 vectors; anything else, or larger, takes .cpu()) and held under a lock
    from the copy until the bytes have left the buffer, so concurrent gathers of one size never see each other's data."""
    import torch
    if not t.is_cuda:
        return t.numpy()
    key = (t.numel(), t.dtype)
    if t.numel() * t.element_size() > _PINNED_BYTES:
        return t.reshape(-1).cpu().numpy()
    with _PINNED_LOCK:
        buf = _PINNED.get(key)
        if buf is None:
            if len(_PINNED) >= _PINNED_MAX:
                return t.reshape(-1).cpu().numpy()
            buf = _PINNED[key] = torch.empty((t.numel(),), dtype=t.dtype, pin_memory=True)
        buf.copy_(t.reshape(-1), non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return buf.numpy().copy()


def all_gather_rows(local_t, group=None, force_collective=False):
    """The exchange step: every rank's 1-D uint8 tensor (same length everywhere) -> (world, length) numpy array on the
    host, identical on every rank.  With the nccl backend the payload goes device to device (RCCL) and visits the host
    once, after the collective; with gloo it is copied to the host first."""
    import torch
    dist, world = _dist_state(group, force_collective)
    if dist is None:
        return _to_host(local_t.detach()).reshape(1, -1)
    width = local_t.numel()
    if dist.get_backend(group) == "nccl":
        assert local_t.is_cuda, "the nccl backend exchanges device tensors"
        allp = torch.empty((world * width,), dtype=torch.uint8, device=local_t.device)
        dist.all_gather_into_tensor(allp, local_t.contiguous(), group=group)
    else:
        allp = torch.empty((world * width,), dtype=torch.uint8)
        dist.all_gather_into_tensor(allp, local_t.detach().cpu().contiguous(), group=group)
    return _to_host(allp).reshape(world, width)


def gather_fold(partial160, out_fmt=_e.FMT_EDWARDS_Y, group=None, device=None, force_collective=False):
    """Exchange + fold for a participant that holds its partial sum as a 160-byte point on the HOST (e.g. from the
    host-pointer entry points): the point is packed into a record and takes the same path as the device records."""
    import torch
    rec = torch.frombuffer(bytearray(_e.partial_record_pack(partial160)), dtype=torch.uint8)
    dist, _ = _dist_state(group, force_collective)
    if dist is not None and dist.get_backend(group) == "nccl":
        rec = rec.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    st, out = _e.fold_partial_records(all_gather_rows(rec, group, force_collective), out_fmt)
    assert st == _e.OK
    return out


class StepTimes:
    """Where a sharded step spends its time, accumulated over calls (bench.py --gpus N): `shard` = this rank's pipeline on its GPU (HIP events on
    the stream the library launches on), `collective` = the all_gather of the records (events around it on the same stream with the nccl backend --
    RCCL orders its work against the caller's stream; host clock with gloo), `d2h_fold` = what is left of the call's wall-clock: the read-back of the
    gathered records, the host fold (c25519_fold_partial_records) and launch latency.  The three add up to the wall-clock of the calls by
    construction; none of the measurements synchronises anything the call does not synchronise anyway (the events are read after the read-back)."""

    def __init__(self):
        self.calls, self.shard_ms, self.collective_ms, self.d2h_fold_ms, self.wall_ms = 0, 0.0, 0.0, 0.0, 0.0

    def mean(self):
        k = max(self.calls, 1)
        return {"shard_ms": self.shard_ms / k, "collective_ms": self.collective_ms / k, "d2h_fold_ms": self.d2h_fold_ms / k, "sum_ms": self.wall_ms / k, "calls": self.calls}


def msm_vartime_sharded(eng, scalars_t, points_t, in_fmt=_e.FMT_RAW160, out_fmt=_e.FMT_EDWARDS_Y, group=None, force_collective=False, times=None):
    """scalars_t / points_t: THIS rank's shard, already on its GPU.  -> (status, bytes | None), the same on every rank.
    status NONE if any rank saw a point that does not decompress (the counters ride in the records).  ONE collective per
    call, and this rank's result reaches the host only as part of the gathered records.  times: a StepTimes to add this call to."""
    if times is None:
        rec = eng.msm_partial_record_t(scalars_t, points_t, in_fmt)
        return _e.fold_partial_records(all_gather_rows(rec, group, force_collective), out_fmt)
    import time
    import torch
    on_gpu = bool(getattr(scalars_t, "is_cuda", False))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if on_gpu else None
    t0 = time.perf_counter()
    if on_gpu:
        ev[0].record()
    rec = eng.msm_partial_record_t(scalars_t, points_t, in_fmt)
    t1 = time.perf_counter()
    dist, _ = _dist_state(group, force_collective)
    device_collective = on_gpu and dist is not None and dist.get_backend(group) == "nccl"
    if on_gpu:
        ev[1].record()
    if device_collective:
        # the collective alone, between two events on the caller's stream; the read-back comes after the second one
        world = dist.get_world_size(group)
        allp = torch.empty((world * rec.numel(),), dtype=torch.uint8, device=rec.device)
        dist.all_gather_into_tensor(allp, rec.contiguous(), group=group)
        ev[2].record()
        rows = _to_host(allp).reshape(world, rec.numel())
        t2 = t1
    else:
        if on_gpu:
            ev[2].record()
            ev[1].synchronize()                                          # (the host-side exchange starts by copying the record to the host, which waits for the shard anyway:
            #                                                               waiting here keeps the shard's GPU time out of the collective's host clock)
        c0 = time.perf_counter()
        rows = all_gather_rows(rec, group, force_collective)          # gloo / no group: host clock (includes the copy of this rank's record to the host)
        t2 = t1 + (time.perf_counter() - c0)
    out = _e.fold_partial_records(rows, out_fmt)
    wall = (time.perf_counter() - t0) * 1e3
    if on_gpu:
        ev[2].synchronize()
        shard = ev[0].elapsed_time(ev[1])
        coll = ev[1].elapsed_time(ev[2]) if device_collective else (t2 - t1) * 1e3
        # (ev[0] is processed when the GPU reaches it: the host's launch latency before the first kernel is inside `shard`; whatever of the wall-clock the
        #  two parts do not cover -- read-back, fold, Python -- is d2h_fold)
        if shard + coll > wall:
            shard = max(wall - coll, 0.0)
    else:
        shard, coll = (t1 - t0) * 1e3, (t2 - t1) * 1e3
    if not device_collective and dist is None:
        coll = 0.0                                                   # no collective ran: the read-back belongs to d2h_fold
    times.calls += 1
    times.shard_ms += shard; times.collective_ms += coll; times.d2h_fold_ms += max(wall - shard - coll, 0.0); times.wall_ms += wall
    return out


# the reference's error precedence (batch.rs:208-211 before :244-250; a key that does not decode never reaches
# verify_batch, verifying.rs:167) as a rank: the batch verdict is the maximum over the shards
_VERDICT_RANK = {_e.OK: 0, _e.VERIFY: 1, _e.SCALAR_FORMAT: 2, _e.NONE: 3}
_RANK_VERDICT = {v: k for k, v in _VERDICT_RANK.items()}


def combine_verdicts(status, group=None, device=None, force_collective=False):
    """all_reduce(MAX) of this rank's shard verdict in precedence order -> the batch verdict on every rank."""
    import torch
    dist, _ = _dist_state(group, force_collective)
    if dist is None:
        return status
    backend = dist.get_backend(group)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    t = torch.tensor([_VERDICT_RANK[status]], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return _RANK_VERDICT[int(t.item())]


def gather_transcript_inputs(hram_t, sigs_t, group=None, force_collective=False):
    """Every rank's H(R||A||M) (device tensor from Engine.batch_hram_t, trailer excluded here) and s_i -> the whole batch's
    (N, 64) hram and (N, 64) signature arrays on the host (only the s half of a signature is exchanged; the R half of the
    returned rows is zero), in rank order, plus this rank's offset into them.  Two collectives: the shard sizes, then the
    padded 96-byte rows."""
    import torch
    n = sigs_t.shape[0]
    dist, world = _dist_state(group, force_collective)
    pay = torch.cat([hram_t[:n * 64].reshape(n, 64), sigs_t.reshape(n, 64)[:, 32:]], dim=1).contiguous()      # (n, 96)
    if dist is None:
        rows = pay.cpu().numpy()
        sizes, lo = [n], 0
    else:
        nccl = dist.get_backend(group) == "nccl"
        dev = sigs_t.device if nccl else torch.device("cpu")
        cnt = torch.tensor([n], dtype=torch.int64, device=dev)
        allc = torch.empty((world,), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, cnt, group=group)
        sizes = [int(v) for v in allc.cpu().tolist()]
        nmax = max(max(sizes), 1)
        padded = torch.zeros((nmax * 96,), dtype=torch.uint8, device=pay.device)
        padded[:n * 96] = pay.reshape(-1)
        g = all_gather_rows(padded, group, force_collective).reshape(world, nmax, 96)
        rows = np.concatenate([g[r, :sizes[r]] for r in range(world)], axis=0) if sum(sizes) else np.zeros((0, 96), np.uint8)
        lo = sum(sizes[:dist.get_rank(group)])
    total = rows.shape[0]
    hram = np.ascontiguousarray(rows[:, :64])
    sigs = np.zeros((total, 64), dtype=np.uint8)
    sigs[:, 32:] = rows[:, 64:]
    return hram, sigs, lo


def verify_batch_sharded(eng, msgs_t, msg_off_t, sigs_t, pks_t, z_mode=_e.Z_TRANSCRIPT, pk_points=None, group=None, force_collective=False):
    """THIS rank's shard of the batch (device tensors, as Engine.verify_batch_t) -> the verdict of the whole batch, the
    same on every rank.  Z_TRANSCRIPT: the reference's z_i and its single equation over the whole batch (module
    docstring); Z_DEVICE: independent shard checks."""
    import torch
    if z_mode != _e.Z_TRANSCRIPT:
        st = eng.verify_batch_t(msgs_t, msg_off_t, sigs_t, pks_t, z_mode, pk_points=pk_points)
        return combine_verdicts(st, group, sigs_t.device, force_collective)
    n = sigs_t.shape[0]
    hram_t = eng.batch_hram_t(msgs_t, msg_off_t, sigs_t, pks_t)
    hram_all, sigs_all, lo = gather_transcript_inputs(hram_t, sigs_t, group, force_collective)
    z_all = _e.batch_transcript_zs(hram_all, sigs_all)                      # sequential, one host core (the reference's algorithm)
    z_t = torch.from_numpy(np.ascontiguousarray(z_all[lo:lo + n])).to(sigs_t.device)
    rec = eng.verify_batch_record_t(sigs_t, pks_t, hram_t, z_t, pk_points=pk_points)
    return _e.fold_verify_records(all_gather_rows(rec, group, force_collective))
