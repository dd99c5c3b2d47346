#!/usr/bin/env python3
"""bench.py -- one JSON line per run (driver contract).

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM when the timed region starts.  Default workload = BASELINE.json configs[1]:
    2^20 fixed-base scalar multiplications (EdwardsBasepointTable . scalar) -> CompressedEdwardsY
Other configs are selectable with --workload (x25519 | msm | verify) for DESIGN.md's tables; they are
parity-test cases, not the driver's bench line.

N > 1: one process per GPU (torchrun), units sharded across ranks with no data-path collective for
the replicated workloads ("weak" scaling: per-GPU work is fixed); the MSM workload exchanges one
160-byte partial point per rank (all_gather over RCCL) and folds.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic bytes and integer multiply-accumulates per unit
ALGO = {
    "fixed_base": {"bytes": 64, "unit": "scalar-mults/s", "metric": "fixed-base scalar mults/sec (2^20 EdwardsBasepointTable*scalar, compressed out)"},
    "x25519": {"bytes": 96, "unit": "ladders/s", "metric": "X25519 key agreements/sec (2^20 Montgomery ladders)"},
    "msm": {"bytes": 192, "unit": "terms/s", "metric": "MSM terms/sec (variable-base Pippenger)"},
    "verify": {"bytes": 128, "unit": "verifies/s", "metric": "Ed25519 batch verifies/sec (verify_batch)"},
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# The roof that actually binds: 32x32->64 multiply-accumulate issue (v_mad_u64_u32).
# Field work per unit as implemented, counted from the kernels: M = fe_mul = 100 MACs, S = fe_sq = 55 MACs;
# "ref" = the reference algorithm's count from SURVEY.md 8d (M = 100, S = 60 in its 5x51 schoolbook terms).
VALU = {
    "fixed_base": {"M": 112, "S": 16, "ref": 47100, "what": "1M (first window) + 15 madd x 7M (radix-2^16 tables in HBM) + 5M compress + 1/16 inversion"},
    "fixed_base_comb": {"M": 239, "S": 32, "ref": 47100, "what": "31 madd x 7M + 4 dbl x (4S+4M) (LDS comb) + 5M compress + 1/16 inversion"},
    "x25519": {"M": 1303, "S": 1036, "ref": 231000, "what": "255 x (5M + 4S + 10-product a24 mul) + 3M + 1/16 inversion"},
    "msm": {"M": 121, "S": 0, "ref": 26500, "what": "16 windows x 7M bucket adds + 8M normalise + ~1M reduce (c = 16)"},
    "verify": {"M": 197, "S": 255, "ref": 57000, "what": "decompress R (255S + 21M) + normalise A (8M) + (16 + 8) windows x 7M; SHA-512 and scalar muls not counted"},
    "verify_bytes": {"M": 210, "S": 510, "ref": 74400, "what": "decompress R and A (2 x (255S + 21M)) + (16 + 8) windows x 7M; SHA-512 and scalar muls not counted"},
}


def host_cores():
    """CPUs this process may actually use (affinity mask and cgroup quota), not the box's core count."""
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            c = max(1, min(c, int(float(q[0]) / float(q[1]))))
    except Exception:
        pass
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="fixed_base", choices=sorted(ALGO))
    ap.add_argument("--log2n", type=int, default=None, help="units per GPU = 2^log2n (default: the BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--keys-as-bytes", action="store_true",
                    help="verify: pass only the 32-byte keys (A_i is decompressed inside the call); default: the keys' points "
                         "are cached like the reference's VerifyingKey (verifying.rs:64-71, batch.rs:236)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import curve25519_dalek_amd as pkg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # launched by torch.distributed.run
    torch.cuda.set_device(local_rank)
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    eng = pkg.Engine(local_rank, window=int(os.environ.get("C25519_WINDOW", "0")))

    wl = args.workload
    log2n = args.log2n if args.log2n is not None else {"fixed_base": 20, "x25519": 20, "msm": 21, "verify": 20}[wl]
    n = 1 << log2n
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC25519 + rank)

    def rnd(rows, width=32):
        return torch.randint(0, 256, (rows, width), dtype=torch.uint8, device=dev, generator=gen)

    run = None
    if wl == "fixed_base":
        scalars = rnd(n)
        scalars[:, 31] &= 0x0F                     # uniform in [0, 2^252): reduced scalars
        out = torch.empty((n, 32), dtype=torch.uint8, device=dev)

        def run():
            eng.mul_base_batch_t(scalars, pkg.engine.FMT_EDWARDS_Y, out)
    elif wl == "x25519":
        ks, us = rnd(n), rnd(n)
        out = torch.empty((n, 32), dtype=torch.uint8, device=dev)

        def run():
            eng.x25519_batch_t(ks, us, out)
    elif wl == "msm":
        # config 4 shape: P_i = y_i * B generated on the device (the host never materialises the points)
        xs, ys = rnd(n), rnd(n)
        xs[:, 31] &= int(os.environ.get("C25519_BENCH_TOPMASK", "0x0F"), 16); ys[:, 31] &= 0x0F
        pts = eng.mul_base_batch_t(ys, pkg.engine.FMT_RAW160)
        result = {}

        def run():
            # partial sum on this GPU, then the one exchange step (160 B per rank over RCCL) + fold
            st, out = pkg.multi.msm_vartime_sharded(eng, xs, pts, pkg.engine.FMT_RAW160, pkg.engine.FMT_EDWARDS_Y)
            assert st == 0
            result["out"] = out
    elif wl == "verify":
        # inputs: 2^k independent keypairs and 32-byte messages, signed by the engine's own batched signer
        # (byte-exact against the reference's TESTVECTORS in tests/test_gpu_single.py; not on the measured path)
        seeds, d_msgs = rnd(n), rnd(n).reshape(-1)
        d_off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
        d_pks, d_sigs = eng.sign_batch_t(seeds, d_msgs, d_off)
        d_pk_points = None
        if not args.keys_as_bytes:
            _, d_pk_points, ok = eng.decompress_batch_t(d_pks)       # VerifyingKey::from_bytes, done once per key
            assert bool(ok.all())
        result = {}

        def run():
            result["st"] = eng.verify_batch_t(d_msgs, d_off, d_sigs, d_pks, pkg.engine.Z_DEVICE, pk_points=d_pk_points)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- sanity outside the timed region.  The oracle (test infrastructure) is used ONLY in the cpu_baseline leg:
    # there its outputs on the sample are also compared with the GPU's.  Here: product-only checks. -------------
    cpu_baseline = None
    use_oracle = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    run()
    torch.cuda.synchronize(dev)
    if wl == "verify" and result["st"] != 0:
        raise SystemExit("SELF-CHECK FAILURE: valid batch rejected (status %d)" % result["st"])
    if wl == "msm" and world == 1:
        # (sum x_i y_i mod l) * B through the engine's fixed-base kernel must equal the MSM over P_i = y_i * B
        L = 2**252 + 27742317777372353535851937790883648493
        def limbs(t):
            return (t.view(torch.int16).to(torch.int64) & 0xFFFF)
        ax, ay = limbs(xs), limbs(ys)
        acc = 0
        for j in range(16):
            col = (ax[:, j:j + 1] * ay).sum(0).cpu().tolist()           # exact: each entry < 2^21 * 2^32
            for k in range(16):
                acc += int(col[k]) << (16 * (j + k))
        want = eng.mul_base_batch(np.frombuffer((acc % L).to_bytes(32, "little"), np.uint8).reshape(1, 32))[0].tobytes()
        if result["out"] != want:
            raise SystemExit("SELF-CHECK FAILURE: MSM result differs from (sum x_i y_i) B")

    # live peak of the binding unit on THIS box (box-to-box spread is ~6 %): v_mad_u64_u32 issue rate, measured right
    # before the warmup steps by the library's own probe kernel (c25519_microbench, kernels.hip), best of 100 runs.
    # The ~60 ms of full-rate integer work also bring the GPU from its idle clock to the sustained one (measured: the
    # probe reads 27 T/s cold and 33 T/s after ~50 ms; a 0.8 ms step is 15 % slower on a cold clock), so a short
    # --steps measures steady-state throughput, not the ramp, and `valu.peak` is taken in the same clock state.
    mac_peak = max(eng.microbench(0, 4000) for _ in range(int(os.environ.get("C25519_BENCH_PROBES", "100")))) * 1e9
    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant-kernel duration: HIP events recorded inside the library on the launch stream,
    # averaged over the timed steps (ring of the last 64 calls)
    k = min(args.steps, 64)
    dom_ms = sum(eng.phase_ms(b, 0) for b in range(k)) / k
    rest_ms = sum(eng.phase_ms(b, 1) for b in range(k)) / k

    if use_oracle:
        from oracle import orc
        cores = host_cores()
        # parity of the GPU outputs against the oracle on the sample it is about to be timed on
        if wl in ("fixed_base", "x25519"):
            idx = torch.randperm(n, device=dev, generator=gen)[:1024].cpu().numpy()
            if wl == "fixed_base":
                want = orc.mul_base_compress_batch(scalars[idx].cpu().numpy(), threads=cores)
            else:
                want = orc.x25519_batch(ks[idx].cpu().numpy(), us[idx].cpu().numpy(), threads=cores)
            if not np.array_equal(out[idx].cpu().numpy(), want):
                raise SystemExit("PARITY FAILURE: GPU output differs from the oracle")
        elif wl == "msm":
            m = 2048
            want = orc.ed_compress(orc.ed_msm([xs[i].cpu().numpy().tobytes() for i in range(m)], [pts[i].cpu().numpy().tobytes() for i in range(m)]))
            st, got = eng.msm_vartime_t(xs[:m].contiguous(), pts[:m].contiguous(), pkg.engine.FMT_RAW160, pkg.engine.FMT_EDWARDS_Y)
            if st != 0 or got != want:
                raise SystemExit("PARITY FAILURE: MSM result differs from the oracle's")
        else:
            mh = d_msgs.reshape(n, 32).cpu().numpy(); sig_h = d_sigs.cpu().numpy(); pk_h = d_pks.cpu().numpy()
            for i in range(0, n, n // 64):
                if orc.ed25519_verify(pk_h[i].tobytes(), mh[i].tobytes(), sig_h[i].tobytes()) != 0:
                    raise SystemExit("PARITY FAILURE: the oracle rejects a signature the engine accepts")
        if wl in ("fixed_base", "x25519"):
            # embarrassingly parallel in the reference too: one slice per host core
            probe = min(n, 1024 * cores)
            if wl == "fixed_base":
                a = scalars[:probe].cpu().numpy()
                f = lambda m: orc.mul_base_compress_batch(np.resize(a, (m, 32)), threads=cores)
            else:
                a, b = ks[:probe].cpu().numpy(), us[:probe].cpu().numpy()
                f = lambda m: orc.x25519_batch(np.resize(a, (m, 32)), np.resize(b, (m, 32)), threads=cores)
            used = cores
        elif wl == "msm":
            # the reference's MSM is one single-threaded call (Pippenger, w = 8): time it as such
            probe = 4096
            xa = xs[:1 << 17].cpu().numpy(); pa = pts[:1 << 17].cpu().numpy()
            f = lambda m: orc.ed_msm([xa[i].tobytes() for i in range(m)], [pa[i].tobytes() for i in range(m)])
            used = 1
        else:
            probe = 2048
            f = lambda m: orc.ed25519_verify_batch([mh[i].tobytes() for i in range(m)], [sig_h[i].tobytes() for i in range(m)], [pk_h[i].tobytes() for i in range(m)])
            used = 1
        f(min(probe, 256))                                              # warm caches / tables
        c0 = time.perf_counter(); f(probe); c1 = time.perf_counter() - c0
        cap = (16 * n) if wl in ("fixed_base", "x25519") else (n if wl == "verify" else (1 << 17))
        m = int(max(probe, min(cap, probe * 10.0 / max(c1, 1e-4))))     # ~10 s of work
        c0 = time.perf_counter(); f(m); c1 = time.perf_counter() - c0
        cpu_baseline = {"value": m / c1, "unit": ALGO[wl]["unit"], "cores": used, "kind": "port",
                        "sample": "%d units of the same workload through the C restatement of the reference serial_u64 path (oracle/), %d thread(s), %.1f s; host exposes %d usable cores" % (m, used, c1, cores)}

    traffic = None
    traffic_src = None
    if rank == 0:
        # HBM bytes per launch of the dominant kernel from the committed PMC profile of this workload
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/profile_all.sh; FETCH_SIZE doubled per the
        # gfx950 correction in MI355X_MICROARCH.md).  null if no profile has been committed yet.
        dom_kernel = {"fixed_base": "k_mul_base_comb" if os.environ.get("C25519_WINDOW", "0") == "9" else "k_mul_base_wide", "x25519": "k_x25519", "msm": "k_accumulate", "verify": "k_accumulate"}[wl]
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc.txt" % wl)), reverse=True):
            fetch = write = None
            cur = None
            for line in open(f):
                if not line.startswith(" "):
                    cur = line.strip()
                elif cur and dom_kernel in cur:
                    parts = line.split()
                    if parts[0] == "FETCH_SIZE":
                        fetch = float(parts[1])
                    if parts[0] == "WRITE_SIZE":
                        write = float(parts[1])
            if fetch is not None and write is not None:
                traffic = (2.0 * fetch + write) * 1024.0
                traffic_src = os.path.relpath(f, ROOT)
                break
    if rank == 0:
        units = float(n) * world * args.steps
        algo_bytes = ALGO[wl]["bytes"] * n
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else None
        res = {
            "metric": ALGO[wl]["metric"], "value": units / dt, "unit": ALGO[wl]["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 limbs (radix 2^25.5), u64 accumulators", "data": "synthetic",
            "config": {"workload": "%s: 2^%d units per GPU, inputs resident in HBM, canonical 32-byte outputs%s" % (
                wl, log2n, "" if wl != "verify" else (", keys as 32 bytes (decompressed inside)" if args.keys_as_bytes else ", keys = VerifyingKey (bytes + cached point), as in the reference")),
                       "units_per_gpu": n, "parallelism": ("sharded terms, all_gather of 160-B partials x%d" if wl == "msm" else "replicas x%d") % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": algo_bytes,
                         "dominant_kernel_ms": dom_ms, "other_kernels_ms": rest_ms,
                         "note": "integer (VALU v_mad_u64_u32) bound kernel: HBM fraction is tiny by construction; see DESIGN.md"},
            "cpu_baseline": cpu_baseline,
        }
        v = VALU["fixed_base_comb" if (wl == "fixed_base" and os.environ.get("C25519_WINDOW", "0") == "9") else
                 "verify_bytes" if (wl == "verify" and args.keys_as_bytes) else wl]
        mac_impl = 100 * v["M"] + 55 * v["S"]
        per_gpu = units / dt / world
        res["valu"] = {"bound": "v_mad_u64_u32 issue", "mac_per_unit_implemented": mac_impl, "mac_per_unit_reference": v["ref"],
                       "field_ops_per_unit": "%d M + %d S: %s" % (v["M"], v["S"], v["what"]),
                       "achieved": per_gpu * mac_impl / 1e12, "peak": mac_peak / 1e12, "unit": "TMAC/s (per GPU)",
                       "frac": per_gpu * mac_impl / mac_peak, "peak_source": "c25519_microbench(0) on this GPU, this run"}
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
