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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="fixed_base", choices=sorted(ALGO))
    ap.add_argument("--log2n", type=int, default=None, help="units per GPU = 2^log2n (default: the BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import curve25519_dalek_amd as pkg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    eng = pkg.Engine(local_rank)

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
    else:
        raise SystemExit("workload %s: not wired into bench.py yet" % wl)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- parity spot-check against the oracle, outside the timed region (rank 0) ----------------
    cpu_baseline = None
    run()
    torch.cuda.synchronize(dev)
    if rank == 0:
        from oracle import orc
        idx = torch.randperm(n, device=dev, generator=gen)[:1024].cpu().numpy()
        if wl == "fixed_base":
            want = orc.mul_base_compress_batch(scalars[idx].cpu().numpy(), threads=os.cpu_count() or 1)
        else:
            want = orc.x25519_batch(ks[idx].cpu().numpy(), us[idx].cpu().numpy(), threads=os.cpu_count() or 1)
        if not np.array_equal(out[idx].cpu().numpy(), want):
            raise SystemExit("PARITY FAILURE: GPU output differs from the oracle")

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant-kernel duration: HIP events recorded inside the library on the launch stream,
    # averaged over the timed steps (ring of the last 64 calls)
    k = min(args.steps, 64)
    dom_ms = sum(eng.phase_ms(b, 0) for b in range(k)) / k
    rest_ms = sum(eng.phase_ms(b, 1) for b in range(k)) / k

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import orc
        cores = os.cpu_count() or 1
        probe = 2048 * cores
        if wl == "fixed_base":
            a = scalars[:probe].cpu().numpy()
            f = lambda m: orc.mul_base_compress_batch(a[:m] if m <= probe else np.resize(a, (m, 32)), threads=cores)
        else:
            a, b = ks[:probe].cpu().numpy(), us[:probe].cpu().numpy()
            f = lambda m: orc.x25519_batch(np.resize(a, (m, 32)), np.resize(b, (m, 32)), threads=cores)
        c0 = time.perf_counter(); f(probe); c1 = time.perf_counter() - c0
        m = int(max(probe, min(n, probe * 12.0 / max(c1, 1e-3))))     # ~12 s of wall-clock work
        c0 = time.perf_counter(); f(m); c1 = time.perf_counter() - c0
        cpu_baseline = {"value": m / c1, "unit": ALGO[wl]["unit"], "cores": cores, "kind": "port",
                        "sample": "%d units of the same workload, C restatement of the reference serial_u64 path (oracle/), %d threads, %.1f s" % (m, cores, c1)}

    if rank == 0:
        units = float(n) * world * args.steps
        algo_bytes = ALGO[wl]["bytes"] * n
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else None
        res = {
            "metric": ALGO[wl]["metric"], "value": units / dt, "unit": ALGO[wl]["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 limbs (radix 2^25.5), u64 accumulators", "data": "synthetic",
            "config": {"workload": "%s: 2^%d units per GPU, inputs resident in HBM, canonical 32-byte outputs" % (wl, log2n),
                       "units_per_gpu": n, "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": None,
                         "dominant_kernel_ms": dom_ms, "other_kernels_ms": rest_ms,
                         "note": "integer (VALU v_mad_u64_u32) bound kernel: HBM fraction is tiny by construction; see DESIGN.md"},
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
