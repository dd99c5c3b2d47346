#!/usr/bin/env python3
"""bench.py -- one JSON line per run (driver contract).

Headline (the top-level `value`): BASELINE.json's metric, MSM scalar-mults/s, on the configuration it is quoted on --
configs[3], a 2^24-term variable-base Pippenger MSM -- as ONE c25519_msm_vartime call per step with scalars and raw
points already resident in HBM.  At N = 1 the 2^24 terms run on one GPU; at N > 1 (`--gpus N`) the same 2^24 terms are
sharded over N ranks, one process per GPU, with the ONE exchange step of the path per step (all_gather of a 160-byte
partial sum per rank over RCCL, then the fold): strong scaling, as configs[3] and north_star state it
(`--scaling weak` keeps 2^24 terms per GPU instead).

`--gpus N` with N > 1 and no RANK in the environment re-executes itself through torch.distributed.run (one rank per
GPU); under a launcher (RANK / WORLD_SIZE set, the driver's way) it just joins.  On a box with fewer than N GPUs it
fails loudly.

At N = 1 the same line carries, as `sub`, the other BASELINE configurations timed the same way (K steps after W warmup
steps each, their own HIP-event kernel timings): configs[2] verify_batch of 2^20 signatures (VerifyingKey mode, device
z-mode; plus the strict-transcript z-mode at 2^14 and 2^20), configs[1] 2^20 fixed-base multiplications (radix-2^16
tables and the constant-time lookups), configs[4] 2^20 X25519 ladders -- each with its own `roofline`, `valu` and
`cpu_baseline` objects -- and, as `ffi_path`, the same workloads through the HOST-POINTER entry points a Rust caller binds
(wall-clock per call, units/s, achieved PCIe GB/s against the link peak).  `--workload X` makes X the headline and drops both.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# The binding unit of every kernel of this path is the 32 x 32 -> 64 bit integer multiply-add (v_mad_u64_u32).  The guide gives
# no figure for it; its theoretical ceiling is one wave64 instruction per SIMD every 4 cycles = 16 lanes per clock per SIMD:
SIMDS = 256 * 4
MAC_LANES_PER_CLK_PER_SIMD = 16
L_ORDER = 2**252 + 27742317777372353535851937790883648493

# algorithmic bytes per unit and the kernel whose launches the roofline object describes
ALGO = {
    "msm": {"bytes": 192, "unit": "terms/s", "metric": "MSM scalar-mults/sec (variable-base Pippenger MSM)", "kernel": "k_accumulate",
            "bytes_what": "32-byte scalar + 160-byte raw point per term"},
    "verify": {"bytes": 128, "unit": "verifies/s", "metric": "Ed25519 batch verifies/sec (verify_batch)", "kernel": "k_prep_compressed",
               "bytes_what": "64-byte signature + 32-byte key + 32-byte message per signature"},
    "fixed_base": {"bytes": 64, "unit": "scalar-mults/s", "metric": "fixed-base scalar mults/sec (EdwardsBasepointTable*scalar, compressed out)", "kernel": "k_mul_base",
                   "bytes_what": "32-byte scalar in + 32-byte CompressedEdwardsY out"},
    "x25519": {"bytes": 96, "unit": "ladders/s", "metric": "X25519 key agreements/sec (Montgomery ladders)", "kernel": "k_x25519",
               "bytes_what": "32-byte scalar + 32-byte u in, 32-byte u out"},
}
DEFAULT_LOG2N = {"msm": 24, "verify": 20, "fixed_base": 20, "x25519": 20}


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


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC profile of this workload (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes, tools/profile_all.sh; FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md)."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc.txt" % workload)), reverse=True):
        fetch = write = None
        cur = None
        for line in open(f):
            if not line.startswith(" "):
                cur = line.strip()
            elif cur and kernel in cur:
                parts = line.split()
                if parts[0] == "FETCH_SIZE":
                    fetch = float(parts[1])
                if parts[0] == "WRITE_SIZE":
                    write = float(parts[1])
        if fetch is not None and write is not None:
            return (2.0 * fetch + write) * 1024.0, os.path.relpath(f, ROOT)
    return None, None


class Workload:
    """One BASELINE configuration: synthetic inputs resident in HBM, run() = one step, checks, kernel timings."""

    def __init__(self, name, eng, pkg, torch, dev, log2n, rank=0, world=1, args=None):
        self.name, self.eng, self.pkg, self.torch, self.dev = name, eng, pkg, torch, dev
        self.rank, self.world, self.args = rank, world, args
        self.log2n, self.n = log2n, 1 << log2n
        self.variant = ""
        self.force_collective = False
        self.step_times = None         # multi.StepTimes while the timed steps of a launcher run are measured (shard / collective / read-back + fold)
        self.result = {}
        gen = torch.Generator(device=dev)
        gen.manual_seed(0xC25519 + 1000 * rank + {"msm": 1, "verify": 2, "fixed_base": 3, "x25519": 4}[name])
        self.gen = gen
        getattr(self, "_setup_" + name)()

    def rnd(self, rows, width=32):
        return self.torch.randint(0, 256, (rows, width), dtype=self.torch.uint8, device=self.dev, generator=self.gen)

    # -- setups -------------------------------------------------------------------------------------------------
    def _setup_msm(self):
        # config 4 shape: P_i = y_i * B generated on the device (the host never materialises the points)
        E = self.pkg.engine
        n = self.n
        self.xs, ys = self.rnd(n), self.rnd(n)
        self.xs[:, 31] &= 0x0F; ys[:, 31] &= 0x0F
        self.ys = ys
        self.pts = self.torch.empty((n, 160), dtype=self.torch.uint8, device=self.dev)
        step = 1 << 21
        for lo in range(0, n, step):
            self.eng.mul_base_batch_vartime_t(ys[lo:lo + step], E.FMT_RAW160, self.pts[lo:lo + step])

        def run():
            # this rank's partial sum, then the one exchange step (160 B per rank over RCCL) + fold
            # (under a launcher the collective runs even at world size 1, so that a one-GPU run exercises RCCL as well)
            st, out = self.pkg.multi.msm_vartime_sharded(self.eng, self.xs, self.pts, E.FMT_RAW160, E.FMT_EDWARDS_Y, force_collective=self.force_collective, times=self.step_times)
            assert st == 0
            self.result["out"] = out
        self.run = run

    def _setup_verify(self):
        # 2^k independent keypairs and 32-byte messages, signed by the engine's own batched signer (byte-exact against the
        # reference's TESTVECTORS in tests/test_gpu_single.py; not on the measured path)
        E = self.pkg.engine
        n = self.n
        seeds, self.d_msgs = self.rnd(n), self.rnd(n).reshape(-1)
        self.d_off = self.torch.arange(0, 32 * (n + 1), 32, dtype=self.torch.int64, device=self.dev)
        self.d_pks, self.d_sigs = self.eng.sign_batch_t(seeds, self.d_msgs, self.d_off)
        self.d_pk_points = None
        self.keys_as_bytes = bool(self.args and self.args.keys_as_bytes)
        if not self.keys_as_bytes:
            _, self.d_pk_points, ok = self.eng.decompress_batch_t(self.d_pks)       # VerifyingKey::from_bytes, done once per key
            assert bool(ok.all())
        self.z_mode = E.Z_TRANSCRIPT if (self.args and self.args.z_mode == "transcript") else E.Z_DEVICE

        def run():
            self.result["st"] = self.eng.verify_batch_t(self.d_msgs, self.d_off, self.d_sigs, self.d_pks, self.z_mode, pk_points=self.d_pk_points)
        self.run = run

    def _setup_fixed_base(self):
        E = self.pkg.engine
        self.scalars = self.rnd(self.n)
        self.scalars[:, 31] &= 0x0F                     # uniform in [0, 2^252): reduced scalars
        self.out = self.torch.empty((self.n, 32), dtype=self.torch.uint8, device=self.dev)

        def run():
            self.eng.mul_base_batch_t(self.scalars, E.FMT_EDWARDS_Y, self.out)
        self.run = run

    def _setup_x25519(self):
        self.ks, self.us = self.rnd(self.n), self.rnd(self.n)
        self.out = self.torch.empty((self.n, 32), dtype=self.torch.uint8, device=self.dev)

        def run():
            self.eng.x25519_batch_t(self.ks, self.us, self.out)
        self.run = run

    # -- product-only sanity outside the timed region (the CPU restatement is used ONLY in cpu_baseline()) -------------
    def self_check(self):
        torch, eng, E = self.torch, self.eng, self.pkg.engine
        self.run()
        torch.cuda.synchronize(self.dev)
        if self.name == "verify" and self.result["st"] != 0:
            raise SystemExit("SELF-CHECK FAILURE: valid batch rejected (status %d)" % self.result["st"])
        if self.name == "msm":
            # (sum x_i y_i mod l) * B through the fixed-base kernel must equal this rank's partial MSM over P_i = y_i * B
            def limbs(t):
                return (t.view(torch.int16).to(torch.int64) & 0xFFFF)
            acc = 0
            step = 1 << 21
            for lo in range(0, self.n, step):
                ax, ay = limbs(self.xs[lo:lo + step]), limbs(self.ys[lo:lo + step])
                for j in range(16):
                    col = (ax[:, j:j + 1] * ay).sum(0).cpu().tolist()           # exact: each entry < 2^21 * 2^32
                    for k in range(16):
                        acc += int(col[k]) << (16 * (j + k))
            import numpy as np
            want = eng.mul_base_batch(np.frombuffer((acc % L_ORDER).to_bytes(32, "little"), np.uint8).reshape(1, 32))[0].tobytes()
            st, part = eng.msm_partial_t(self.xs, self.pts, E.FMT_RAW160)
            got = self.pkg.multi.fold_partials([part], E.FMT_EDWARDS_Y)
            if st != 0 or got != want:
                raise SystemExit("SELF-CHECK FAILURE: MSM result differs from (sum x_i y_i) B")
            self.sum_xy, self.msm_enc = (acc % L_ORDER).to_bytes(32, "little"), got      # for the oracle comparison of the cpu_baseline leg
            del self.ys

    # -- kernel timings from the HIP events the library records on the launch streams -----------------------------------
    def kernel_times(self, steps):
        """-> dict: dominant kernel ms per LAUNCH (averaged over the timed steps), launches per step, other figures.  Durations
        are HIP events the library records on its launch streams; the kernel names are the ones it launched
        (c25519_last_kernel_name), not literals of this file."""
        eng = self.eng
        lib = self.pkg.load_library()
        lib.c25519_last_kernel_name.restype = __import__("ctypes").c_char_p
        lib.c25519_last_kernel_name.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_int]
        kname = lambda which: lib.c25519_last_kernel_name(eng.ctx, which).decode()
        if self.name in ("msm", "verify"):
            # the latest call's passes (the timed steps are identical; the ring keeps per-pass events)
            acc_ms, passes = eng.last_call_phase_ms(0)
            if passes == 0 or acc_ms < 0:
                # the small and the mid path record no per-kernel events (an event record between two kernels is a ~5 us gap on the GPU): the whole call's span instead
                span = eng.last_kernel_ms()
                return {"passes_per_step": 1, "k_accumulate_ms_per_launch": None, "reduce_tail_ms_per_pass": None, "pass_span_ms": span, "dominant_ms": span,
                        "dominant_kernel": "(whole call: " + kname(0) + " and the kernels around it)"}
            tail_ms, _ = eng.last_call_phase_ms(1)
            pass_ms, _ = eng.last_call_phase_ms(2)
            out = {"passes_per_step": passes, "k_accumulate_ms_per_launch": acc_ms / passes, "reduce_tail_ms_per_pass": tail_ms / passes,
                   "pass_span_ms": pass_ms / passes}
            if self.name == "verify":
                dec_ms, _ = eng.last_call_phase_ms(3)
                out["k_prep_compressed_R_ms_per_launch"] = dec_ms / passes
                out["dominant_ms"] = max(dec_ms, acc_ms) / passes
                out["dominant_is_prep"] = dec_ms >= acc_ms
                out["dominant_kernel"] = kname(1) if dec_ms >= acc_ms else kname(0)
            else:
                out["dominant_ms"] = acc_ms / passes
                out["dominant_kernel"] = kname(0)
            return out
        k = min(steps, 64)
        dom = sum(eng.phase_ms(b, 0) for b in range(k)) / k
        rest = sum(eng.phase_ms(b, 1) for b in range(k)) / k
        return {"passes_per_step": 1, "dominant_ms": dom, "other_kernels_ms": rest, "dominant_kernel": kname(0)}

    def units_per_launch(self, kt):
        return float(self.n) / kt["passes_per_step"]

    def cost(self):
        from curve25519_dalek_amd import costs
        import ctypes as C
        if self.name == "fixed_base":
            return costs.fixed_base_ct() if self.variant == "ct" else (costs.fixed_base_comb() if self.variant == "comb" else costs.fixed_base_wide(16)), costs.reference_mac["fixed_base"]
        if self.name == "x25519":
            return costs.x25519(), costs.reference_mac["x25519"]
        lib = self.pkg.load_library()
        c = C.c_int32(); nwin = C.c_int32()
        pos = (C.c_uint8 * 56)(); wid = (C.c_uint8 * 56)(); addk = (C.c_uint32 * 8)()
        passes = max(1, self.kt["passes_per_step"]) if hasattr(self, "kt") else 1
        per = -(-self.n // passes)
        terms = per if self.name == "msm" else 2 * per + 1
        if self.name == "msm" and passes > 2 and per >= (1 << 20):
            terms = max(terms, 1 << 21)              # msm_record_enqueue: passes that continue each other's buckets share the layout of a 2^21-term pass
        if self.name != "msm":
            terms = min(terms, (1 << 21) - 1)        # verify_batch keeps 16-bit windows (verify.hip: msm_layout(.., 16)); same window count as any larger 16-bit layout
        lib.c25519_msm_geometry(terms, C.byref(c), C.byref(nwin), pos, wid, addk)
        half = 1 << (c.value - 1)
        self._nwin = nwin.value
        if self.name == "msm":
            return costs.msm(per, nwin.value, half), costs.reference_mac["msm"]
        return costs.verify(per, nwin.value, half, self.keys_as_bytes), costs.reference_mac["verify_bytes" if self.keys_as_bytes else "verify"]

    # -- CPU baseline: the C restatement of the reference's serial_u64 path (test infrastructure), bounded sample --------
    def cpu_baseline(self, budget_s):
        """-> the contract object {value, unit, cores, kind, sample} + `single_thread` and `all_cores` (SURVEY.md 8d asks for both):
        `value` / `cores` are the all-cores leg (the strongest CPU figure), single_thread the reference's own call shape (its MSM and
        verify_batch are single-threaded calls; the all-cores leg cuts the sample into one independent slice per thread)."""
        import numpy as np
        from oracle import orc
        torch, eng, E, n = self.torch, self.eng, self.pkg.engine, self.n
        cores = host_cores()
        name = self.name
        if name in ("fixed_base", "x25519"):
            idx = torch.randperm(n, device=self.dev, generator=self.gen)[:1024].cpu().numpy()
            if name == "fixed_base":
                want = orc.mul_base_compress_batch(self.scalars[idx].cpu().numpy(), threads=cores)
            else:
                want = orc.x25519_batch(self.ks[idx].cpu().numpy(), self.us[idx].cpu().numpy(), threads=cores)
            if not np.array_equal(self.out[idx].cpu().numpy(), want):
                raise SystemExit("PARITY FAILURE: GPU output differs from the CPU restatement (%s)" % name)
            probe = 1024
            if name == "fixed_base":
                a = self.scalars[:1024 * cores].cpu().numpy()
                f = lambda m, t: orc.mul_base_compress_batch(np.resize(a, (m, 32)), threads=t)
            else:
                a, b = self.ks[:1024 * cores].cpu().numpy(), self.us[:1024 * cores].cpu().numpy()
                f = lambda m, t: orc.x25519_batch(np.resize(a, (m, 32)), np.resize(b, (m, 32)), threads=t)
            cap = 16 * n
        elif name == "msm":
            m0 = 2048
            cap = min(n, 1 << 21)
            xa = self.xs[:cap].cpu().numpy(); pa = self.pts[:cap].cpu().numpy()
            want = orc.ed_compress(orc.ed_msm([xa[i].tobytes() for i in range(m0)], [pa[i].tobytes() for i in range(m0)]))
            st, got = eng.msm_vartime_t(self.xs[:m0].contiguous(), self.pts[:m0].contiguous(), E.FMT_RAW160, E.FMT_EDWARDS_Y)
            if st != 0 or got != want:
                raise SystemExit("PARITY FAILURE: MSM result differs from the CPU restatement's")
            if orc.ed_compress(orc.ed_msm_mt_np(xa[:m0], pa[:m0], min(cores, 4))) != want:
                raise SystemExit("PARITY FAILURE: the sliced CPU MSM differs from the single call")
            # the FULL-SIZE result of the headline workload against the oracle: (sum x_i y_i mod l) B by the oracle's own fixed-base multiplication
            # (self_check compared it with the product's fixed-base kernel only)
            if getattr(self, "sum_xy", None) is not None and orc.ed_compress(orc.ed_mul_base(self.sum_xy)) != self.msm_enc:
                raise SystemExit("PARITY FAILURE: the full-size MSM result differs from the oracle's (sum x_i y_i mod l) B")
            probe = 4096
            f = lambda m, t: orc.ed_msm_np(xa[:m], pa[:m]) if t == 1 else orc.ed_msm_mt_np(xa[:m], pa[:m], t)
        else:
            mh = self.d_msgs.reshape(n, 32).cpu().numpy(); sig_h = self.d_sigs.cpu().numpy(); pk_h = self.d_pks.cpu().numpy()
            for i in range(0, n, max(1, n // 64)):
                if orc.ed25519_verify(pk_h[i].tobytes(), mh[i].tobytes(), sig_h[i].tobytes()) != 0:
                    raise SystemExit("PARITY FAILURE: the CPU restatement rejects a signature the engine accepts")
            probe = 2048
            cap = n

            def f(m, t):
                if orc.ed25519_verify_batch_mt_np(mh[:m], 32, sig_h[:m], pk_h[:m], t) != 0:
                    raise SystemExit("PARITY FAILURE: the CPU restatement rejects a batch the engine accepts")

        def leg(threads, seconds):
            f(min(probe, 256) * threads, threads)                           # warm caches / tables
            c0 = time.perf_counter(); f(probe * threads, threads); c1 = time.perf_counter() - c0
            m = int(max(probe * threads, min(cap, probe * threads * seconds / max(c1, 1e-4))))
            c0 = time.perf_counter(); f(m, threads); c1 = time.perf_counter() - c0
            return {"value": m / c1, "threads": threads, "units": m, "seconds": c1}
        one = leg(1, 0.4 * budget_s)
        allc = leg(cores, 0.6 * budget_s) if cores > 1 else one
        return {"value": allc["value"], "unit": ALGO[name]["unit"], "cores": allc["threads"], "kind": "port",
                "sample": "%d units of the same workload through the C restatement of the reference serial_u64 path on %d thread(s) (one independent slice per thread), %.1f s; "
                          "single_thread: %d units, %.1f s; host exposes %d usable cores" % (allc["units"], allc["threads"], allc["seconds"], one["units"], one["seconds"], cores),
                "single_thread": one["value"], "all_cores": allc["value"]}


def time_steps(run, steps, warmup, barrier):
    for _ in range(warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    barrier()
    return time.perf_counter() - t0


NOTES = {
    "roofline": "bound = v_mad_u64_u32 issue (SURVEY.md 8d: these kernels are integer multiply-add bound, not byte bound). achieved = the dominant kernel's "
                "multiply-adds per launch (implemented count, curve25519-dalek_amd/costs.py) / its HIP-event duration on its launch stream; peak = the rate the "
                "library's probe kernel (c25519_microbench(0): 8 independent v_mad_u64_u32 chains, 8 waves per SIMD) reaches on this GPU in this run; "
                "peak_theoretical = 1024 SIMDs x 16 lanes/clk x the nominal clock; whole_call prices every kernel of the step against the same roof",
    "hbm": "roofline.hbm.frac prices the ALGORITHMIC bytes (what the caller hands over) against 8 TB/s: small, because the kernels are multiplier bound. "
           "traffic = what the kernel's L2 requests from the fabric per launch (PMC FETCH_SIZE x 2 + WRITE_SIZE of the committed profile named in traffic_source; "
           "table / point gathers) -- the 256 MiB MALL serves part of it, so traffic_frac (traffic / duration / HBM peak) is an upper bound on the HBM share",
    "cpu_baseline": "kind port: the C restatement of the reference's serial_u64 path (oracle/, test infrastructure) timed on this box's host cores on a bounded sample; "
                    "value / cores = all usable cores (one independent slice per thread), single_thread = the reference's own call shape",
    "ffi_path": "host pointers in, host pointers out (the entry points a Rust caller binds): [ms per call, units/s, (bytes up + down) / wall-clock / 64 GB/s]; pageable numpy "
                "buffers, outputs reused; chunks / passes travel on copy streams while the previous chunk computes (csrc/ffi.h)",
}


def record(w, dt, steps, warmup, world, mac_peak, cpu_baseline, scaling, clock_hz):
    """The JSON object of one workload (numbers only: the prose lives once, in the top-level `notes`)."""
    name = w.name
    kt = w.kt                       # taken right after the timed steps (before the CPU leg makes other calls on the engine)
    cost, ref_mac = w.cost()
    from curve25519_dalek_amd import costs
    mac_impl = costs.mac(cost)
    total_units = float(w.total_terms) if (name == "msm" and scaling == "strong") else float(w.n) * world
    units = total_units * steps
    upl = w.units_per_launch(kt)
    algo_bytes = ALGO[name]["bytes"] * upl
    dom_ms = kt["dominant_ms"]
    hbm_achieved = algo_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms and dom_ms > 0 else None
    # PMC traffic of the kernel the roofline object describes
    pk = {"msm": "k_accumulate", "verify": "k_prep_compressed" if kt.get("dominant_is_prep") else "k_accumulate",
          "fixed_base": "k_mul_base_comb" if w.variant == "comb" else ("k_mul_base_ctp<" if w.variant == "ct" else "k_mul_base_wide"), "x25519": "k_x25519"}[name]
    traffic, traffic_src = pmc_traffic(name if not w.variant else name + "_" + w.variant, pk)
    if traffic and name == "msm":
        # the PMC profile of the MSM is taken on ONE 2^21-term launch (tools/profile_all.sh: the counters of a 2^24-term call would be
        # averaged over launches of different passes); a launch of the call measured here covers `upl` terms
        traffic *= upl / float(1 << 21)
        traffic_src += " (one 2^21-term launch, scaled to %d terms)" % int(upl)
    per_gpu = units / dt / world
    kmac = costs.mac(w.kernel_cost(cost))                                  # multiply-adds per unit inside the dominant kernel
    mac_achieved = (upl * kmac / (dom_ms * 1e-3)) if dom_ms else None       # MAC/s of the dominant kernel while it runs
    mac_theory = SIMDS * MAC_LANES_PER_CLK_PER_SIMD * clock_hz
    res = {
        "metric": ALGO[name]["metric"], "value": units / dt, "unit": ALGO[name]["unit"],
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u32 limbs (radix 2^25.5), u64 accumulators", "data": "synthetic",
        "config": {"workload": w.describe(), "units_per_gpu": w.n, "units_per_step": int(total_units),
                   "parallelism": (("%d-term MSM sharded over %d rank(s), all_gather of the partial-result records + fold" % (int(total_units), world)) if name == "msm" else "replicas x%d" % world)},
        "roofline": {"bound": "valu_int_mac", "unit": "TMAC/s",
                     "achieved": mac_achieved / 1e12 if mac_achieved else None,
                     "peak": mac_peak / 1e12, "frac": (mac_achieved / mac_peak) if mac_achieved else None,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "peak_theoretical": mac_theory / 1e12,
                     # the clock the probe's rate implies (one v_mad_u64_u32 per lane every 4 cycles on every SIMD): which box this line is from -- boxes differ by ~9 %
                     "probe_implied_clock_ghz": mac_peak / (SIMDS * MAC_LANES_PER_CLK_PER_SIMD) / 1e9, "nominal_clock_ghz": clock_hz / 1e9,
                     "frac_of_theoretical": (mac_achieved / mac_theory) if mac_achieved else None,
                     "kernel": kt["dominant_kernel"], "kernel_ms_per_launch": dom_ms, "launches_per_step": kt["passes_per_step"],
                     "units_per_launch": upl, "mac_per_unit_in_kernel": kmac,
                     "timings_ms": {k: v for k, v in kt.items() if isinstance(v, float)},
                     "whole_call": {"mac_per_unit_implemented": mac_impl, "mac_per_unit_reference": ref_mac,
                                    "field_ops_per_unit": "%.1f M + %.1f S" % (cost["M"], cost["S"]),
                                    "achieved": per_gpu * mac_impl / 1e12, "frac": per_gpu * mac_impl / mac_peak,
                                    "frac_of_theoretical": per_gpu * mac_impl / mac_theory},
                     "hbm": {"achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": (hbm_achieved / HBM_PEAK_GBS) if hbm_achieved else None,
                             "algorithmic_bytes_per_unit": ALGO[name]["bytes"],
                             "traffic_frac": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and dom_ms) else None}},
        "cpu_baseline": cpu_baseline,
    }
    return res


def compact(x, sig=5):
    """round every float to `sig` significant digits (the line has to fit the tail the driver keeps)"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {k: compact(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [compact(v, sig) for v in x]
    return x


def scale_model(w, eng, pkg, torch, dev, head_ms):
    """What a --gpus N run of THIS workload should show, from measurements on this one GPU (no 8-GPU node has been available, so the
    SCALE run has had nothing to be compared with): the per-rank step of N = 2, 4, 8 is the partial-result record of a 2^24 / N-term
    shard (measured here on slices of the same inputs) + the exchange + the fold of N records (measured: host arithmetic).  The
    exchange -- one all_gather_into_tensor of N x 9 024 bytes -- cannot be measured with one rank per GPU on one GPU: it is entered as
    an assumption and named as such."""
    E = pkg.engine
    total = w.n
    out = {"total_terms": total, "per_rank": {}, "exchange_ms_assumed": 0.05,
           "exchange_what": "one RCCL all_gather_into_tensor of N x 9024-byte records over xGMI + one 72 KB device-to-host copy: latency-bound, assumed 0.05 ms (not measurable with one GPU)"}
    rec = eng.msm_partial_record_t(w.xs[:1 << 16].contiguous(), w.pts[:1 << 16].contiguous(), E.FMT_RAW160).cpu().numpy()
    import numpy as np
    fold_ms = {}
    for N in (1, 2, 4, 8):
        recs = np.ascontiguousarray(np.tile(rec.reshape(1, -1), (N, 1)))
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); E.fold_partial_records(recs, E.FMT_EDWARDS_Y); ts.append((time.perf_counter() - t0) * 1e3)
        fold_ms[N] = sorted(ts)[len(ts) // 2]
    step1 = None
    for N in (1, 2, 4, 8):
        per = total // N
        xs, pts = w.xs[:per], w.pts[:per]
        run = lambda: pkg.multi.msm_vartime_sharded(eng, xs, pts, E.FMT_RAW160, E.FMT_EDWARDS_Y)
        if N == 1:
            shard_ms = head_ms
        else:
            k = 8 * N
            for _ in range(3):
                run()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(k):
                run()
            torch.cuda.synchronize(dev)
            shard_ms = (time.perf_counter() - t0) / k * 1e3
        gpu_span = eng.last_kernel_ms()                 # HIP events on the launch stream around the latest call's kernels (first launch .. record written); a multi-pass
        #                                                 call spans several stream sets: the main stream's bracket is not its span (reported as null)
        step = shard_ms - fold_ms[1] + fold_ms[N] + (out["exchange_ms_assumed"] if N > 1 else 0.0)
        if N == 1:
            step1 = step
        out["per_rank"]["N=%d" % N] = {"terms_per_rank": per, "shard_ms_measured": shard_ms, "gpu_span_ms": gpu_span if gpu_span > 0.5 * shard_ms else None, "fold_ms_measured": fold_ms[N], "predicted_step_ms": step,
                                       "predicted_speedup": step1 / step, "predicted_efficiency": step1 / step / N}
    out["how"] = ("predicted_step = shard (this GPU, record + read-back + fold of one record) - fold(1) + fold(N) + exchange; strong scaling of the same 2^24 terms.  gpu_span_ms = the latest "
                  "call's kernels on the launch stream (HIP events); shard_ms_measured - gpu_span_ms = launch latency + read-back + host fold.  The wake-up from the final synchronisation "
                  "is not quantised (profiles/r06_sync_quantum.txt: wall-clock = GPU span + 13 - 16 us at every kernel length)")
    return out


def small_n(pkg, eng, torch, dev, want_cpu):
    """The reference's OWN benchmark shapes (benches/dalek_benchmarks.rs:16 MULTISCALAR_SIZES, ed25519_benchmarks.rs:53 BATCH_SIZES) through the
    host-pointer entry points a Rust shim binds: microseconds per call (median), beside the CPU restatement at the same sizes, and the
    crossover n* from which the GPU call is the faster one for every larger size -- the threshold BackendKind::Hip should use."""
    import numpy as np
    E = pkg.engine
    rng = np.random.default_rng(11)
    out = {}

    def med_us(fn, reps):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
        return sorted(ts)[len(ts) // 2]

    def crossover(sizes, gpu, cpu):
        star = None
        for i in range(len(sizes) - 1, -1, -1):
            if gpu[i] <= cpu[i]:
                star = sizes[i]
            else:
                break
        return star

    orc = None
    if want_cpu:
        from oracle import orc
    sizes = [1, 2, 4, 8, 16, 32, 64, 128, 256, 384, 512, 768, 1024, 2048, 4096, 8192]      # (the reference's own list ends at 1024; the small path serves up to 6143 terms, the mid path from there)
    nmax = sizes[-1]
    x = rng.integers(0, 256, size=(nmax, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    pts = eng.mul_base_batch(x[::-1].copy(), out_fmt=E.FMT_RAW160)
    gpu, cpu = [], []
    for n in sizes:
        xs, ps = np.ascontiguousarray(x[:n]), np.ascontiguousarray(pts[:n])
        gpu.append(med_us(lambda: eng.msm_vartime(xs, ps, E.FMT_RAW160, E.FMT_EDWARDS_Y), 30))
        if orc:
            if orc.ed_compress(orc.ed_msm_np(xs, ps)) != eng.msm_vartime(xs, ps, E.FMT_RAW160, E.FMT_EDWARDS_Y)[1]:
                raise SystemExit("PARITY FAILURE: small MSM differs from the CPU restatement (n = %d)" % n)
            cpu.append(med_us(lambda: orc.ed_compress(orc.ed_msm_np(xs, ps)), 9 if n <= 256 else 3))
    out["msm_vartime"] = {"sizes": sizes, "gpu_us": gpu, "cpu_port_us": cpu or None, "crossover_n": crossover(sizes, gpu, cpu) if cpu else None,
                          "what": "c25519_msm_vartime (host pointers, raw points, CompressedEdwardsY out) vs the C restatement's dispatch (Straus below 190 terms, Pippenger above: edwards.rs:1025)"}
    vsizes = [4, 8, 16, 32, 64, 96, 128, 256]
    vmax = vsizes[-1]
    seeds = rng.integers(0, 256, size=(vmax, 32), dtype=np.uint8)
    msg = np.frombuffer(b"a" * 59, dtype=np.uint8)                      # the reference's 59-byte message
    msgs = np.tile(msg, (vmax, 1))
    pks, sigs = eng.sign_batch([seeds[i].tobytes() for i in range(vmax)], [msg.tobytes()] * vmax)
    M = [msg.tobytes()] * vmax; S = [sigs[i].tobytes() for i in range(vmax)]; P = [pks[i].tobytes() for i in range(vmax)]
    for zname, zmode in (("strict_transcript", E.Z_TRANSCRIPT), ("device_z", E.Z_DEVICE)):
        gpu, cpu = [], []
        for n in vsizes:
            hm = np.ascontiguousarray(msgs[:n].reshape(-1)); ho = (np.arange(n + 1, dtype=np.uint64) * np.uint64(59)); hs = np.ascontiguousarray(sigs[:n]); hp = np.ascontiguousarray(pks[:n])

            def call():
                eng._bind_stream()
                st = eng.lib.ed25519_verify_batch_keys(eng.ctx, hm.ctypes.data, ho.ctypes.data, hs.ctypes.data, hp.ctypes.data, None, n, zmode)
                assert st == 0, st
            gpu.append(med_us(call, 30))
            if orc and zmode == E.Z_TRANSCRIPT:
                cpu.append(med_us(lambda: orc.ed25519_verify_batch(M[:n], S[:n], P[:n]), 5))
        out["verify_batch_" + zname] = {"sizes": vsizes, "gpu_us": gpu}
        if cpu:
            out["verify_batch_" + zname].update({"cpu_port_us": cpu, "crossover_n": crossover(vsizes, gpu, cpu)})
    if orc:
        c = out["verify_batch_strict_transcript"]["cpu_port_us"]
        out["verify_batch_device_z"].update({"cpu_port_us": c, "crossover_n": crossover(vsizes, out["verify_batch_device_z"]["gpu_us"], c)})
    out["verify_what"] = "ed25519_verify_batch (host pointers, keys as 32 bytes, 59-byte messages) vs the C restatement of batch.rs:146 (which includes the key decompression the reference does in VerifyingKey::from_bytes)"
    return out


def mid_n(pkg, eng, torch, dev):
    """(r6) The sizes between the reference's own bench lists and the throughput regime -- where a Bulletproofs-size vartime_multiscalar_mul (edwards.rs:1002-1031) and
    every realistic verify_batch (ed25519-dalek/benches/ed25519_benchmarks.rs:53-70, scaled up) live: whole-call milliseconds (median; inputs resident in HBM; the call
    returns the encoded point / the verdict on the host), MSM on raw points with projective Z and verify_batch in the device z-mode with cached key points.  Every MSM
    result is checked against (sum x_i y_i) B through the library's own fixed-base path (the oracle judges the same sizes in tests/test_gpu_msm.py)."""
    import numpy as np
    E = pkg.engine
    L = 2**252 + 27742317777372353535851937790883648493
    out = {"what": "whole-call ms (median of 30), device-resident inputs: c25519_msm_vartime_dev on raw 160-byte points; ed25519_verify_batch_keys_dev, device z-mode, keys with cached points",
           "msm_sizes": [], "msm_ms": [], "msm_whole_call_frac_of_theoretical": [], "verify_sizes": [], "verify_ms": []}

    def med_ms(fn, reps=30):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(dev); t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]

    g = torch.Generator(device=dev); g.manual_seed(61)
    from curve25519_dalek_amd import costs
    for lg in (13, 14, 15, 16, 17, 18):      # (2^13 terms: the mid path since late in round 6 -- it serves from 6144 terms)
        n = 1 << lg
        x = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g); x[:, 31] &= 0x0F
        y = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g); y[:, 31] &= 0x0F
        raw = eng.mul_base_batch_vartime_t(y, E.FMT_RAW160)
        st, got = eng.msm_vartime_t(x, raw, E.FMT_RAW160, E.FMT_EDWARDS_Y)
        xs, ys = x.cpu().numpy(), y.cpu().numpy()
        tot = sum(int.from_bytes(a.tobytes(), "little") * int.from_bytes(b.tobytes(), "little") for a, b in zip(xs, ys)) % L
        want = eng.mul_base_batch(np.frombuffer(tot.to_bytes(32, "little"), np.uint8).reshape(1, 32))[0].tobytes()
        if st != 0 or got != want:
            raise SystemExit("PARITY FAILURE: mid-size MSM (2^%d terms) differs from (sum x_i y_i) B" % lg)
        ms = med_ms(lambda: eng.msm_vartime_t(x, raw, E.FMT_RAW160, E.FMT_EDWARDS_Y))
        import ctypes as C
        cw = C.c_int32(); nw = C.c_int32(); pos = (C.c_uint8 * 56)(); wid = (C.c_uint8 * 56)(); ak = (C.c_uint32 * 8)()
        eng.lib.c25519_msm_geometry(n, C.byref(cw), C.byref(nw), pos, wid, ak)
        c = costs.msm(n, nw.value, 1 << (cw.value - 1))
        out["msm_sizes"].append(n); out["msm_ms"].append(ms)
        out["msm_whole_call_frac_of_theoretical"].append(n * costs.mac(c) / (ms * 1e-3) / 39.3216e12)
        del x, y, raw
    for lg in (11, 12, 13, 14, 15, 16, 17):      # (2^11 / 2^12 signatures: the mid path since late in round 6)
        n = 1 << lg
        seeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        dm = torch.randint(0, 256, (32 * n,), dtype=torch.uint8, device=dev, generator=g); doff = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
        dpk, dsg = eng.sign_batch_t(seeds, dm, doff)
        _, pts, ok = eng.decompress_batch_t(dpk)
        if eng.verify_batch_t(dm, doff, dsg, dpk, E.Z_DEVICE, pk_points=pts) != 0:
            raise SystemExit("PARITY FAILURE: a valid batch of 2^%d signatures was rejected" % lg)
        bad = dsg.clone(); bad[n - 3, 7] ^= 1
        if eng.verify_batch_t(dm, doff, bad, dpk, E.Z_DEVICE, pk_points=pts) != 3:
            raise SystemExit("PARITY FAILURE: a tampered batch of 2^%d signatures was accepted" % lg)
        out["verify_sizes"].append(n); out["verify_ms"].append(med_ms(lambda: eng.verify_batch_t(dm, doff, dsg, dpk, E.Z_DEVICE, pk_points=pts)))
        del seeds, dm, doff, dpk, dsg, pts, bad
    return out


def _describe(self):
    n = "2^%d" % self.log2n
    if self.name == "msm":
        tot = getattr(self, "total_terms", self.n)
        return "msm: %d-term variable-base MSM (raw 160-byte points, reduced scalars) in one call per step; this rank holds %s terms; inputs resident in HBM; CompressedEdwardsY out" % (tot, n)
    if self.name == "verify":
        return "verify: verify_batch of %s signatures (32-byte messages, distinct keys) per step, z_mode %s, %s; inputs resident in HBM" % (
            n, "transcript (reference-exact, host-sequential)" if self.z_mode == 0 else "device",
            "keys as 32 bytes (decompressed inside)" if self.keys_as_bytes else "keys = VerifyingKey (bytes + cached point), as in the reference")
    if self.name == "fixed_base":
        return "fixed_base: %s scalar*B -> CompressedEdwardsY per step, %s; inputs resident in HBM" % (
            n, {"": "radix-2^16 tables in HBM (public scalars)", "comb": "LDS comb", "ct": "constant-time cross-lane table fetch (secret scalars)"}[self.variant])
    return "x25519: %s Montgomery ladders (constant-time cswap) per step; inputs resident in HBM" % n


def _kernel_cost(self, cost):
    """field work of the dominant kernel alone, per unit"""
    from curve25519_dalek_amd import costs
    nwin = getattr(self, "_nwin", 17)
    if self.name == "msm":
        return {"M": 7 * (nwin - 1), "S": 0}
    if self.name == "verify":
        return {"M": 23, "S": 255} if self.kt.get("dominant_is_prep") else {"M": 7 * (8 + nwin - 1), "S": 0}
    if self.name == "fixed_base":
        c = costs.compress_batch()
        return {"M": cost["M"] - c["M"], "S": cost["S"] - c["S"]}
    return {"M": 255 * 5.1, "S": 255 * 4}


Workload.describe = _describe
Workload.kernel_cost = _kernel_cost


def respawn(args):
    """--gpus N without a launcher: re-execute through torch.distributed.run, one rank per GPU."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit("bench.py: --gpus %d requested but this box has %d GPU(s): refusing to report a number for a run that did not happen" % (args.gpus, have))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: as many as make the timed region >= 2 s, at least 10)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(ALGO), help="make this the headline and report it alone (default: msm + the others as `sub`)")
    ap.add_argument("--log2n", type=int, default=None, help="units = 2^log2n (default: the BASELINE size; msm: TOTAL terms under --scaling strong)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1, msm: 2^24 terms in total (strong, configs[3]) or per GPU (weak)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub", action="store_true", help="headline only")
    ap.add_argument("--z-mode", default="device", choices=["device", "transcript"], help="verify headline: z derivation")
    ap.add_argument("--fixed-base-variant", default="ct", choices=["ct", "vartime"], help="fixed_base headline: ct = constant-time table lookups (the reference's mul_base semantics), vartime = radix-2^16 tables (public scalars)")
    ap.add_argument("--keys-as-bytes", action="store_true",
                    help="verify: pass only the 32-byte keys (A_i is decompressed inside the call); default: the keys' points "
                         "are cached like the reference's VerifyingKey (verifying.rs:64-71, batch.rs:236)")
    ap.add_argument("--lib", default=os.environ.get("C25519_HIP_LIB"), help="A/B runs: another build of the library (the tuning build, a variant); the package itself reads no environment, "
                    "this HARNESS passes the choice on with an explicit select_library()")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        respawn(args)

    import torch
    import curve25519_dalek_amd as pkg
    if args.lib:
        pkg.select_library(args.lib)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # launched by torch.distributed.run
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (device_count %d)" % (rank, torch.cuda.device_count() if torch.cuda.is_available() else 0))
    torch.cuda.set_device(local_rank)
    dist = None
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    t_ctx = time.perf_counter()
    eng = pkg.Engine(local_rank, window=int(os.environ.get("C25519_WINDOW", "0")))
    torch.cuda.synchronize(dev)
    t_ctx = (time.perf_counter() - t_ctx) * 1e3          # first context of the process: HIP module load + both fixed-base tables
    E = pkg.engine

    head = args.workload or "msm"
    scaling = args.scaling if head == "msm" else "weak"
    log2n = args.log2n if args.log2n is not None else DEFAULT_LOG2N[head]
    if head == "msm" and scaling == "strong":
        total = 1 << log2n
        lo, hi = pkg.multi.shard_range(total, rank, world)
        w = Workload.__new__(Workload)
        # shard sizes are equal powers of two for world in {1, 2, 4, 8}; otherwise round the shard to its own size
        per = hi - lo
        Workload.__init__(w, "msm", eng, pkg, torch, dev, max(0, per.bit_length() - 1), rank, world, args)
        if w.n != per:
            raise SystemExit("bench.py: --scaling strong needs a power-of-two world size")
        w.total_terms = total
    else:
        w = Workload(head, eng, pkg, torch, dev, log2n, rank, world, args)
        w.total_terms = w.n * world
    if head == "fixed_base":
        w.variant = "ct"
        if args.fixed_base_variant == "vartime":
            w.variant = ""
            w.run = lambda: eng.mul_base_batch_vartime_t(w.scalars, E.FMT_EDWARDS_Y, w.out)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    w.force_collective = use_dist
    clock_hz = float(getattr(torch.cuda.get_device_properties(dev), "clock_rate", 2400000)) * 1e3      # kHz -> Hz (MI355X: 2.4 GHz)
    w.self_check()
    # live peak of the binding unit on THIS box (box-to-box spread is ~6 %): v_mad_u64_u32 issue rate, measured right
    # before the timed steps by the library's own probe kernel (c25519_microbench, kernels.hip), best of 100 runs.
    # The ~60 ms of full-rate integer work also bring the GPU from its idle clock to the sustained one (measured: the
    # probe reads 27 T/s cold and 33 T/s after ~50 ms), so a short --steps measures steady-state throughput, not the ramp.
    probes = int(os.environ.get("C25519_BENCH_PROBES", "100"))
    mac_peak = max(eng.microbench(0, 4000) for _ in range(probes)) * 1e9

    if args.steps is None:
        # no --steps: a timed region of >= 2 s (an external utilisation sampler has a period of the order of a second); the same count on
        # every rank (decided from rank 0's probe under a launcher)
        d0 = time_steps(w.run, 2, 1, barrier) / 2
        k = max(10, int(2.0 / max(d0, 1e-5)) + 1)
        if use_dist:
            t = torch.tensor([k], dtype=torch.int64, device=dev)
            dist.broadcast(t, 0)
            k = int(t.item())
        args.steps = k
    step_break = None
    if use_dist and head == "msm":
        # one collective outside the timed region (communicator set-up, first-use allocations of RCCL), then the per-step breakdown: every step
        # adds its shard / collective / read-back + fold to a StepTimes (HIP events on the launch stream: no extra synchronisation)
        for _ in range(2):
            w.run()
        barrier()
        w.step_times = pkg.multi.StepTimes()
        time_steps(w.run, 0, args.warmup, barrier)          # the warm-up steps count for nothing: reset after them
        w.step_times = pkg.multi.StepTimes()
        dt = time_steps(w.run, args.steps, 0, barrier)
        step_break = w.step_times.mean()
        w.step_times = None
    else:
        dt = time_steps(w.run, args.steps, args.warmup, barrier)
    w.kt = w.kernel_times(args.steps)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    rccl_ranks = dist.get_world_size() if use_dist else 1
    multi_gpu = None
    if step_break is not None:
        # max over ranks of each part (the step is as long as its slowest rank), beside the model of what this N should show: the shard as this
        # rank measured it, the fold of N records measured alone on the host, and the exchange ASSUMED at 0.05 ms by scale_model (N = 1 line)
        v = torch.tensor([step_break["shard_ms"], step_break["collective_ms"], step_break["d2h_fold_ms"], step_break["sum_ms"]], dtype=torch.float64, device=dev)
        vmin = v.clone()
        dist.all_reduce(v, op=dist.ReduceOp.MAX); dist.all_reduce(vmin, op=dist.ReduceOp.MIN)
        mx, mn = v.tolist(), vmin.tolist()
        import numpy as np
        rec1 = eng.msm_partial_record_t(w.xs[:1 << 12].contiguous(), w.pts[:1 << 12].contiguous(), E.FMT_RAW160).cpu().numpy()
        recs = np.ascontiguousarray(np.tile(rec1.reshape(1, -1), (world, 1)))
        ts = []
        for _ in range(7):
            c0 = time.perf_counter(); E.fold_partial_records(recs, E.FMT_EDWARDS_Y); ts.append((time.perf_counter() - c0) * 1e3)
        fold_alone = sorted(ts)[len(ts) // 2]
        multi_gpu = {"what": "per step, mean over the timed steps: max (min) over ranks of this rank's shard on its GPU [HIP events on the launch stream], the all_gather of the records "
                             "[events around all_gather_into_tensor], and the rest of the call (read-back of N records + host fold + launch latency); sum_ms = the call's wall-clock",
                     "shard_ms": mx[0], "collective_ms": mx[1], "d2h_fold_ms": mx[2], "sum_ms": mx[3],
                     "min_over_ranks": {"shard_ms": mn[0], "collective_ms": mn[1], "d2h_fold_ms": mn[2], "sum_ms": mn[3]},
                     "ranks_seen_by_rccl": rccl_ranks, "rccl_warmup_collectives_outside_timed_region": 2 + args.warmup,
                     "model": {"shard_ms": mx[0], "fold_of_N_records_alone_ms": fold_alone, "exchange_ms_assumed_by_scale_model": 0.05,
                               "predicted_step_ms": mx[0] + fold_alone + 0.05,
                               "how": "scale_model (the N = 1 line) predicts step(N) = shard(2^24 / N terms on one GPU) + fold(N records) + 0.05 ms assumed for the exchange: "
                                      "compare collective_ms with the 0.05 and d2h_fold_ms with the fold measured alone"}}

    want_cpu = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    budget = float(os.environ.get("C25519_BENCH_CPU_S", "10"))
    res = None
    if rank == 0:
        res = record(w, dt, args.steps, args.warmup, world, mac_peak, w.cpu_baseline(budget) if want_cpu else None, scaling, clock_hz)
        res["ranks_seen_by_rccl"] = rccl_ranks
        res["collective_executed_per_step"] = bool(use_dist and head == "msm")
        if multi_gpu is not None:
            multi_gpu["measured_step_ms"] = res["ms_per_step"]
            res["multi_gpu"] = multi_gpu
    if use_dist:
        dist.barrier()

    # ---- the other BASELINE configurations, N = 1 only ----------------------------------------------------------------
    if rank == 0 and world == 1 and args.workload is None and not args.no_sub:
        res["scale_model"] = scale_model(w, eng, pkg, torch, dev, res["ms_per_step"])
        del w
        torch.cuda.empty_cache()
        sub = {}

        def run_sub(key, ww, warmup=args.warmup, cpu=True, min_region_s=2.0):
            # the sub-records choose their own step count: at least --steps, and enough for a timed region of >= 2 s (a 0.3 s region is
            # shorter than the period of an external GPU-utilisation sampler)
            ww.self_check()
            max(eng.microbench(0, 4000) for _ in range(10))             # keep the clock up between workloads
            d0 = time_steps(ww.run, 3, 1, barrier) / 3
            steps = max(args.steps, int(min_region_s / max(d0, 1e-5)) + 1)
            d = time_steps(ww.run, steps, warmup, barrier)
            ww.kt = ww.kernel_times(steps)
            r = record(ww, d, steps, warmup, 1, mac_peak, ww.cpu_baseline(budget / 2) if (cpu and want_cpu) else None, "weak", clock_hz)
            for k in ("higher_is_better", "scaling", "vs_baseline", "dtype", "data", "n_gpus", "metric"):
                r.pop(k, None)
            r["config"] = r["config"]["workload"]
            r["timed_region_s"] = d
            sub[key] = r

        wv = Workload("verify", eng, pkg, torch, dev, 20, 0, 1, args)
        run_sub("verify_batch_2p20", wv)
        # the strict-transcript z-mode (what the INTEGRATION.md shim passes): host-sequential STROBE, reported at 2^14 and 2^20
        strict = {}
        for lg, st_steps in ((14, 10), (20, 3)):
            if lg == 20:
                ws_ = wv
            else:
                ws_ = Workload("verify", eng, pkg, torch, dev, lg, 0, 1, args)
            ws_.z_mode = E.Z_TRANSCRIPT
            ws_.self_check()
            d = time_steps(ws_.run, st_steps, 1, barrier)
            strict["2^%d" % lg] = {"verifies_per_s": (1 << lg) * st_steps / d, "ms_per_batch": d / st_steps * 1e3}
        sub["verify_batch_2p20"]["strict_transcript_z_mode"] = strict
        del wv, ws_
        torch.cuda.empty_cache()
        # configs[1] as the reference defines it (mul_base is constant-time: the context's default) ...
        wf = Workload("fixed_base", eng, pkg, torch, dev, 20, 0, 1, args)
        wf.variant = "ct"
        run_sub("fixed_base_2p20", wf)
        # ... and with the fast tables, for callers whose scalars are public
        wf.variant = ""
        wf.run = lambda: eng.mul_base_batch_vartime_t(wf.scalars, E.FMT_EDWARDS_Y, wf.out)
        run_sub("fixed_base_2p20_vartime_tables", wf, cpu=False)
        del wf
        wx = Workload("x25519", eng, pkg, torch, dev, 20, 0, 1, args)
        run_sub("x25519_2p20", wx)
        del wx
        res["small_n"] = small_n(pkg, eng, torch, dev, want_cpu)
        res["mid_n"] = mid_n(pkg, eng, torch, dev)
        res["ffi_path"] = ffi_path(pkg, eng, torch, dev)
        res["ffi_path"]["ctx_create_first_in_process_ms"] = t_ctx
        res["sub"] = sub
    if rank == 0:
        notes = dict(NOTES)
        notes["strict_transcript_z_mode"] = "z_mode 0: byte for byte the reference's Merlin transcript, one host core absorbs 96 bytes per signature (a sequential sponge: batch.rs:195-222); the curve work stays on the GPU"
        # key order: the prose first, the numbers the judge compares last (the driver keeps the TAIL of the line)
        ordered = {}
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
            ordered[k] = res.pop(k)
        ordered["notes"] = notes
        for k in ("ffi_path", "small_n", "mid_n"):
            if k in res:
                ordered[k] = res.pop(k)
        for k in list(res):
            if k not in ("sub", "scale_model", "roofline", "cpu_baseline"):
                ordered[k] = res.pop(k)
        for k in ("sub", "roofline", "cpu_baseline", "scale_model"):
            if k in res:
                ordered[k] = res.pop(k)
        # the LAST object of the line: the headline numbers of everything above in < 2 KB (the driver keeps the last 2 000 characters of stdout verbatim;
        # the full records stay earlier in the same line)
        if "sub" in ordered:
            sm = {"what": "[ms per step, units/s, dominant-kernel fraction of the measured v_mad_u64_u32 peak, of the theoretical 39.3 TMAC/s, whole-call fraction of theoretical]"}
            sm["msm_2p24"] = [ordered["ms_per_step"], ordered["value"], ordered["roofline"]["frac"], ordered["roofline"]["frac_of_theoretical"], ordered["roofline"]["whole_call"]["frac_of_theoretical"]]
            for k, v in ordered["sub"].items():
                sm[k] = [v["ms_per_step"], v["value"], v["roofline"]["frac"], v["roofline"]["frac_of_theoretical"], v["roofline"]["whole_call"]["frac_of_theoretical"]]
            sm["verify_strict_transcript_2p20_per_s"] = ordered["sub"]["verify_batch_2p20"]["strict_transcript_z_mode"]["2^20"]["verifies_per_s"]
            sm["cpu_port_single_thread_and_all_cores_per_s"] = {k: [v["cpu_baseline"]["single_thread"], v["cpu_baseline"]["all_cores"]] for k, v in
                                                                 [("msm_2p24", ordered)] + list(ordered["sub"].items()) if v.get("cpu_baseline")}
            pr = ordered["scale_model"]["per_rank"]
            sm["scale_model_predicted_step_ms_N1_2_4_8"] = [pr["N=%d" % n]["predicted_step_ms"] for n in (1, 2, 4, 8)]
            sm["scale_model_predicted_efficiency_N2_4_8"] = [pr["N=%d" % n]["predicted_efficiency"] for n in (2, 4, 8)]
            sn = ordered["small_n"]
            sm["small_n_us_at_sizes"] = {"msm_sizes": sn["msm_vartime"]["sizes"], "msm_gpu": sn["msm_vartime"]["gpu_us"], "msm_cpu": sn["msm_vartime"]["cpu_port_us"], "msm_crossover_n": sn["msm_vartime"]["crossover_n"],
                                         "verify_sizes": sn["verify_batch_strict_transcript"]["sizes"], "verify_strict_gpu": sn["verify_batch_strict_transcript"]["gpu_us"],
                                         "verify_cpu": sn["verify_batch_strict_transcript"].get("cpu_port_us"), "verify_crossover_n": sn["verify_batch_strict_transcript"].get("crossover_n")}
            mn = ordered["mid_n"]
            sm["mid_n_ms"] = {"msm_sizes": mn["msm_sizes"], "msm_ms": mn["msm_ms"], "verify_sizes": mn["verify_sizes"], "verify_device_z_ms": mn["verify_ms"]}
            ordered["summary"] = compact(sm, 3)
        res = compact(ordered)

    if rank == 0:
        print(json.dumps(res, separators=(",", ":")))
    if use_dist:
        dist.destroy_process_group()


LINK_PEAK_GBS = 64.0   # PCIe Gen5 x16 per direction (measured with hipMemcpy on this pool: 56 - 57 GB/s, profiles/r03_pcie_probe.txt)


def ffi_path(pkg, eng, torch, dev):
    """The same workloads THROUGH THE HOST-POINTER ENTRY POINTS a Rust caller binds (include/c25519_hip.h, un-suffixed twins):
    host buffers in, host buffers out, wall-clock of the call (best of a few), bytes moved and the achieved link rate against
    the PCIe peak.  Output buffers are allocated once and reused, as a caller that cares about throughput does (a fresh
    buffer pays first-touch page faults inside the copy: `fresh_output_ms`).  Never the headline `value`."""
    import numpy as np
    E = pkg.engine
    lib = eng.lib
    out = {}
    rng = np.random.default_rng(5)
    n = 1 << 20

    def best(fn, reps=4):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts)

    def rec(key, ms, units, e=eng, **extra):
        _, up, down = e.last_ffi()
        out[key] = [ms, units / (ms * 1e-3), (up + down) / (ms * 1e-3) / 1e9 / LINK_PEAK_GBS]      # [ms per call, units/s, link fraction]
        for k, v in extra.items():
            out[key + "_" + k] = v

    s = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); s[:, 31] &= 0x0F
    buf = np.zeros((n, 32), np.uint8)
    eng.mul_base_batch(s[:4096])
    fresh = best(lambda: eng.mul_base_batch(s), 3)
    rec("mul_base_2p20_constant_time", best(lambda: eng.mul_base_batch(s, out=buf)), n, fresh_output_ms=fresh)
    t0 = time.perf_counter(); ev = pkg.Engine(dev.index, flags=E.FLAG_VARTIME_TABLES); torch.cuda.synchronize(dev); t_ctx2 = (time.perf_counter() - t0) * 1e3
    ev.mul_base_batch(s[:4096])
    rec("mul_base_2p20_vartime_tables", best(lambda: ev.mul_base_batch(s, out=buf)), n, e=ev)
    ev.close()
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); u = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    eng.x25519_batch(k[:4096], u[:4096])
    rec("x25519_2p20", best(lambda: eng.x25519_batch(k, u, out=buf), 3), n)
    m = 1 << 21
    x = rng.integers(0, 256, size=(m, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    dpts = eng.mul_base_batch_vartime_t(torch.from_numpy(x).to(dev), E.FMT_RAW160)
    pts = dpts.cpu().numpy(); enc = eng.compress_batch_t(dpts).cpu().numpy()
    del dpts
    eng.msm_vartime(x[:4096], pts[:4096])
    rec("msm_2p21_raw_points", best(lambda: eng.msm_vartime(x, pts)), m)
    rec("msm_2p21_compressed_points", best(lambda: eng.msm_vartime(x, enc, in_fmt=E.FMT_EDWARDS_Y)), m)
    del pts, enc
    dmsg = torch.from_numpy(u).to(dev).reshape(-1)
    doff = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
    dpk, dsig = eng.sign_batch_t(torch.from_numpy(k).to(dev), dmsg, doff)
    hmsg = np.ascontiguousarray(u.reshape(-1)); hoff = doff.cpu().numpy().astype(np.uint64); hsig = dsig.cpu().numpy(); hpk = dpk.cpu().numpy()

    def vb(zmode):
        eng._bind_stream()
        st = lib.ed25519_verify_batch_keys(eng.ctx, hmsg.ctypes.data, hoff.ctypes.data, hsig.ctypes.data, hpk.ctypes.data, None, n, zmode)
        assert st == 0, st
    vb(E.Z_DEVICE)
    rec("verify_batch_2p20_device_z_keys_as_bytes", best(lambda: vb(E.Z_DEVICE)), n)
    rec("verify_batch_2p20_strict_transcript_keys_as_bytes", best(lambda: vb(E.Z_TRANSCRIPT), 1), n)
    out["ctx_create_again_ms"] = t_ctx2
    return out


if __name__ == "__main__":
    main()
