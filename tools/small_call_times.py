import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)
e = pkg.Engine(0); E = pkg.engine
rng = np.random.default_rng(1)
for _ in range(40): e.microbench(0, 4000)
for n in (1, 16, 64, 256, 1024, 2047, 4000):
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    dx = torch.from_numpy(x).cuda(); dp = e.mul_base_batch_vartime_t(dx, E.FMT_RAW160); pts = dp.cpu().numpy()
    enc = e.compress_batch_t(dp).cpu().numpy()
    for _ in range(5): e.msm_vartime(x, pts); e.msm_vartime(x, enc, in_fmt=0)
    t = []
    for _ in range(200):
        t0 = time.perf_counter(); e.msm_vartime(x, pts); t.append(time.perf_counter() - t0)
    t2 = []
    for _ in range(200):
        t0 = time.perf_counter(); e.msm_vartime(x, enc, in_fmt=0); t2.append(time.perf_counter() - t0)
    print("msm n=%5d  raw %.1f us  compressed %.1f us" % (n, sorted(t)[100] * 1e6, sorted(t2)[100] * 1e6))
for n in (4, 64, 256, 1024, 2047):
    seeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda")
    dm = torch.randint(0, 256, (59 * n,), dtype=torch.uint8, device="cuda"); doff = torch.arange(0, 59 * (n + 1), 59, dtype=torch.int64, device="cuda")
    dpk, dsg = e.sign_batch_t(seeds, dm, doff)
    M = dm.cpu().numpy(); P = dpk.cpu().numpy(); S = dsg.cpu().numpy()
    msgs = [M[59 * i:59 * i + 59].tobytes() for i in range(n)]; sigs = [S[i].tobytes() for i in range(n)]; pks = [P[i].tobytes() for i in range(n)]
    for zm in (0, 1):
        for _ in range(5): assert e.verify_batch(msgs, sigs, pks, zm) == 0
        t = []
        for _ in range(100):
            t0 = time.perf_counter(); e.verify_batch(msgs, sigs, pks, zm); t.append(time.perf_counter() - t0)
        print("verify n=%5d z_mode %d  %.1f us (python list marshalling included)" % (n, zm, sorted(t)[50] * 1e6))
