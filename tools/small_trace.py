"""Traced under rocprofv3 --kernel-trace (docs/lab/r05_call18.sh): six calls each of the small host-pointer entry points, so that the kernel
timeline of the LAST call of every group can be read off (tools/timeline_tail.py)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)
e = pkg.Engine(0); E = pkg.engine
rng = np.random.default_rng(1)
what = sys.argv[1] if len(sys.argv) > 1 else "msm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if what == "msm":
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    pts = e.mul_base_batch_vartime_t(torch.from_numpy(x).cuda(), E.FMT_RAW160).cpu().numpy()
    torch.cuda.synchronize()
    for _ in range(6): e.msm_vartime(x, pts)
else:
    seeds = [bytes(rng.integers(0, 256, 32, dtype=np.uint8)) for _ in range(n)]
    msgs = [bytes(rng.integers(0, 256, 59, dtype=np.uint8)) for _ in range(n)]
    pks, sigs = e.sign_batch(seeds, msgs)
    pks = [bytes(p) for p in pks]; sigs = [bytes(s) for s in sigs]
    torch.cuda.synchronize()
    for _ in range(6): assert e.verify_batch(msgs, sigs, pks, 0) == 0
