import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import curve25519_dalek_amd as pkg
e = pkg.Engine(0); E = pkg.engine
rng = np.random.default_rng(1)
for n in (1, 256, 1024):
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    pts = e.mul_base_batch_vartime_t(torch.from_numpy(x).cuda(), E.FMT_RAW160).cpu().numpy()
    for _ in range(6): e.msm_vartime(x, pts)
