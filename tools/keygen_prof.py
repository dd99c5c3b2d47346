"""keygen_batch / sign_batch at 2^16 for rocprofv3 --kernel-trace (profiles/rNN_sign_keygen_2p16.txt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)
e = pkg.Engine(0)
g = torch.Generator(device="cuda"); g.manual_seed(7)
m = 1 << 16
seeds = torch.randint(0, 256, (m, 32), dtype=torch.uint8, device="cuda", generator=g)
msgs = torch.randint(0, 256, (m * 59,), dtype=torch.uint8, device="cuda", generator=g)
off = torch.arange(0, 59 * (m + 1), 59, dtype=torch.int64, device="cuda")
for _ in range(40): e.microbench(0, 4000)
for _ in range(10):
    e.keygen_batch_t(seeds)
torch.cuda.synchronize()
for _ in range(10):
    e.sign_batch_t(seeds, msgs, off)
torch.cuda.synchronize()
