import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)
E = pkg.engine
eng = pkg.Engine(0)
dev = torch.device('cuda', 0)
for lg in (14, 20):
    n = 1 << lg
    g = torch.Generator(device=dev); g.manual_seed(5)
    seeds = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    msgs = torch.randint(0, 256, (n * 32,), dtype=torch.uint8, device=dev, generator=g)
    off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int64, device=dev)
    pks, sigs = eng.sign_batch_t(seeds, msgs, off)
    _, pts, ok = eng.decompress_batch_t(pks)
    for z in (E.Z_TRANSCRIPT,):
        assert eng.verify_batch_t(msgs, off, sigs, pks, z, pk_points=pts) == 0
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); st = eng.verify_batch_t(msgs, off, sigs, pks, z, pk_points=pts); ts.append(time.perf_counter() - t0)
        print("strict verify_batch 2^%d: %.2f ms = %.2f M/s" % (lg, min(ts) * 1e3, n / min(ts) / 1e6))
