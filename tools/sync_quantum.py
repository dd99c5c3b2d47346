#!/usr/bin/env python3
"""Is the host's wake-up from a stream synchronisation quantised?  A spin kernel of controlled length, then torch.cuda.synchronize(), wall-clock per call.
(Round 5: bench lines of ~1.9 ms GPU spans clustered at 2.000 ms per step.)"""
import time, torch
torch.cuda.init()
x = torch.zeros(1, device="cuda")
for _ in range(20):
    torch.cuda._sleep(100000); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("%12s %12s %12s" % ("cycles", "GPU span us", "wall us"))
for cyc in [int(2.0e6 * f) for f in (0.2, 0.5, 0.8, 0.85, 0.9, 0.92, 0.94, 0.96, 0.98, 1.0, 1.02, 1.05, 1.1, 1.3, 1.6, 1.9, 1.95, 2.0, 2.05)]:
    w, g = [], []
    for _ in range(30):
        t0 = time.perf_counter(); e0.record(); torch.cuda._sleep(cyc); e1.record(); torch.cuda.synchronize(); w.append((time.perf_counter() - t0) * 1e6); g.append(e0.elapsed_time(e1) * 1e3)
    w.sort(); g.sort()
    print("%12d %12.1f %12.1f" % (cyc, g[15], w[15]))
