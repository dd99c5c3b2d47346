import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)
E = pkg.engine
rng = np.random.default_rng(3)
n = 1 << 18
h = rng.integers(0, 256, size=(n, 64), dtype=np.uint8); s = rng.integers(0, 256, size=(n, 64), dtype=np.uint8)
E.batch_transcript_zs(h[:1000], s[:1000])
t0 = time.perf_counter(); z = E.batch_transcript_zs(h, s); dt = time.perf_counter() - t0
print("transcript zs: %.2f M signatures/s (%.1f ns each), digest %s" % (n / dt / 1e6, dt / n * 1e9, __import__('hashlib').sha256(z.tobytes()).hexdigest()[:16]))
