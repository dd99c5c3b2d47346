import os, sys, random, faulthandler, time
faulthandler.enable()
faulthandler.dump_traceback_later(90, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)
from oracle import orc
L = 2**252 + 27742317777372353535851937790883648493
rnd = random.Random(5)
t0 = time.time()
def P(*a):
    print("%.2f" % (time.time() - t0), *a, flush=True)
e0 = pkg.Engine(0)
P("engine")
N = 130
msgs = [bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 1, 37, 47, 48, 63, 64, 111, 112, 175, 176, 300]))) for _ in range(N)]
seeds = [bytes(rnd.randrange(256) for _ in range(32)) for _ in range(N)]
pks, sigs = e0.sign_batch(seeds, msgs)
P("signed")
pks = [bytes(p) for p in pks]; sigs = [bytes(s) for s in sigs]
assert orc.ed25519_verify(pks[7], msgs[7], sigs[7]) == 0
P("orc single ok")
i2b = lambda x: int(x).to_bytes(32, "little")
for n in (1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129):
    eng = pkg.Engine(0)
    P("n", n, "fresh engine")
    m, s, p = msgs[:n], sigs[:n], pks[:n]
    a = eng.verify_batch(m, s, p, 0); P(" eng", a)
    b = orc.ed25519_verify_batch(m, s, p); P(" orc", b)
    assert a == 0 == b, n
    for idx in (0, n - 1):
        bad = list(s); bb = bytearray(bad[idx]); bb[40] ^= 1; bad[idx] = bytes(bb)
        a = eng.verify_batch(m, bad, p, 0); P(" bad eng", a); b = orc.ed25519_verify_batch(m, bad, p); P(" bad orc", b)
        assert a == b != 0
        sbig = list(s); sbig[idx] = s[idx][:32] + i2b(int.from_bytes(s[idx][32:], "little") + L)
        a = eng.verify_batch(m, sbig, p, 0); P(" sbig eng", a); b = orc.ed25519_verify_batch(m, sbig, p); P(" sbig orc", b)
        assert a == b == 2
        rbad = list(sbig); rbad[n - 1 - idx] = i2b(2) + rbad[n - 1 - idx][32:]
        a = eng.verify_batch(m, rbad, p, 0); P(" rbad eng", a); b = orc.ed25519_verify_batch(m, rbad, p); P(" rbad orc", b)
        assert a == b == 2
        ronly = list(s); ronly[idx] = i2b(2) + s[idx][32:]
        a = eng.verify_batch(m, ronly, p, 0); b = orc.ed25519_verify_batch(m, ronly, p); P(" ronly", a, b)
        assert a == b == 3
        abad = list(p); abad[idx] = i2b(2)
        a = eng.verify_batch(m, rbad, abad, 0); b = orc.ed25519_verify_batch(m, rbad, abad); P(" abad", a, b)
        assert a == b == 1
        mm = list(m); mm[idx] = m[idx] + b"!"
        a = eng.verify_batch(mm, s, p, 0); P(" mm", a)
        assert a == 3
    assert eng.verify_batch(m, s, p, 0) == 0
    assert eng.verify_batch(m, s, p, 1) == 0
print("ok")
