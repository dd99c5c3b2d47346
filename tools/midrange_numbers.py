#!/usr/bin/env python3
"""c25519_msm_vartime (host pointers, raw points) across the path boundaries: small path (<= 4095 terms), digit-matrix sort (4096 .. 65535; the
chunk-local sort there with C25519_SORT_CHUNK_LOCAL_MIN=4096), the full pipeline above; device-resident calls beside them.   python tools/midrange_numbers.py > profiles/rNN_msm_midrange.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import curve25519_dalek_amd as pkg
import devlib; devlib.apply(pkg)      # (C25519_HIP_LIB of this TOOL's environment selects another build; the package reads no environment)

e = pkg.Engine(0)
E = pkg.engine
rng = np.random.default_rng(1)
for _ in range(40):
    e.microbench(0, 4000)
print("%10s %16s %16s %8s" % ("n", "host-pointer ms", "device-resident ms", "window"))
import ctypes as C
lib = pkg.load_library()
for n in [int(v) for v in os.environ.get('MIDRANGE_SIZES', '256,1024,2047,2048,4096,8192,16384,32768,65535,65536,131072,262144,524288,1048576').split(',')]:
    x = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    dx = torch.from_numpy(x).cuda()
    dp = e.mul_base_batch_vartime_t(dx, E.FMT_RAW160)
    fmt = int(os.environ.get('MIDRANGE_FMT', E.FMT_RAW160))      # 2 = raw 160-byte points (default), 0 = CompressedEdwardsY (the decompression's affine records: prepared)
    if fmt != E.FMT_RAW160:
        dp = torch.from_numpy(e.compress_batch(dp.cpu().numpy(), out_fmt=fmt)).cuda()
    pts = dp.cpu().numpy()
    st, res = e.msm_vartime(x, pts, in_fmt=fmt); e.msm_vartime_t(dx, dp, fmt)
    if os.environ.get('MIDRANGE_DUMP'):      # the encoded result of every size, so that two arms of an A/B can be compared byte for byte
        with open(os.environ['MIDRANGE_DUMP'], 'a') as f: f.write("%d %d %s\n" % (n, st, res.hex()))
    reps = 30 if n <= 1 << 16 else 8
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); e.msm_vartime(x, pts, in_fmt=fmt); ts.append(time.perf_counter() - t0)
    td = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); e.msm_vartime_t(dx, dp, fmt); td.append(time.perf_counter() - t0)
    c = C.c_int32(); nw = C.c_int32(); pos = (C.c_uint8 * 56)(); wid = (C.c_uint8 * 56)(); ak = (C.c_uint32 * 8)()
    lib.c25519_msm_geometry(n, C.byref(c), C.byref(nw), pos, wid, ak)
    print("%10d %16.3f %16.3f %8d" % (n, sorted(ts)[len(ts) // 2] * 1e3, sorted(td)[len(td) // 2] * 1e3, c.value))
