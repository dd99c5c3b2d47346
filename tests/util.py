"""Shared helpers for the GPU parity tests: seeded input generators (numpy) and oracle batch calls."""
import os

import numpy as np

L = 2**252 + 27742317777372353535851937790883648493
P = 2**255 - 19

# The release library reads no environment; the C25519_* knobs (A/B arms, pass-size overrides that make small inputs run many passes) exist only
# in the tuning build of the same sources (make tune, csrc/msm_internal.h C25519_KNOB).  Tests that need a knob run a fresh process on that file.
TUNE_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "curve25519-dalek_amd", "lib", "libc25519hip_tune.so")


def tune_env(knobs=None, **more):
    """environment of a child process that loads the tuning build with the given C25519_* knobs set"""
    assert os.path.exists(TUNE_LIB), "run __graft_entry__.build() (make tune)"
    env = dict(os.environ, C25519_HIP_LIB=TUNE_LIB)
    env.update(knobs or {})
    env.update(more)
    return env


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child_argv(code):
    """argv of a child process that runs `code` against the library C25519_HIP_LIB of ITS environment names (the tuning / debug build).  The variable belongs to the
    TEST HARNESS: the package reads no environment (tests/test_abi_cpu.py asserts it); the child selects the build with an explicit select_library() call."""
    import sys
    pre = ("import os as _os, sys as _sys\n_sys.path.insert(0, %r)\n_p = _os.environ.get('C25519_HIP_LIB')\n"
           "if _p:\n    import curve25519_dalek_amd as _pkg\n    _pkg.select_library(_p)\n" % ROOT)
    return [sys.executable, "-c", pre + code]


def rand_bytes(seed, n, width=32):
    return np.random.default_rng(seed).integers(0, 256, size=(n, width), dtype=np.uint8)


def rand_scalars(seed, n):
    """uniform in [0, 2^252) -- canonical (reduced) scalars"""
    s = rand_bytes(seed, n)
    s[:, 31] &= 0x0F
    return s


def edge_scalars():
    vals = [0, 1, 2, 8, 31, 32, 33, 63, 64, 65, L - 1, L, L + 1, 2**252 - 1, 2**252, 2**255 - 1, 2**255 - 19,
            int.from_bytes(bytes([0xF8] + [0xFF] * 30 + [0x7F]), "little"), 2**254, 2**253 + 5, (1 << 255) - (1 << 200)]
    vals += [(1 << k) for k in range(0, 255, 17)] + [(1 << k) - 1 for k in range(5, 255, 23)]
    vals += [sum(32 << (6 * i) for i in range(42)), sum(31 << (6 * i) for i in range(42)), sum(63 << (6 * i) for i in range(42))]
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32)
