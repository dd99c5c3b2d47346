"""Shared helpers for the GPU parity tests: seeded input generators (numpy) and oracle batch calls."""
import numpy as np

L = 2**252 + 27742317777372353535851937790883648493
P = 2**255 - 19


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
