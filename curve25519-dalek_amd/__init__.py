"""curve25519-dalek_amd -- MI355X (gfx950) batched Curve25519 engine.

Only what the hot path needs: csrc/ (HIP kernels + the C ABI of include/c25519_hip.h), the ctypes
binding (engine.py) and the host-side mirror of the reference's interface for this path (dalek.py).
There is no CPU fallback: constructing an Engine without the built library or without a GPU raises.
"""
from .engine import Engine, EngineError, lib_path, load_library, select_library  # noqa: F401
from . import dalek  # noqa: F401
from . import multi  # noqa: F401

__all__ = ["Engine", "EngineError", "lib_path", "load_library", "select_library", "dalek", "multi"]
