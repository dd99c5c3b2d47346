"""ctypes binding of libc25519hip.so (include/c25519_hip.h).

PyTorch is plumbing here: device buffers are torch uint8 CUDA tensors, the engine enqueues on
torch's current stream, and torch.distributed carries the multi-GPU exchange.  The arithmetic is
all in the HIP library; if it is missing this module raises -- it never falls back to a CPU path.
"""
import ctypes as C
import os

import numpy as np

OK, NONE, SCALAR_FORMAT, VERIFY, ARRAY_LENGTH, PREHASHED_CONTEXT_LENGTH = 0, 1, 2, 3, 4, 5
FMT_EDWARDS_Y, FMT_RISTRETTO, FMT_RAW160 = 0, 1, 2
POINT_DECODES, POINT_SMALL_ORDER, POINT_TORSION_FREE = 1, 2, 4      # flags of c25519_point_order_checks_batch
Z_TRANSCRIPT, Z_DEVICE = 0, 1
FLAG_VARTIME_TABLES = 0x100      # c25519_ctx_create: fast secret-indexed tables for mul_base / mul_batch / sign / keygen (public scalars only)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class EngineError(RuntimeError):
    pass


_LIB_OVERRIDE = None


def lib_path():
    """the library this process loads: lib/libc25519hip.so unless select_library() named another build.  Nothing here reads the environment."""
    return _LIB_OVERRIDE or os.path.join(_HERE, "lib", "libc25519hip.so")


def select_library(path):
    """Load another build of the same sources (the tuning build lib/libc25519hip_tune.so with its A/B knobs, the bound-checking debug build, a variant
    made by tools/build_variant.sh) instead of the release library.  An explicit call of the test harness / an A/B tool, before the first Engine; the
    package itself never looks at the environment for this (a crypto library must not be substitutable through a variable)."""
    global _LIB_OVERRIDE
    path = os.path.abspath(path)
    if _LIB is not None and os.path.abspath(lib_path()) != path:
        raise EngineError("select_library(%s): %s is already loaded in this process" % (path, lib_path()))
    _LIB_OVERRIDE = path


def load_library():
    """Load the HIP library (after torch, so both share one HIP runtime).  Raises if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise EngineError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % path)
    try:
        import torch  # noqa: F401  (loads libamdhip64 first; our library binds to the same runtime)
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(path)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int32
    sigs = {
        "c25519_ctx_create": (vp, [C.c_int, C.c_uint32]),
        "c25519_ctx_destroy": (None, [vp]),
        "c25519_ctx_set_stream": (i32, [vp, vp]),
        "c25519_ctx_synchronize": (i32, [vp]),
        "c25519_last_error": (C.c_char_p, [vp]),
        "c25519_last_kernel_ms": (C.c_float, [vp]),
        "c25519_phase_ms": (C.c_float, [vp, C.c_uint32, C.c_int]),
        "c25519_last_call_phase_ms": (C.c_float, [vp, C.c_int, vp]),
        "c25519_debug_batch_zs": (i32, [vp, vp, vp, vp, vp, u64, C.c_uint32, vp]),
        "c25519_debug_sort": (i32, [vp, vp, u64, u64, C.c_int32]),
        "c25519_mul_base_batch_dev": (i32, [vp, vp, u64, C.c_int, vp]),
        "c25519_mul_base_batch": (i32, [vp, vp, u64, C.c_int, vp]),
        "c25519_mul_base_batch_vartime_dev": (i32, [vp, vp, u64, C.c_int, vp]),
        "c25519_mul_base_clamped_batch_dev": (i32, [vp, vp, u64, C.c_int, vp]),
        "c25519_mul_base_clamped_batch": (i32, [vp, vp, u64, C.c_int, vp]),
        "c25519_basetable_create": (vp, [vp, vp, C.c_int]),
        "c25519_basetable_destroy": (None, [vp, vp]),
        "c25519_mul_table_batch_dev": (i32, [vp, vp, vp, u64, C.c_int, vp]),
        "c25519_mul_table_batch": (i32, [vp, vp, vp, u64, C.c_int, vp]),
        "c25519_x25519_contributory_batch_dev": (i32, [vp, vp, vp, u64, vp, vp]),
        "c25519_x25519_contributory_batch": (i32, [vp, vp, vp, u64, vp, vp]),
        "c25519_mul_clamped_batch_dev": (i32, [vp, vp, vp, u64, C.c_int, C.c_int, vp, vp]),
        "c25519_mul_clamped_batch": (i32, [vp, vp, vp, u64, C.c_int, C.c_int, vp, vp]),
        "c25519_point_order_checks_batch_dev": (i32, [vp, vp, u64, C.c_int, C.c_int, vp]),
        "c25519_point_order_checks_batch": (i32, [vp, vp, u64, C.c_int, C.c_int, vp]),
        "c25519_host_alloc": (vp, [C.c_size_t]),
        "c25519_host_free": (None, [vp]),
        "c25519_last_ffi_ms": (C.c_double, [vp, vp, vp]),
        "c25519_ctx_trim": (i32, [vp]),
        "c25519_last_kernel_name": (C.c_char_p, [vp, C.c_int]),
        "c25519_x25519_batch_dev": (i32, [vp, vp, vp, u64, vp]),
        "c25519_x25519_batch": (i32, [vp, vp, vp, u64, vp]),
        "c25519_x25519_base_batch_dev": (i32, [vp, vp, u64, vp]),
        "c25519_x25519_base_batch": (i32, [vp, vp, u64, vp]),
        "c25519_decompress_batch_dev": (i32, [vp, vp, u64, C.c_int, vp, vp]),
        "c25519_decompress_batch": (i32, [vp, vp, u64, C.c_int, vp, vp]),
        "c25519_compress_batch_dev": (i32, [vp, vp, u64, C.c_int, vp]),
        "c25519_compress_batch": (i32, [vp, vp, u64, C.c_int, vp]),
        "c25519_to_montgomery_batch_dev": (i32, [vp, vp, u64, vp]),
        "c25519_to_montgomery_batch": (i32, [vp, vp, u64, vp]),
        "c25519_msm_vartime_dev": (i32, [vp, vp, vp, u64, C.c_int, C.c_int, vp]),
        "c25519_msm_vartime": (i32, [vp, vp, vp, u64, C.c_int, C.c_int, vp]),
        "c25519_msm_partial_dev": (i32, [vp, vp, vp, u64, C.c_int, vp]),
        "c25519_fold_partials": (i32, [vp, vp, u64, C.c_int, vp]),
        "c25519_msm_partial_record_dev": (i32, [vp, vp, vp, u64, C.c_int, vp]),
        "c25519_fold_partial_records": (i32, [vp, vp, u64, C.c_int, vp]),
        "c25519_partial_record_pack": (i32, [vp, i32, vp, vp]),
        "ed25519_batch_hram_dev": (i32, [vp, vp, vp, u64, vp, vp, u64, vp]),
        "ed25519_batch_transcript_zs": (i32, [vp, vp, u64, vp]),
        "ed25519_verify_batch_record_dev": (i32, [vp, vp, vp, vp, vp, vp, u64, vp]),
        "ed25519_fold_verify_records": (i32, [vp, vp, u64]),
        "c25519_msm_vartime_multi": (i32, [vp, i32, vp, vp, u64, C.c_int, C.c_int, vp]),
        "ed25519_verify_batch_multi": (i32, [vp, i32, vp, vp, vp, vp, u64, C.c_uint32]),
        "ed25519_verify_batch_dev": (i32, [vp, vp, vp, u64, vp, vp, u64, C.c_uint32]),
        "ed25519_verify_batch": (i32, [vp, vp, vp, vp, vp, u64, C.c_uint32]),
        "ed25519_verify_batch_keys_dev": (i32, [vp, vp, vp, u64, vp, vp, vp, u64, C.c_uint32]),
        "ed25519_verify_batch_keys": (i32, [vp, vp, vp, vp, vp, vp, u64, C.c_uint32]),
        "c25519_mul_batch_dev": (i32, [vp, vp, vp, u64, C.c_int, C.c_int, vp, vp]),
        "c25519_mul_batch": (i32, [vp, vp, vp, u64, C.c_int, C.c_int, vp, vp]),
        "c25519_double_base_batch_dev": (i32, [vp, vp, vp, vp, u64, C.c_int, C.c_int, vp, vp]),
        "c25519_double_base_batch": (i32, [vp, vp, vp, vp, u64, C.c_int, C.c_int, vp, vp]),
        "c25519_last_call_host_us": (i32, [vp, vp]),
        "c25519_ctx_counter": (u64, [vp, i32]),
        "ed25519_verify_each_dev": (i32, [vp, vp, vp, u64, vp, vp, u64, C.c_int, vp]),
        "ed25519_verify_each": (i32, [vp, vp, vp, vp, vp, u64, C.c_int, vp]),
        "ed25519_verify_each_prehashed_dev": (i32, [vp, vp, C.c_char_p, C.c_uint32, vp, vp, u64, C.c_int, vp]),
        "ed25519_verify_each_prehashed": (i32, [vp, vp, C.c_char_p, C.c_uint32, vp, vp, u64, C.c_int, vp]),
        "ed25519_sign_batch_prehashed_dev": (i32, [vp, vp, vp, C.c_char_p, C.c_uint32, u64, vp, vp]),
        "ed25519_sign_batch_prehashed": (i32, [vp, vp, vp, C.c_char_p, C.c_uint32, u64, vp, vp]),
        "ed25519_keygen_batch_dev": (i32, [vp, vp, u64, vp]),
        "ed25519_sign_batch_dev": (i32, [vp, vp, vp, vp, u64, u64, vp, vp]),
        "ed25519_sign_batch": (i32, [vp, vp, vp, vp, u64, vp, vp]),
        "c25519_precomp_create": (vp, [vp, vp, u64, C.c_int]),
        "c25519_precomp_destroy": (None, [vp, vp]),
        "c25519_precomp_len": (u64, [vp]),
        "c25519_precomp_msm_vartime": (i32, [vp, vp, vp, u64, vp, vp, u64, C.c_int, C.c_int, vp]),
        "c25519_msm_consttime": (i32, [vp, vp, vp, u64, C.c_int, C.c_int, vp]),
        "c25519_double_and_compress_batch_dev": (i32, [vp, vp, u64, vp]),
        "c25519_double_and_compress_batch": (i32, [vp, vp, u64, vp]),
        "c25519_scalar_invert_batch": (i32, [vp, vp, u64, vp]),
        "c25519_microbench": (C.c_double, [vp, C.c_int, C.c_int]),
        "c25519_selftest_field": (i32, [vp, C.c_int, C.c_int, vp, vp, u64, vp]),
        "c25519_selftest_scalar": (i32, [vp, C.c_int, vp, vp, u64, vp]),
        "c25519_msm_geometry": (i32, [u64, vp, vp, vp, vp, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not export the ABI
        fn.restype, fn.argtypes = res, args
    _LIB = lib
    return lib


ABI_SYMBOLS = [
    "c25519_ctx_create", "c25519_ctx_destroy", "c25519_ctx_set_stream", "c25519_ctx_synchronize", "c25519_last_error",
    "c25519_mul_base_clamped_batch_dev", "c25519_mul_base_clamped_batch", "c25519_basetable_create", "c25519_basetable_destroy", "c25519_mul_table_batch_dev",
    "c25519_mul_table_batch", "c25519_x25519_contributory_batch_dev", "c25519_x25519_contributory_batch", "c25519_mul_clamped_batch_dev", "c25519_mul_clamped_batch",
    "c25519_point_order_checks_batch_dev", "c25519_point_order_checks_batch",
    "c25519_host_alloc", "c25519_host_free", "c25519_last_ffi_ms", "c25519_ctx_trim", "c25519_last_kernel_name",
    "c25519_last_kernel_ms", "c25519_phase_ms", "c25519_last_call_phase_ms", "c25519_debug_batch_zs", "c25519_debug_sort", "c25519_mul_base_batch_dev", "c25519_mul_base_batch_vartime_dev", "c25519_mul_base_batch", "c25519_x25519_batch_dev",
    "c25519_x25519_batch", "c25519_x25519_base_batch_dev", "c25519_x25519_base_batch", "c25519_decompress_batch_dev", "c25519_decompress_batch", "c25519_compress_batch_dev",
    "c25519_compress_batch", "c25519_msm_vartime_dev", "c25519_msm_vartime", "c25519_msm_partial_dev",
    "c25519_fold_partials", "c25519_msm_partial_record_dev", "c25519_fold_partial_records", "c25519_partial_record_pack",
    "ed25519_batch_hram_dev", "ed25519_batch_transcript_zs", "ed25519_verify_batch_record_dev", "ed25519_fold_verify_records", "c25519_msm_vartime_multi", "ed25519_verify_batch_multi", "ed25519_verify_batch_dev", "ed25519_verify_batch", "ed25519_verify_batch_keys_dev", "ed25519_verify_batch_keys", "c25519_microbench", "c25519_selftest_field", "c25519_selftest_scalar", "c25519_msm_geometry",
    "c25519_mul_batch_dev", "c25519_mul_batch", "c25519_double_base_batch_dev", "c25519_double_base_batch", "ed25519_verify_each_dev", "ed25519_verify_each",
    "ed25519_keygen_batch_dev", "ed25519_sign_batch_dev", "ed25519_sign_batch",
    "c25519_last_call_host_us", "c25519_ctx_counter", "ed25519_verify_each_prehashed_dev", "ed25519_verify_each_prehashed", "ed25519_sign_batch_prehashed_dev", "ed25519_sign_batch_prehashed",
    "c25519_to_montgomery_batch_dev", "c25519_to_montgomery_batch",
    "c25519_precomp_create", "c25519_precomp_destroy", "c25519_precomp_len", "c25519_precomp_msm_vartime",
    "c25519_msm_consttime", "c25519_double_and_compress_batch_dev", "c25519_double_and_compress_batch",
    "c25519_scalar_invert_batch",
]

_PT = {FMT_EDWARDS_Y: 32, FMT_RISTRETTO: 32, FMT_RAW160: 160}
PARTIAL_RECORD_BYTES = 9024      # C25519_PARTIAL_RECORD_BYTES


def _np8(x, width):
    a = np.ascontiguousarray(np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray)) else x, dtype=np.uint8)
    return a.reshape(-1, width)


class Engine:
    """One engine = one GPU + one stream (`c25519_ctx`).  Tensor methods (`*_t`) take/return torch
    uint8 CUDA tensors already resident in HBM; the plain methods take/return numpy arrays / bytes
    and go through the host-pointer entry points (PCIe copies included)."""

    def __init__(self, device=0, window=0, flags=0):
        """window: fixed-base table algorithm (c25519_ctx_create flags & 0x1f); flags: FLAG_VARTIME_TABLES or 0"""
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise EngineError("no GPU visible to PyTorch-ROCm: the engine has no CPU fallback")
        self.device = torch.device("cuda", device)
        # (round 5 let C25519_DEFAULT_WINDOW / C25519_DEFAULT_VARTIME_TABLES steer these two from the environment: a default context could be talked out of
        #  its constant-time tables by a variable.  The arguments are the only way now; tests pass them.)
        self.ctx = self.lib.c25519_ctx_create(device, (window & 0x1f) | flags)
        if not self.ctx:
            raise EngineError("c25519_ctx_create(%d) failed" % device)
        self._bind_stream()

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.c25519_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ----------------------------------------------------------------------------
    def _bind_stream(self):
        if not self.ctx:
            raise EngineError("this Engine has been closed")
        st = self.torch.cuda.current_stream(self.device).cuda_stream
        self.lib.c25519_ctx_set_stream(self.ctx, C.c_void_p(st))

    def _chk(self, st, allowed=(OK,)):
        if st < 0:
            raise EngineError("HIP error %d: %s" % (-st, self.lib.c25519_last_error(self.ctx).decode()))
        if st not in allowed:
            raise EngineError("unexpected status %d" % st)
        return st

    def _t(self, t, width):
        torch = self.torch
        assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous(), "need a contiguous uint8 CUDA tensor"
        assert t.numel() % width == 0
        assert t.data_ptr() % 16 == 0, "device buffers must be 16-byte aligned"
        return t.numel() // width

    def last_call_host_us(self):
        """host clock of the latest synchronous MSM / verify_batch call, microseconds since its entry: (upload enqueued, kernels enqueued, results on the host, folded)"""
        out = (C.c_double * 4)()
        self._chk(self.lib.c25519_last_call_host_us(self.ctx, out))
        return tuple(out)

    def counter(self, which):
        """event counters of the context (c25519_ctx_counter): 0 publications that blocked on the stream, 1 lost publications (re-run), 2 directly published calls"""
        return int(self.lib.c25519_ctx_counter(self.ctx, which))

    def synchronize(self):
        self._chk(self.lib.c25519_ctx_synchronize(self.ctx))

    def last_kernel_ms(self):
        return float(self.lib.c25519_last_kernel_ms(self.ctx))

    def phase_ms(self, back=0, phase=0):
        return float(self.lib.c25519_phase_ms(self.ctx, back, phase))

    def last_call_phase_ms(self, phase=0):
        """-> (ms summed over the passes of the latest MSM / verify_batch call, number of passes)"""
        passes = C.c_uint32(0)
        ms = float(self.lib.c25519_last_call_phase_ms(self.ctx, phase, C.byref(passes)))
        return ms, int(passes.value)

    def selftest_field(self, op, a_limbs, b_limbs=None, chain=0):
        """one field operation per row on the GPU: (n, 10) uint32 limbs in -> (n, 32) canonical bytes (c25519_selftest_field)"""
        a = np.ascontiguousarray(a_limbs, dtype=np.uint32).reshape(-1, 10); n = a.shape[0]
        b = None if b_limbs is None else np.ascontiguousarray(b_limbs, dtype=np.uint32).reshape(-1, 10)
        assert b is None or b.shape[0] == n
        out = np.empty((n, 32), dtype=np.uint8)
        self._bind_stream()
        self._chk(self.lib.c25519_selftest_field(self.ctx, op, chain, a.ctypes.data, b.ctypes.data if b is not None else None, n, out.ctypes.data))
        return out

    def selftest_scalar(self, op, a_words, b_words=None):
        """one sc28.h operation per row on the GPU: (n, 16) uint32 per operand -> (n, 32) bytes (c25519_selftest_scalar)"""
        a = np.ascontiguousarray(a_words, dtype=np.uint32).reshape(-1, 16); n = a.shape[0]
        b = None if b_words is None else np.ascontiguousarray(b_words, dtype=np.uint32).reshape(-1, 16)
        assert b is None or b.shape[0] == n
        out = np.empty((n, 32), dtype=np.uint8)
        self._bind_stream()
        self._chk(self.lib.c25519_selftest_scalar(self.ctx, op, a.ctypes.data, b.ctypes.data if b is not None else None, n, out.ctypes.data))
        return out

    def microbench(self, which, iters=2000):
        self._bind_stream()
        return float(self.lib.c25519_microbench(self.ctx, which, iters))

    # -- device-tensor API -------------------------------------------------------------------
    def mul_base_batch_t(self, scalars, out_fmt=FMT_EDWARDS_Y, out=None):
        n = self._t(scalars, 32)
        if out is None:
            out = self.torch.empty((n, _PT[out_fmt]), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.c25519_mul_base_batch_dev(self.ctx, scalars.data_ptr(), n, out_fmt, out.data_ptr()))
        return out

    def mul_base_batch_vartime_t(self, scalars, out_fmt=FMT_EDWARDS_Y, out=None):
        """scalars declared PUBLIC: the fast tables whatever the context's flags"""
        n = self._t(scalars, 32)
        if out is None:
            out = self.torch.empty((n, _PT[out_fmt]), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.c25519_mul_base_batch_vartime_dev(self.ctx, scalars.data_ptr(), n, out_fmt, out.data_ptr()))
        return out

    def x25519_base_batch_t(self, k, out=None):
        n = self._t(k, 32)
        if out is None:
            out = self.torch.empty((n, 32), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.c25519_x25519_base_batch_dev(self.ctx, k.data_ptr(), n, out.data_ptr()))
        return out

    def x25519_batch_t(self, k, u, out=None):
        n = self._t(k, 32)
        assert self._t(u, 32) == n
        if out is None:
            out = self.torch.empty((n, 32), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.c25519_x25519_batch_dev(self.ctx, k.data_ptr(), u.data_ptr(), n, out.data_ptr()))
        return out

    def decompress_batch_t(self, enc, in_fmt=FMT_EDWARDS_Y):
        n = self._t(enc, 32)
        pts = self.torch.empty((n, 160), dtype=self.torch.uint8, device=self.device)
        ok = self.torch.empty((n,), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        st = self._chk(self.lib.c25519_decompress_batch_dev(self.ctx, enc.data_ptr(), n, in_fmt, pts.data_ptr(), ok.data_ptr()), (OK, NONE))
        return st, pts, ok

    def compress_batch_t(self, pts, out_fmt=FMT_EDWARDS_Y):
        n = self._t(pts, 160)
        out = self.torch.empty((n, 32), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.c25519_compress_batch_dev(self.ctx, pts.data_ptr(), n, out_fmt, out.data_ptr()))
        return out

    def msm_vartime_t(self, scalars, points, in_fmt=FMT_RAW160, out_fmt=FMT_EDWARDS_Y):
        """-> (status, bytes).  status NONE mirrors Option::None of the reference."""
        n = self._t(scalars, 32)
        assert self._t(points, _PT[in_fmt]) == n
        out = C.create_string_buffer(_PT[out_fmt])
        self._bind_stream()
        st = self._chk(self.lib.c25519_msm_vartime_dev(self.ctx, scalars.data_ptr(), points.data_ptr(), n, in_fmt, out_fmt, out), (OK, NONE))
        return st, out.raw

    def msm_partial_t(self, scalars, points, in_fmt=FMT_RAW160):
        n = self._t(scalars, 32)
        assert self._t(points, _PT[in_fmt]) == n
        out = C.create_string_buffer(160)
        self._bind_stream()
        st = self._chk(self.lib.c25519_msm_partial_dev(self.ctx, scalars.data_ptr(), points.data_ptr(), n, in_fmt, out), (OK, NONE))
        return st, out.raw

    def msm_partial_record_t(self, scalars, points, in_fmt=FMT_RAW160, out=None):
        """This rank's share of a sharded MSM as a partial-result RECORD in device memory (uint8 tensor of
        PARTIAL_RECORD_BYTES).  Enqueue only: nothing has waited for the host when this returns."""
        n = self._t(scalars, 32)
        assert self._t(points, _PT[in_fmt]) == n
        if out is None:
            out = self.torch.empty((PARTIAL_RECORD_BYTES,), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.c25519_msm_partial_record_dev(self.ctx, scalars.data_ptr(), points.data_ptr(), n, in_fmt, out.data_ptr()))
        return out

    def batch_hram_t(self, msgs, msg_off, sigs, pks):
        """H(R_i || A_i || M_i) of this shard: uint8 device tensor of n * 64 + 64 bytes (the trailer holds the counters of
        ed25519_batch_hram_dev).  Enqueue only."""
        n = self._t(sigs, 64)
        assert self._t(pks, 32) == n and msg_off.numel() == n + 1
        out = self.torch.empty((n * 64 + 64,), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.ed25519_batch_hram_dev(self.ctx, msgs.data_ptr(), msg_off.data_ptr(), msgs.numel(), sigs.data_ptr(), pks.data_ptr(), n, out.data_ptr()))
        return out

    def verify_batch_record_t(self, sigs, pks, hram, z16, pk_points=None):
        """This shard's share of the batch equation with GIVEN z_i (n x 16 device tensor) and its hram buffer from
        batch_hram_t, as a partial-result record in device memory.  Enqueue only."""
        n = self._t(sigs, 64)
        assert self._t(pks, 32) == n and hram.numel() == n * 64 + 64 and z16.numel() == n * 16
        assert hram.is_cuda and z16.is_cuda and z16.is_contiguous() and (n == 0 or z16.data_ptr() % 16 == 0)
        if pk_points is not None:
            assert self._t(pk_points, 160) == n
        out = self.torch.empty((PARTIAL_RECORD_BYTES,), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.ed25519_verify_batch_record_dev(self.ctx, sigs.data_ptr(), pks.data_ptr(), pk_points.data_ptr() if pk_points is not None else None,
                                                           hram.data_ptr(), z16.data_ptr(), n, out.data_ptr()))
        return out

    def fold_partials(self, partials, out_fmt=FMT_EDWARDS_Y):
        blob = b"".join(partials)
        out = C.create_string_buffer(_PT[out_fmt])
        self._chk(self.lib.c25519_fold_partials(self.ctx, blob, len(partials), out_fmt, out))
        return out.raw

    def verify_batch_t(self, msgs, msg_off, sigs, pks, z_mode=Z_TRANSCRIPT, pk_points=None):
        """msgs: uint8 CUDA tensor of the concatenated messages; msg_off: int64/uint64 CUDA tensor (n+1);
        pk_points: optional (n, 160) raw points of the keys (VerifyingKey.point), skips their decompression."""
        n = self._t(sigs, 64)
        assert self._t(pks, 32) == n and msg_off.numel() == n + 1
        if pk_points is not None:
            assert self._t(pk_points, 160) == n
        self._bind_stream()
        return self._chk(self.lib.ed25519_verify_batch_keys_dev(self.ctx, msgs.data_ptr(), msg_off.data_ptr(), msgs.numel(),
                                                                sigs.data_ptr(), pks.data_ptr(), pk_points.data_ptr() if pk_points is not None else None,
                                                                n, z_mode),
                         (OK, NONE, SCALAR_FORMAT, VERIFY))

    def mul_batch_t(self, scalars, points, in_fmt=FMT_RAW160, out_fmt=FMT_EDWARDS_Y):
        n = self._t(scalars, 32)
        assert self._t(points, _PT[in_fmt]) == n
        out = self.torch.empty((n, _PT[out_fmt]), dtype=self.torch.uint8, device=self.device)
        ok = self.torch.empty((n,), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.c25519_mul_batch_dev(self.ctx, scalars.data_ptr(), points.data_ptr(), n, in_fmt, out_fmt, out.data_ptr(), ok.data_ptr()))
        return out, ok

    def double_base_batch_t(self, a, A, b, in_fmt=FMT_RAW160, out_fmt=FMT_EDWARDS_Y):
        n = self._t(a, 32)
        assert self._t(A, _PT[in_fmt]) == n and self._t(b, 32) == n
        out = self.torch.empty((n, _PT[out_fmt]), dtype=self.torch.uint8, device=self.device)
        ok = self.torch.empty((n,), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.c25519_double_base_batch_dev(self.ctx, a.data_ptr(), A.data_ptr(), b.data_ptr(), n, in_fmt, out_fmt, out.data_ptr(), ok.data_ptr()))
        return out, ok

    def verify_each_t(self, msgs, msg_off, sigs, pks, strict=False):
        n = self._t(sigs, 64)
        assert self._t(pks, 32) == n and msg_off.numel() == n + 1
        status = self.torch.empty((n,), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.ed25519_verify_each_dev(self.ctx, msgs.data_ptr(), msg_off.data_ptr(), msgs.numel(), sigs.data_ptr(), pks.data_ptr(),
                                                   n, 1 if strict else 0, status.data_ptr()))
        return status

    def verify_each_prehashed_t(self, prehashes, sigs, pks, context=b"", strict=False):
        """Ed25519ph on device tensors (prehashes: n x 64 bytes).  -> status tensor, or PREHASHED_CONTEXT_LENGTH for a context beyond 255 bytes"""
        n = self._t(sigs, 64)
        assert self._t(pks, 32) == n and self._t(prehashes, 64) == n
        status = self.torch.empty((n,), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        st = self._chk(self.lib.ed25519_verify_each_prehashed_dev(self.ctx, prehashes.data_ptr(), bytes(context), len(context), sigs.data_ptr(), pks.data_ptr(),
                                                                  n, 1 if strict else 0, status.data_ptr()), (OK, PREHASHED_CONTEXT_LENGTH))
        return status if st == OK else st

    def sign_batch_prehashed_t(self, seeds, prehashes, context=b""):
        n = self._t(seeds, 32)
        assert self._t(prehashes, 64) == n
        pks = self.torch.empty((n, 32), dtype=self.torch.uint8, device=self.device)
        sigs = self.torch.empty((n, 64), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        st = self._chk(self.lib.ed25519_sign_batch_prehashed_dev(self.ctx, seeds.data_ptr(), prehashes.data_ptr(), bytes(context), len(context), n, pks.data_ptr(), sigs.data_ptr()),
                       (OK, PREHASHED_CONTEXT_LENGTH))
        return (pks, sigs) if st == OK else st

    def keygen_batch_t(self, seeds):
        n = self._t(seeds, 32)
        pks = self.torch.empty((n, 32), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.ed25519_keygen_batch_dev(self.ctx, seeds.data_ptr(), n, pks.data_ptr()))
        return pks

    def sign_batch_t(self, seeds, msgs, msg_off):
        n = self._t(seeds, 32)
        assert msg_off.numel() == n + 1
        pks = self.torch.empty((n, 32), dtype=self.torch.uint8, device=self.device)
        sigs = self.torch.empty((n, 64), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.ed25519_sign_batch_dev(self.ctx, seeds.data_ptr(), msgs.data_ptr(), msg_off.data_ptr(), msgs.numel(), n, pks.data_ptr(), sigs.data_ptr()))
        return pks, sigs

    # -- host-buffer API (numpy in / numpy out) ---------------------------------------------------
    @staticmethod
    def _out(out, n, width):
        """the caller's output buffer (reuse it: a fresh allocation pays first-touch page faults inside the copy) or a new one"""
        if out is None:
            return np.empty((n, width) if width > 1 else (n,), dtype=np.uint8)
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.size == n * width
        return out

    def mul_base_batch(self, scalars, out_fmt=FMT_EDWARDS_Y, out=None):
        s = _np8(scalars, 32); n = s.shape[0]
        out = self._out(out, n, _PT[out_fmt])
        self._bind_stream()
        self._chk(self.lib.c25519_mul_base_batch(self.ctx, s.ctypes.data, n, out_fmt, out.ctypes.data))
        return out

    def mul_base_clamped_batch(self, raw, out_fmt=FMT_EDWARDS_Y, out=None):
        """EdwardsPoint::mul_base_clamped (edwards.rs:948): clamp_integer(raw_i) * B, not reduced mod l"""
        s = _np8(raw, 32); n = s.shape[0]
        out = self._out(out, n, _PT[out_fmt])
        self._bind_stream()
        self._chk(self.lib.c25519_mul_base_clamped_batch(self.ctx, s.ctypes.data, n, out_fmt, out.ctypes.data))
        return out

    def mul_clamped_batch(self, raw, points, in_fmt=FMT_RAW160, out_fmt=FMT_EDWARDS_Y):
        """EdwardsPoint::mul_clamped (edwards.rs:932): clamp_integer(raw_i) * P_i"""
        s = _np8(raw, 32); p = _np8(points, _PT[in_fmt]); n = s.shape[0]
        assert p.shape[0] == n
        out = np.empty((n, _PT[out_fmt]), dtype=np.uint8); ok = np.empty((n,), dtype=np.uint8)
        self._bind_stream()
        self._chk(self.lib.c25519_mul_clamped_batch(self.ctx, s.ctypes.data, p.ctypes.data, n, in_fmt, out_fmt, out.ctypes.data, ok.ctypes.data))
        return out, ok

    def point_order_checks(self, points, in_fmt=FMT_EDWARDS_Y, which=POINT_SMALL_ORDER | POINT_TORSION_FREE):
        """EdwardsPoint::is_small_order / is_torsion_free (edwards.rs:1405 / :1435) -> (n,) uint8 flags: POINT_DECODES | POINT_SMALL_ORDER | POINT_TORSION_FREE"""
        p = _np8(points, _PT[in_fmt]); n = p.shape[0]
        fl = np.empty((n,), dtype=np.uint8)
        self._bind_stream()
        self._chk(self.lib.c25519_point_order_checks_batch(self.ctx, p.ctypes.data, n, in_fmt, which, fl.ctypes.data))
        return fl

    def point_order_checks_t(self, points, in_fmt=FMT_EDWARDS_Y, which=POINT_SMALL_ORDER | POINT_TORSION_FREE):
        n = self._t(points, _PT[in_fmt])
        fl = self.torch.empty((n,), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.c25519_point_order_checks_batch_dev(self.ctx, points.data_ptr(), n, in_fmt, which, fl.data_ptr()))
        return fl

    # -- constant-time fixed-base tables for a caller's point (EdwardsBasepointTable::create / RistrettoBasepointTable::create)
    def basetable_create(self, point, in_fmt=FMT_EDWARDS_Y):
        b = bytes(point)
        assert len(b) == _PT[in_fmt]
        self._bind_stream()
        h = self.lib.c25519_basetable_create(self.ctx, b, in_fmt)
        if not h:
            raise EngineError("basetable_create failed: %s" % self.lib.c25519_last_error(self.ctx).decode())
        return h

    def basetable_destroy(self, h):
        self.lib.c25519_basetable_destroy(self.ctx, h)

    def mul_table_batch(self, h, scalars, out_fmt=FMT_EDWARDS_Y, out=None):
        s = _np8(scalars, 32); n = s.shape[0]
        out = self._out(out, n, _PT[out_fmt])
        self._bind_stream()
        self._chk(self.lib.c25519_mul_table_batch(self.ctx, h, s.ctypes.data, n, out_fmt, out.ctypes.data))
        return out

    def mul_table_batch_t(self, h, scalars, out_fmt=FMT_EDWARDS_Y, out=None):
        n = self._t(scalars, 32)
        if out is None:
            out = self.torch.empty((n, _PT[out_fmt]), dtype=self.torch.uint8, device=self.device)
        self._bind_stream()
        self._chk(self.lib.c25519_mul_table_batch_dev(self.ctx, h, scalars.data_ptr(), n, out_fmt, out.data_ptr()))
        return out

    def last_ffi(self):
        """-> (wall-clock ms, bytes up, bytes down) of the latest host-pointer call"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        ms = float(self.lib.c25519_last_ffi_ms(self.ctx, C.byref(a), C.byref(b)))
        return ms, int(a.value), int(b.value)

    def trim(self):
        self._chk(self.lib.c25519_ctx_trim(self.ctx))

    def x25519_base_batch(self, k, out=None):
        """X25519 public keys x25519(k_i, 9) through the fixed-base path (x25519.rs:105-109)."""
        k = _np8(k, 32); n = k.shape[0]
        out = self._out(out, n, 32)
        self._bind_stream()
        self._chk(self.lib.c25519_x25519_base_batch(self.ctx, k.ctypes.data, n, out.ctypes.data))
        return out

    def x25519_batch(self, k, u, out=None):
        k = _np8(k, 32); u = _np8(u, 32); n = k.shape[0]
        assert u.shape[0] == n
        out = self._out(out, n, 32)
        self._bind_stream()
        self._chk(self.lib.c25519_x25519_batch(self.ctx, k.ctypes.data, u.ctypes.data, n, out.ctypes.data))
        return out

    def x25519_contributory_batch(self, k, u):
        """-> (shared secrets (n, 32), was_contributory (n,)): x25519.rs:335 as a batched flag"""
        k = _np8(k, 32); u = _np8(u, 32); n = k.shape[0]
        assert u.shape[0] == n
        out = np.empty((n, 32), dtype=np.uint8); fl = np.empty((n,), dtype=np.uint8)
        self._bind_stream()
        self._chk(self.lib.c25519_x25519_contributory_batch(self.ctx, k.ctypes.data, u.ctypes.data, n, out.ctypes.data, fl.ctypes.data))
        return out, fl

    def decompress_batch(self, enc, in_fmt=FMT_EDWARDS_Y):
        e = _np8(enc, 32); n = e.shape[0]
        pts = np.empty((n, 160), dtype=np.uint8); ok = np.empty((n,), dtype=np.uint8)
        self._bind_stream()
        st = self._chk(self.lib.c25519_decompress_batch(self.ctx, e.ctypes.data, n, in_fmt, pts.ctypes.data, ok.ctypes.data), (OK, NONE))
        return st, pts, ok

    def compress_batch(self, pts, out_fmt=FMT_EDWARDS_Y):
        p = _np8(pts, 160); n = p.shape[0]
        out = np.empty((n, 32), dtype=np.uint8)
        self._bind_stream()
        self._chk(self.lib.c25519_compress_batch(self.ctx, p.ctypes.data, n, out_fmt, out.ctypes.data))
        return out

    def msm_vartime(self, scalars, points, in_fmt=FMT_RAW160, out_fmt=FMT_EDWARDS_Y):
        s = _np8(scalars, 32); p = _np8(points, _PT[in_fmt]); n = s.shape[0]
        assert p.shape[0] == n
        out = C.create_string_buffer(_PT[out_fmt])
        self._bind_stream()
        st = self._chk(self.lib.c25519_msm_vartime(self.ctx, s.ctypes.data, p.ctypes.data, n, in_fmt, out_fmt, out), (OK, NONE))
        return st, out.raw

    def verify_batch(self, msgs, sigs, pks, z_mode=Z_TRANSCRIPT, pk_points=None):
        """msgs: list of bytes; sigs: list of 64-byte; pks: list of 32-byte; pk_points: optional (n, 160) uint8
        array of the keys' decompressed points (VerifyingKey.point).  Status code out."""
        if not (len(msgs) == len(sigs) == len(pks)):
            return ARRAY_LENGTH  # batch.rs:152-165
        n = len(msgs)
        self._check_items(sigs, 64, "signature"); self._check_items(pks, 32, "public key")
        off = np.zeros(n + 1, dtype=np.uint64)
        for i, m in enumerate(msgs):
            off[i + 1] = off[i] + len(m)
        blob = np.frombuffer(b"".join(msgs) + b"\0" * 16, dtype=np.uint8)
        s = _np8(b"".join(sigs) if n else b"", 64) if n else np.zeros((0, 64), np.uint8)
        p = _np8(b"".join(pks) if n else b"", 32) if n else np.zeros((0, 32), np.uint8)
        self._bind_stream()
        pp = None
        if pk_points is not None:
            pp = _np8(pk_points, 160)
            assert pp.shape[0] == n
        return self._chk(self.lib.ed25519_verify_batch_keys(self.ctx, blob.ctypes.data, off.ctypes.data, s.ctypes.data, p.ctypes.data,
                                                            pp.ctypes.data if pp is not None else None, n, z_mode),
                         (OK, NONE, SCALAR_FORMAT, VERIFY))

    def debug_batch_zs(self, msgs, sigs, pks, z_mode=Z_DEVICE):
        """The z_i of a batch as an (n, 16) uint8 array (diagnostics, see c25519_debug_batch_zs)."""
        n = len(msgs)
        assert len(sigs) == n and len(pks) == n
        self._check_items(sigs, 64, "signature"); self._check_items(pks, 32, "public key")
        out = np.zeros((n, 16), dtype=np.uint8)
        if n == 0:
            return out
        blob, off = self._pack(list(msgs))
        s = _np8(b"".join(sigs), 64); p = _np8(b"".join(pks), 32)
        self._bind_stream()
        self._chk(self.lib.c25519_debug_batch_zs(self.ctx, blob.ctypes.data, off.ctypes.data, s.ctypes.data, p.ctypes.data, n, z_mode, out.ctypes.data))
        return out

    def mul_batch(self, scalars, points, in_fmt=FMT_RAW160, out_fmt=FMT_EDWARDS_Y):
        s = _np8(scalars, 32); p = _np8(points, _PT[in_fmt]); n = s.shape[0]
        assert p.shape[0] == n
        out = np.empty((n, _PT[out_fmt]), dtype=np.uint8); ok = np.empty((n,), dtype=np.uint8)
        self._bind_stream()
        self._chk(self.lib.c25519_mul_batch(self.ctx, s.ctypes.data, p.ctypes.data, n, in_fmt, out_fmt, out.ctypes.data, ok.ctypes.data))
        return out, ok

    def double_base_batch(self, a, A, b, in_fmt=FMT_RAW160, out_fmt=FMT_EDWARDS_Y):
        """out[i] = a[i]*A[i] + b[i]*B (vartime_double_scalar_mul_basepoint, edwards.rs:1099)."""
        sa = _np8(a, 32); p = _np8(A, _PT[in_fmt]); sb = _np8(b, 32); n = sa.shape[0]
        assert p.shape[0] == n and sb.shape[0] == n
        out = np.empty((n, _PT[out_fmt]), dtype=np.uint8); ok = np.empty((n,), dtype=np.uint8)
        self._bind_stream()
        self._chk(self.lib.c25519_double_base_batch(self.ctx, sa.ctypes.data, p.ctypes.data, sb.ctypes.data, n, in_fmt, out_fmt, out.ctypes.data, ok.ctypes.data))
        return out, ok

    @staticmethod
    def _check_items(items, width, what):
        for i, it in enumerate(items):
            if len(it) != width:
                raise ValueError("%s %d has %d bytes, expected %d" % (what, i, len(it), width))

    @staticmethod
    def _pack(msgs):
        n = len(msgs)
        off = np.zeros(n + 1, dtype=np.uint64)
        for i, m in enumerate(msgs):
            off[i + 1] = off[i] + len(m)
        return np.frombuffer(b"".join(msgs) + b"\0" * 16, dtype=np.uint8), off

    def verify_each(self, msgs, sigs, pks, strict=False):
        """-> numpy uint8 status per signature (0 OK, 1 bad key, 2 ScalarFormat, 3 Verify)."""
        n = len(msgs)
        assert len(sigs) == n and len(pks) == n
        status = np.empty((n,), dtype=np.uint8)
        if n == 0:
            return status
        self._check_items(sigs, 64, "signature"); self._check_items(pks, 32, "public key")
        blob, off = self._pack(list(msgs))
        s = _np8(b"".join(sigs), 64); p = _np8(b"".join(pks), 32)
        self._bind_stream()
        self._chk(self.lib.ed25519_verify_each(self.ctx, blob.ctypes.data, off.ctypes.data, s.ctypes.data, p.ctypes.data, n, 1 if strict else 0, status.ctypes.data))
        return status

    def verify_each_prehashed(self, prehashes, sigs, pks, context=b"", strict=False):
        """Ed25519ph / Ed25519ctx per signature (verifying.rs:284 / :424): prehashes = the 64-byte SHA-512 of each message.
        -> numpy uint8 status per signature, or PREHASHED_CONTEXT_LENGTH (an int) when the context exceeds 255 bytes."""
        n = len(prehashes)
        assert len(sigs) == n and len(pks) == n
        status = np.empty((n,), dtype=np.uint8)
        self._check_items(sigs, 64, "signature"); self._check_items(pks, 32, "public key"); self._check_items(prehashes, 64, "prehash")
        ph = _np8(b"".join(prehashes), 64) if n else np.empty((0, 64), np.uint8)
        s = _np8(b"".join(sigs), 64) if n else np.empty((0, 64), np.uint8); p = _np8(b"".join(pks), 32) if n else np.empty((0, 32), np.uint8)
        self._bind_stream()
        st = self._chk(self.lib.ed25519_verify_each_prehashed(self.ctx, ph.ctypes.data, bytes(context), len(context), s.ctypes.data, p.ctypes.data, n, 1 if strict else 0,
                                                              status.ctypes.data), (OK, PREHASHED_CONTEXT_LENGTH))
        return status if st == OK else st

    def sign_batch_prehashed(self, seeds, prehashes, context=b""):
        """Ed25519ph signing (signing.rs:312): -> (pks, sigs) numpy arrays, or PREHASHED_CONTEXT_LENGTH (an int)."""
        n = len(seeds)
        assert len(prehashes) == n
        pks = np.empty((n, 32), dtype=np.uint8); sigs = np.empty((n, 64), dtype=np.uint8)
        self._check_items(seeds, 32, "seed"); self._check_items(prehashes, 64, "prehash")
        sd = _np8(b"".join(seeds), 32) if n else np.empty((0, 32), np.uint8); ph = _np8(b"".join(prehashes), 64) if n else np.empty((0, 64), np.uint8)
        self._bind_stream()
        st = self._chk(self.lib.ed25519_sign_batch_prehashed(self.ctx, sd.ctypes.data, ph.ctypes.data, bytes(context), len(context), n, pks.ctypes.data, sigs.ctypes.data),
                       (OK, PREHASHED_CONTEXT_LENGTH))
        return (pks, sigs) if st == OK else st

    def sign_batch(self, seeds, msgs):
        """-> (pks (n,32), sigs (n,64)) numpy arrays; seeds: list of 32-byte secret keys."""
        n = len(seeds)
        assert len(msgs) == n
        pks = np.empty((n, 32), dtype=np.uint8); sigs = np.empty((n, 64), dtype=np.uint8)
        if n == 0:
            return pks, sigs
        self._check_items(seeds, 32, "seed")
        blob, off = self._pack(list(msgs))
        sd = _np8(b"".join(seeds), 32)
        self._bind_stream()
        self._chk(self.lib.ed25519_sign_batch(self.ctx, sd.ctypes.data, blob.ctypes.data, off.ctypes.data, n, pks.ctypes.data, sigs.ctypes.data))
        return pks, sigs

    def to_montgomery_batch(self, pts):
        p = _np8(pts, 160); n = p.shape[0]
        out = np.empty((n, 32), dtype=np.uint8)
        self._bind_stream()
        self._chk(self.lib.c25519_to_montgomery_batch(self.ctx, p.ctypes.data, n, out.ctypes.data))
        return out

    # -- §8f widening ------------------------------------------------------------------------------------
    def precomp_create(self, static_points, in_fmt=FMT_RAW160):
        p = _np8(static_points, _PT[in_fmt])
        self._bind_stream()
        h = self.lib.c25519_precomp_create(self.ctx, p.ctypes.data, p.shape[0], in_fmt)
        if not h:
            raise EngineError("precomp_create failed: %s" % self.lib.c25519_last_error(self.ctx).decode())
        return h

    def precomp_destroy(self, h):
        self.lib.c25519_precomp_destroy(self.ctx, h)

    def precomp_len(self, h):
        return int(self.lib.c25519_precomp_len(h))

    def precomp_msm_vartime(self, h, static_scalars, dyn_scalars, dyn_points, in_fmt=FMT_RAW160, out_fmt=FMT_EDWARDS_Y):
        ss = _np8(static_scalars, 32); ds = _np8(dyn_scalars, 32); dp = _np8(dyn_points, _PT[in_fmt])
        assert ds.shape[0] == dp.shape[0]
        out = C.create_string_buffer(_PT[out_fmt])
        self._bind_stream()
        st = self._chk(self.lib.c25519_precomp_msm_vartime(self.ctx, h, ss.ctypes.data, ss.shape[0], ds.ctypes.data, dp.ctypes.data, ds.shape[0],
                                                           in_fmt, out_fmt, out), (OK, NONE))
        return st, out.raw

    def msm_consttime(self, scalars, points, in_fmt=FMT_RAW160, out_fmt=FMT_EDWARDS_Y):
        s = _np8(scalars, 32); p = _np8(points, _PT[in_fmt])
        assert s.shape[0] == p.shape[0]
        out = C.create_string_buffer(_PT[out_fmt])
        self._bind_stream()
        st = self._chk(self.lib.c25519_msm_consttime(self.ctx, s.ctypes.data, p.ctypes.data, s.shape[0], in_fmt, out_fmt, out), (OK, NONE))
        return st, out.raw

    def double_and_compress_batch(self, pts):
        p = _np8(pts, 160); n = p.shape[0]
        out = np.empty((n, 32), dtype=np.uint8)
        self._bind_stream()
        self._chk(self.lib.c25519_double_and_compress_batch(self.ctx, p.ctypes.data, n, out.ctypes.data))
        return out

    def scalar_invert_batch(self, scalars):
        """-> (inverses (n,32), product of all inverses (32 bytes)); inputs must be canonical and non-zero."""
        s = _np8(scalars, 32).copy(); n = s.shape[0]
        prod = C.create_string_buffer(32)
        self._bind_stream()
        self._chk(self.lib.c25519_scalar_invert_batch(self.ctx, s.ctypes.data, n, prod))
        return s, prod.raw


def fold_partial_records(records, out_fmt=FMT_EDWARDS_Y):
    """records: (count, PARTIAL_RECORD_BYTES) uint8 array on the HOST -> (status, bytes | None).  Host arithmetic in the C
    library (c25519_fold_partial_records); needs no GPU."""
    lib = load_library()
    r = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1, PARTIAL_RECORD_BYTES)
    out = C.create_string_buffer(_PT[out_fmt])
    st = lib.c25519_fold_partial_records(None, r.ctypes.data, r.shape[0], out_fmt, out)
    if st < 0:
        raise EngineError("c25519_fold_partial_records failed with status %d" % st)
    return (st, out.raw if st == OK else None)


def fold_verify_records(records):
    """records of every rank's share of ONE batch equation -> the reference's verdict (status code)."""
    lib = load_library()
    r = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1, PARTIAL_RECORD_BYTES)
    st = lib.ed25519_fold_verify_records(None, r.ctypes.data, r.shape[0])
    if st < 0:
        raise EngineError("ed25519_fold_verify_records failed with status %d" % st)
    return st


def partial_record_pack(point160, status=OK, counters=None):
    """A record holding a given 160-byte point (c25519_partial_record_pack) -> bytes."""
    lib = load_library()
    out = C.create_string_buffer(PARTIAL_RECORD_BYTES)
    cnt = None
    if counters is not None:
        cnt = (C.c_uint32 * 8)(*[int(c) for c in counters])
    st = lib.c25519_partial_record_pack(bytes(point160), status, cnt, out)
    if st != 0:
        raise EngineError("c25519_partial_record_pack failed with status %d" % st)
    return out.raw


def batch_transcript_zs(hram, sigs):
    """The reference's z_i (batch.rs:168-222) from every H(R||A||M) (n, 64) and every signature (n, 64; s = bytes 32..63)
    of the batch, on the host: -> (n, 16) uint8."""
    lib = load_library()
    h = _np8(hram, 64); g = _np8(sigs, 64); n = h.shape[0]
    assert g.shape[0] == n
    z = np.zeros((n, 16), dtype=np.uint8)
    lib.ed25519_batch_transcript_zs(h.ctypes.data, g.ctypes.data, n, z.ctypes.data)
    return z


def msm_vartime_multi(engines, scalars, points, in_fmt=FMT_RAW160, out_fmt=FMT_EDWARDS_Y):
    """c25519_msm_vartime_multi: one process, several contexts (GPUs); -> (status, bytes)"""
    s = _np8(scalars, 32); p = _np8(points, _PT[in_fmt]); n = s.shape[0]
    assert p.shape[0] == n and len(engines) >= 1
    arr = (C.c_void_p * len(engines))(*[e.ctx for e in engines])
    out = C.create_string_buffer(_PT[out_fmt])
    st = engines[0]._chk(engines[0].lib.c25519_msm_vartime_multi(arr, len(engines), s.ctypes.data, p.ctypes.data, n, in_fmt, out_fmt, out), (OK, NONE))
    return st, out.raw


def verify_batch_multi(engines, msgs, sigs, pks, z_mode=Z_TRANSCRIPT):
    """ed25519_verify_batch_multi: one process, several contexts (GPUs); -> status"""
    if not (len(msgs) == len(sigs) == len(pks)):
        return ARRAY_LENGTH
    n = len(msgs)
    if n == 0:
        return OK
    Engine._check_items(sigs, 64, "signature"); Engine._check_items(pks, 32, "public key")
    blob, off = Engine._pack(list(msgs))
    s = _np8(b"".join(sigs), 64); p = _np8(b"".join(pks), 32)
    arr = (C.c_void_p * len(engines))(*[e.ctx for e in engines])
    return engines[0]._chk(engines[0].lib.ed25519_verify_batch_multi(arr, len(engines), blob.ctypes.data, off.ctypes.data, s.ctypes.data, p.ctypes.data, n, z_mode),
                           (OK, NONE, SCALAR_FORMAT, VERIFY))
