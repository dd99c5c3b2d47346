"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only; the product
package (curve25519-dalek_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ST_OK, ST_NONE, ST_SCALAR_FORMAT, ST_VERIFY, ST_ARRAY_LENGTH = 0, 1, 2, 3, 4


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("ORACLE_LIB") or os.path.join(HERE, "liboracle.so")      # ORACLE_LIB: the sanitizer build (tests)
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def _b(x, n=None):
    x = bytes(x)
    if n is not None:
        assert len(x) == n, (len(x), n)
    return x


def _out(n):
    return C.create_string_buffer(n)


def _call_b(name, *ins, out=32):
    o = _out(out)
    getattr(lib(), name)(*ins, o)
    return o.raw


# ---- field ----
def fe_mul(a, b): return _call_b("orc_fe_mul", _b(a, 32), _b(b, 32))
def fe_sq(a): return _call_b("orc_fe_sq", _b(a, 32))
def fe_add(a, b): return _call_b("orc_fe_add", _b(a, 32), _b(b, 32))
def fe_sub(a, b): return _call_b("orc_fe_sub", _b(a, 32), _b(b, 32))
def fe_neg(a): return _call_b("orc_fe_neg", _b(a, 32))
def fe_invert(a): return _call_b("orc_fe_invert", _b(a, 32))
def fe_pow_p58(a): return _call_b("orc_fe_pow_p58", _b(a, 32))
def fe_canon(a): return _call_b("orc_fe_canon", _b(a, 32))


def fe_sqrt_ratio_i(u, v):
    o = _out(32)
    ok = lib().orc_fe_sqrt_ratio_i(_b(u, 32), _b(v, 32), o)
    return bool(ok), o.raw


def fe_invert_batch(elems):
    buf = C.create_string_buffer(b"".join(elems), 32 * len(elems))
    lib().orc_fe_invert_batch(buf, C.c_size_t(len(elems)))
    return [buf.raw[32 * i:32 * i + 32] for i in range(len(elems))]


def fe_mul_limbs(a, b):
    A = (C.c_uint64 * 5)(*a); B = (C.c_uint64 * 5)(*b); O = (C.c_uint64 * 5)()
    lib().orc_fe_mul_limbs(A, B, O)
    return list(O)


def fe_to_bytes_limbs(a):
    A = (C.c_uint64 * 5)(*a); o = _out(32)
    lib().orc_fe_to_bytes_limbs(A, o)
    return o.raw


# ---- scalars ----
def sc_reduce(a): return _call_b("orc_sc_from_bytes_mod_order", _b(a, 32))
def sc_reduce_wide(a): return _call_b("orc_sc_from_bytes_mod_order_wide", _b(a, 64))
def sc_is_canonical(a): return bool(lib().orc_sc_is_canonical(_b(a, 32)))
def sc_mul(a, b): return _call_b("orc_sc_mul", _b(a, 32), _b(b, 32))
def sc_add(a, b): return _call_b("orc_sc_add", _b(a, 32), _b(b, 32))
def sc_sub(a, b): return _call_b("orc_sc_sub", _b(a, 32), _b(b, 32))
def sc_neg(a): return _call_b("orc_sc_neg", _b(a, 32))


def sc_naf(s, w):
    o = (C.c_int8 * 256)(); lib().orc_sc_naf(_b(s, 32), C.c_uint(w), o); return list(o)


def sc_radix16(s):
    o = (C.c_int8 * 64)(); lib().orc_sc_radix16(_b(s, 32), o); return list(o)


def sc_radix2w(s, w):
    o = (C.c_int8 * 64)(); lib().orc_sc_radix2w(_b(s, 32), C.c_uint(w), o); return list(o)


def sc_clamp(b):
    buf = C.create_string_buffer(_b(b, 32), 32); lib().orc_sc_clamp(buf); return buf.raw


# ---- Edwards points: opaque 160-byte blobs ({X,Y,Z,T} x 5 x u64) ----
def _pt(): return _out(160)


def ed_basepoint():
    o = _pt(); lib().orc_ed_basepoint(o); return o.raw


def ed_identity():
    o = _pt(); lib().orc_ed_identity(o); return o.raw


def ed_decompress(b):
    o = _pt()
    return o.raw if lib().orc_ed_decompress(_b(b, 32), o) else None


def ed_compress(p): return _call_b("orc_ed_compress", _b(p, 160))


def ed_compress_batch(pts):
    o = _out(32 * len(pts))
    lib().orc_ed_compress_batch(b"".join(pts), C.c_size_t(len(pts)), o)
    return [o.raw[32 * i:32 * i + 32] for i in range(len(pts))]


def ed_add(a, b): return _call_b("orc_ed_add", _b(a, 160), _b(b, 160), out=160)
def ed_sub(a, b): return _call_b("orc_ed_sub", _b(a, 160), _b(b, 160), out=160)
def ed_neg(a): return _call_b("orc_ed_neg", _b(a, 160), out=160)
def ed_double(a): return _call_b("orc_ed_double", _b(a, 160), out=160)
def ed_mul_by_pow_2(a, k): return _call_b("orc_ed_mul_by_pow_2", _b(a, 160), C.c_uint(k), out=160)
def ed_eq(a, b): return bool(lib().orc_ed_eq(_b(a, 160), _b(b, 160)))
def ed_is_identity(a): return bool(lib().orc_ed_is_identity(_b(a, 160)))
def ed_is_small_order(a): return bool(lib().orc_ed_is_small_order(_b(a, 160)))
def ed_is_torsion_free(a): return bool(lib().orc_ed_is_torsion_free(_b(a, 160)))
def ed_mul(p, s): return _call_b("orc_ed_mul", _b(p, 160), _b(s, 32), out=160)
def ed_mul_base(s): return _call_b("orc_ed_mul_base", _b(s, 32), out=160)
def ed_to_montgomery(p): return _call_b("orc_ed_to_montgomery", _b(p, 160))


def ed_double_scalar_mul_basepoint(a, A, b):
    return _call_b("orc_ed_double_scalar_mul_basepoint", _b(a, 32), _b(A, 160), _b(b, 32), out=160)


def ed_basepoint_table_entry(i, j):
    o = (C.c_uint64 * 15)(); lib().orc_ed_basepoint_table_entry(C.c_uint(i), C.c_uint(j), o); return list(o)


def ed_naf8_basepoint_entry(i):
    o = (C.c_uint64 * 15)(); lib().orc_ed_naf8_basepoint_entry(C.c_uint(i), o); return list(o)


def ed_msm(scalars, points, which=0):
    """scalars: list of 32-byte, points: list of 160-byte blobs; which 0=dispatch 1=straus 2=pippenger"""
    assert len(scalars) == len(points)
    o = _pt()
    lib().orc_ed_msm_vartime(b"".join(scalars), b"".join(points), C.c_size_t(len(scalars)), C.c_int(which), o)
    return o.raw


def ed_msm_np(scalars, points, which=0):
    """the same over contiguous numpy arrays (n x 32, n x 160 uint8): no per-item Python work in a timed region"""
    import numpy as np
    s = np.ascontiguousarray(scalars, dtype=np.uint8); p = np.ascontiguousarray(points, dtype=np.uint8)
    assert s.ndim == 2 and s.shape[1] == 32 and p.shape == (s.shape[0], 160)
    o = _pt()
    lib().orc_ed_msm_vartime(s.ctypes.data_as(C.c_char_p), p.ctypes.data_as(C.c_char_p), C.c_size_t(s.shape[0]), C.c_int(which), o)
    return o.raw


# ---- Ristretto ----
def ris_decompress(b):
    o = _pt()
    return o.raw if lib().orc_ris_decompress(_b(b, 32), o) else None


def ris_compress(p): return _call_b("orc_ris_compress", _b(p, 160))
def ris_eq(a, b): return bool(lib().orc_ris_eq(_b(a, 160), _b(b, 160)))


# ---- Montgomery ----
def mont_mul(u, s): return _call_b("orc_mont_mul", _b(u, 32), _b(s, 32))
def x25519(k, u): return _call_b("orc_x25519", _b(k, 32), _b(u, 32))


# ---- hashes ----
def sha512(m):
    o = _out(64); lib().orc_sha512(bytes(m), C.c_size_t(len(m)), o); return o.raw


def sha3_256(m):
    o = _out(32); lib().orc_sha3_256(bytes(m), C.c_size_t(len(m)), o); return o.raw


def merlin_simple(proto, l1, msg, l2, outlen):
    o = _out(outlen)
    lib().orc_merlin_simple(proto, l1, bytes(msg), C.c_size_t(len(msg)), l2, o, C.c_size_t(outlen))
    return o.raw


# ---- Ed25519 ----
def strobe_script(proto, ops):
    """ops: list of (name, more, payload) with name in meta_ad / ad / prf / key; payload = bytes to absorb, or the number of
    bytes to squeeze for prf.  -> the concatenated PRF outputs (orc_strobe_script)."""
    code = {"meta_ad": 0, "ad": 1, "prf": 2, "key": 3}
    enc, data, outlen = b"", b"", 0
    for name, more, payload in ops:
        n = payload if name == "prf" else len(payload)
        enc += bytes([code[name], 1 if more else 0]) + int(n).to_bytes(4, "little")
        if name == "prf":
            outlen += n
        else:
            data += payload
    o = _out(max(outlen, 1))
    lib().orc_strobe_script(proto, C.c_size_t(len(proto)), enc, C.c_size_t(len(ops)), data, o)
    return o.raw[:outlen]


def ed25519_pubkey(sk):
    return _call_b("orc_ed25519_pubkey", _b(sk, 32))


def ed25519_sign(sk, msg):
    o = _out(64); lib().orc_ed25519_sign(_b(sk, 32), bytes(msg), C.c_size_t(len(msg)), o); return o.raw


def ed25519_verify(pk, msg, sig):
    return lib().orc_ed25519_verify(_b(pk, 32), bytes(msg), C.c_size_t(len(msg)), _b(sig, 64))


def ed25519_verify_strict(pk, msg, sig):
    return lib().orc_ed25519_verify_strict(_b(pk, 32), bytes(msg), C.c_size_t(len(msg)), _b(sig, 64))


def ed25519_sign_prehashed(sk, prehash, context=b""):
    """Ed25519ph (signing.rs:917-976): prehash = the 64-byte SHA-512 of the message; -> (status, signature)"""
    o = _out(64)
    st = lib().orc_ed25519_sign_prehashed(_b(sk, 32), _b(prehash, 64), bytes(context), C.c_size_t(len(context)), o)
    return st, o.raw


def ed25519_verify_prehashed(pk, prehash, sig, context=b"", strict=False):
    """verifying.rs:230-257 (strict: :424-461) -> status"""
    return lib().orc_ed25519_verify_prehashed(_b(pk, 32), _b(prehash, 64), bytes(context), C.c_size_t(len(context)), _b(sig, 64), 1 if strict else 0)


def _pack_msgs(msgs):
    off = np.zeros(len(msgs) + 1, dtype=np.uint64)
    for i, m in enumerate(msgs):
        off[i + 1] = off[i] + len(m)
    return b"".join(msgs), off


def batch_transcript_zs(hrams, ss):
    n = len(hrams)
    o = _out(16 * max(n, 1))
    lib().orc_batch_transcript_zs(b"".join(hrams), b"".join(ss), C.c_size_t(n), o)
    return [o.raw[16 * i:16 * i + 16] for i in range(n)]


def ed25519_verify_batch(msgs, sigs, pks, zs=None):
    """Mirrors ed25519_dalek::verify_batch (batch.rs:146); returns a status code."""
    if not (len(msgs) == len(sigs) == len(pks)):
        return ST_ARRAY_LENGTH
    blob, off = _pack_msgs(msgs)
    zarg = None if zs is None else b"".join(zs)
    return lib().orc_ed25519_verify_batch_z(blob, off.ctypes.data_as(C.c_void_p), b"".join(sigs), b"".join(pks),
                                            C.c_size_t(len(msgs)), zarg)


# ---- bulk drivers (numpy uint8 arrays, shape (n,32)) ----
def mul_base_compress_batch(scalars, threads=1):
    s = np.ascontiguousarray(scalars, dtype=np.uint8); n = s.shape[0]
    out = np.empty((n, 32), dtype=np.uint8)
    lib().orc_mul_base_compress_batch(s.ctypes.data_as(C.c_void_p), C.c_size_t(n), out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def x25519_batch(k, u, threads=1):
    k = np.ascontiguousarray(k, dtype=np.uint8); u = np.ascontiguousarray(u, dtype=np.uint8); n = k.shape[0]
    out = np.empty((n, 32), dtype=np.uint8)
    lib().orc_x25519_batch(k.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), C.c_size_t(n), out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def ed_decompress_ok_batch(enc, threads=1):
    e = np.ascontiguousarray(enc, dtype=np.uint8); n = e.shape[0]
    out = np.empty(n, dtype=np.uint8)
    lib().orc_ed_decompress_ok_batch(e.ctypes.data_as(C.c_void_p), C.c_size_t(n), out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def ed25519_keygen_sign_batch(seeds, msgs, threads=1):
    """seeds (n,32) uint8, msgs (n,mlen) uint8 -> (pks (n,32), sigs (n,64))"""
    sd = np.ascontiguousarray(seeds, dtype=np.uint8); m = np.ascontiguousarray(msgs, dtype=np.uint8)
    n, mlen = sd.shape[0], (m.shape[1] if m.ndim == 2 else 0)
    pks = np.empty((n, 32), dtype=np.uint8); sigs = np.empty((n, 64), dtype=np.uint8)
    lib().orc_ed25519_keygen_sign_batch(sd.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), C.c_size_t(mlen), C.c_size_t(n),
                                        pks.ctypes.data_as(C.c_void_p), sigs.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return pks, sigs


def ed_msm_mt_np(scalars, points, threads=1):
    """`threads` independent slices, each the reference's single-threaded MSM, partial sums added (bench.py cpu_baseline all_cores)"""
    s = np.ascontiguousarray(scalars, dtype=np.uint8); p = np.ascontiguousarray(points, dtype=np.uint8)
    assert s.ndim == 2 and s.shape[1] == 32 and p.shape == (s.shape[0], 160)
    o = _pt()
    lib().orc_ed_msm_vartime_mt(s.ctypes.data_as(C.c_char_p), p.ctypes.data_as(C.c_char_p), C.c_size_t(s.shape[0]), C.c_int(threads), o)
    return o.raw


def ed25519_verify_batch_mt_np(msgs, mlen, sigs, pks, threads=1):
    """fixed-length messages as an (n, mlen) array; `threads` independent verify_batch calls over contiguous slices -> worst verdict"""
    m = np.ascontiguousarray(msgs, dtype=np.uint8); sg = np.ascontiguousarray(sigs, dtype=np.uint8); pk = np.ascontiguousarray(pks, dtype=np.uint8)
    n = sg.shape[0]
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(mlen))
    return lib().orc_ed25519_verify_batch_mt(m.ctypes.data_as(C.c_char_p), off.ctypes.data_as(C.c_void_p), sg.ctypes.data_as(C.c_char_p),
                                             pk.ctypes.data_as(C.c_char_p), C.c_size_t(n), C.c_int(threads))
