"""Host-side mirror of the reference's interface for the hot path (same names, argument meaning and
error behaviour), batched, on top of Engine.  Reference entry points mirrored:

  EdwardsPoint::mul_base                      curve25519-dalek/src/edwards.rs:918
  EdwardsPoint::vartime_multiscalar_mul       curve25519-dalek/src/traits.rs:249 / edwards.rs:1002
  RistrettoPoint::vartime_multiscalar_mul     curve25519-dalek/src/ristretto.rs:984
  CompressedEdwardsY::decompress / compress   edwards.rs:211 / :615
  x25519                                      x25519-dalek/src/x25519.rs:390
  verify_batch                                ed25519-dalek/src/batch.rs:146

Values cross this layer as the reference's wire types: Scalar = 32 canonical LE bytes,
CompressedEdwardsY / CompressedRistretto / MontgomeryPoint = 32 bytes.
"""
import numpy as np

from . import engine as _e


class SignatureError(Exception):
    """ed25519-dalek/src/errors.rs:21-42 InternalError, by name."""

    def __init__(self, kind):
        super().__init__(kind)
        self.kind = kind


_ENGINE = None


def default_engine():
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = _e.Engine()
    return _ENGINE


def _cat(items, width):
    if len(items) == 0:
        return np.zeros((0, width), dtype=np.uint8)
    return np.frombuffer(b"".join(bytes(x) for x in items), dtype=np.uint8).reshape(-1, width)


class EdwardsPoint:
    @staticmethod
    def mul_base(scalars, engine=None):
        """[s_i * B] as CompressedEdwardsY bytes (edwards.rs:918 followed by compress :615)."""
        eng = engine or default_engine()
        out = eng.mul_base_batch(_cat(scalars, 32), _e.FMT_EDWARDS_Y)
        return [out[i].tobytes() for i in range(out.shape[0])]

    @staticmethod
    def vartime_multiscalar_mul(scalars, points, engine=None):
        """sum s_i P_i with P_i given as CompressedEdwardsY; returns CompressedEdwardsY bytes, or None
        if a point does not decompress (optional_multiscalar_mul, edwards.rs:1002-1031).  Unequal
        lengths raise, like the reference's assert_eq! (edwards.rs:1017-1019)."""
        if len(scalars) != len(points):
            raise AssertionError("vartime_multiscalar_mul: scalars and points must have equal length")
        eng = engine or default_engine()
        st, out = eng.msm_vartime(_cat(scalars, 32), _cat(points, 32), _e.FMT_EDWARDS_Y, _e.FMT_EDWARDS_Y)
        return None if st == _e.NONE else out


class RistrettoPoint:
    @staticmethod
    def vartime_multiscalar_mul(scalars, points, engine=None):
        """ristretto.rs:984: points and result as CompressedRistretto bytes."""
        if len(scalars) != len(points):
            raise AssertionError("vartime_multiscalar_mul: scalars and points must have equal length")
        eng = engine or default_engine()
        st, out = eng.msm_vartime(_cat(scalars, 32), _cat(points, 32), _e.FMT_RISTRETTO, _e.FMT_RISTRETTO)
        return None if st == _e.NONE else out


class CompressedEdwardsY:
    @staticmethod
    def decompress(encodings, engine=None):
        """-> list of 160-byte raw EdwardsPoints or None (edwards.rs:211-258)."""
        eng = engine or default_engine()
        _, pts, ok = eng.decompress_batch(_cat(encodings, 32), _e.FMT_EDWARDS_Y)
        return [pts[i].tobytes() if ok[i] else None for i in range(len(encodings))]


def x25519(ks, us, engine=None):
    """x25519-dalek/src/x25519.rs:390, batched."""
    eng = engine or default_engine()
    out = eng.x25519_batch(_cat(ks, 32), _cat(us, 32))
    return [out[i].tobytes() for i in range(out.shape[0])]


def verify_batch(messages, signatures, verifying_keys, engine=None, z_mode=_e.Z_TRANSCRIPT):
    """ed25519-dalek/src/batch.rs:146: returns None on success, raises SignatureError otherwise
    (ArrayLength / ScalarFormat / Verify in the reference's precedence; PointDecompression for a key
    that would have failed VerifyingKey::from_bytes, verifying.rs:167)."""
    eng = engine or default_engine()
    st = eng.verify_batch(list(messages), list(signatures), list(verifying_keys), z_mode)
    if st == _e.OK:
        return None
    raise SignatureError({_e.ARRAY_LENGTH: "ArrayLength", _e.SCALAR_FORMAT: "ScalarFormat", _e.VERIFY: "Verify",
                          _e.NONE: "PointDecompression"}[st])
