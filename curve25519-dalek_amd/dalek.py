"""Host-side mirror of the reference's interface for the hot path (same names, argument meaning and
error behaviour), batched, on top of Engine.  Reference entry points mirrored:

  EdwardsPoint::mul_base                      curve25519-dalek/src/edwards.rs:918
  EdwardsPoint::vartime_multiscalar_mul       curve25519-dalek/src/traits.rs:249 / edwards.rs:1002
  RistrettoPoint::vartime_multiscalar_mul     curve25519-dalek/src/ristretto.rs:984
  CompressedEdwardsY::decompress / compress   edwards.rs:211 / :615
  x25519                                      x25519-dalek/src/x25519.rs:390
  verify_batch                                ed25519-dalek/src/batch.rs:146
  VerifyingKey::verify / verify_strict        ed25519-dalek/src/verifying.rs:565 / :359   (verify_each)
  SigningKey::sign                            ed25519-dalek/src/signing.rs:878-905        (sign_batch)
  VerifyingKey::verify_prehashed[_strict] / SigningKey::sign_prehashed   verifying.rs:284 / :424, signing.rs:312   (Ed25519ph: verify_each_prehashed, sign_batch_prehashed)
  EdwardsPoint::multiscalar_mul               curve25519-dalek/src/edwards.rs:966-1000
  VartimeEdwardsPrecomputation                curve25519-dalek/src/edwards.rs:1037-1076
  RistrettoPoint::double_and_compress_batch   curve25519-dalek/src/ristretto.rs:564
  Scalar::invert_batch                        curve25519-dalek/src/scalar.rs:802
  EdwardsBasepointTable::create / mul_base    curve25519-dalek/src/edwards.rs:1131-1141 / :1192-1209   (any basepoint)
  RistrettoBasepointTable::create             curve25519-dalek/src/ristretto.rs:1080-1110
  EdwardsPoint::mul_base_clamped / mul_clamped   curve25519-dalek/src/edwards.rs:948 / :932
  SharedSecret::was_contributory              x25519-dalek/src/x25519.rs:335
  EdwardsPoint::is_small_order / is_torsion_free   curve25519-dalek/src/edwards.rs:1405 / :1435;  VerifyingKey::is_weak  ed25519-dalek/src/verifying.rs:192

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

    @staticmethod
    def vartime_double_scalar_mul_basepoint(a, A, b, engine=None):
        """[a_i * A_i + b_i * B] (edwards.rs:1099-1106 over vartime_double_base.rs:23-72), batched: A_i as
        160-byte raw EdwardsPoints, results as CompressedEdwardsY bytes."""
        if not (len(a) == len(A) == len(b)):
            raise AssertionError("vartime_double_scalar_mul_basepoint: a, A, b must have equal length")
        eng = engine or default_engine()
        out, _ = eng.double_base_batch(_cat(a, 32), _cat(A, 160), _cat(b, 32), _e.FMT_RAW160, _e.FMT_EDWARDS_Y)
        return [out[i].tobytes() for i in range(out.shape[0])]


    @staticmethod
    def mul_base_clamped(raw_bytes, engine=None):
        """[clamp_integer(b_i) * B] as CompressedEdwardsY bytes (edwards.rs:948-956; the clamped integer is not reduced mod l)."""
        eng = engine or default_engine()
        out = eng.mul_base_clamped_batch(_cat(raw_bytes, 32), _e.FMT_EDWARDS_Y)
        return [out[i].tobytes() for i in range(out.shape[0])]

    @staticmethod
    def mul_clamped(points, raw_bytes, engine=None):
        """[clamp_integer(b_i) * P_i] (edwards.rs:932-946): P_i as CompressedEdwardsY, results likewise; None where P_i does not decode."""
        if len(points) != len(raw_bytes):
            raise AssertionError("mul_clamped: points and scalars must have equal length")
        eng = engine or default_engine()
        out, ok = eng.mul_clamped_batch(_cat(raw_bytes, 32), _cat(points, 32), _e.FMT_EDWARDS_Y, _e.FMT_EDWARDS_Y)
        return [out[i].tobytes() if ok[i] else None for i in range(out.shape[0])]


def _order_flags(points, which, engine):
    eng = engine or default_engine()
    return eng.point_order_checks(_cat(points, 32), _e.FMT_EDWARDS_Y, which)


def is_small_order(points, engine=None):
    """[CompressedEdwardsY] -> [EdwardsPoint::is_small_order(), or None where the encoding does not decode] (edwards.rs:1405)"""
    fl = _order_flags(points, _e.POINT_SMALL_ORDER, engine)
    return [bool(f & _e.POINT_SMALL_ORDER) if f & _e.POINT_DECODES else None for f in fl]


def is_torsion_free(points, engine=None):
    """[CompressedEdwardsY] -> [EdwardsPoint::is_torsion_free(), or None] (edwards.rs:1435)"""
    fl = _order_flags(points, _e.POINT_TORSION_FREE, engine)
    return [bool(f & _e.POINT_TORSION_FREE) if f & _e.POINT_DECODES else None for f in fl]


class EdwardsBasepointTable:
    """edwards.rs:1125-1209 for ANY basepoint: `create(&P)` builds the window table once, `mul_base(&s)` = `&s * &table`
    multiplies secret scalars by P with constant-time table lookups (window.rs:54-76 semantics) whatever the engine's flags."""

    _fmt = _e.FMT_EDWARDS_Y

    def __init__(self, basepoint, engine=None):
        self.eng = engine or default_engine()
        self.point = bytes(basepoint)
        self.h = self.eng.basetable_create(self.point, self._fmt)

    @classmethod
    def create(cls, basepoint, engine=None):
        return cls(basepoint, engine)

    def basepoint(self):
        return self.point

    def mul_base(self, scalars):
        out = self.eng.mul_table_batch(self.h, _cat(scalars, 32), self._fmt)
        return [out[i].tobytes() for i in range(out.shape[0])]

    def close(self):
        if self.h:
            self.eng.basetable_destroy(self.h)
            self.h = None


class RistrettoBasepointTable(EdwardsBasepointTable):
    """ristretto.rs:1080-1110: the same table over a CompressedRistretto basepoint, results as CompressedRistretto."""
    _fmt = _e.FMT_RISTRETTO


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


class SharedSecret:
    """x25519-dalek/src/x25519.rs:301-345: the 32 bytes of a Diffie-Hellman result and was_contributory() (:335)."""
    __slots__ = ("bytes", "contributory")

    def __init__(self, b, contributory):
        self.bytes, self.contributory = bytes(b), bool(contributory)

    def as_bytes(self):
        return self.bytes

    def was_contributory(self):
        return self.contributory


def diffie_hellman(secrets, their_publics, engine=None):
    """StaticSecret::diffie_hellman (x25519.rs:219-222), batched -> [SharedSecret]"""
    eng = engine or default_engine()
    out, fl = eng.x25519_contributory_batch(_cat(secrets, 32), _cat(their_publics, 32))
    return [SharedSecret(out[i].tobytes(), fl[i]) for i in range(out.shape[0])]


class VerifyingKey:
    """ed25519-dalek/src/verifying.rs:64-71: the 32 key bytes together with the decompressed point, so that
    verify_batch does not decompress A_i again (batch.rs:236)."""
    __slots__ = ("compressed", "point")
    _from_bytes_token = object()

    def __init__(self, compressed, point, _token=None):
        # the type invariant of the reference (point == decompress(compressed)) is what verify_batch relies on when it
        # hashes the bytes and multiplies the point: only from_bytes, which computes the point itself, may build one
        if _token is not VerifyingKey._from_bytes_token:
            raise TypeError("VerifyingKey objects are built by VerifyingKey.from_bytes (verifying.rs:167-175)")
        self.compressed, self.point = bytes(compressed), bytes(point)

    @staticmethod
    def is_weak(keys, engine=None):
        """VerifyingKey::is_weak (verifying.rs:192-194) for a list of VerifyingKey: the point has small order"""
        eng = engine or default_engine()
        fl = eng.point_order_checks(_cat([k.point for k in keys], 160), _e.FMT_RAW160, _e.POINT_SMALL_ORDER)
        return [bool(f & _e.POINT_SMALL_ORDER) for f in fl]

    @staticmethod
    def from_bytes(keys, engine=None):
        """VerifyingKey::from_bytes (verifying.rs:167-175), batched: a list of VerifyingKey; raises
        SignatureError("PointDecompression") if any key does not decode."""
        eng = engine or default_engine()
        _, pts, ok = eng.decompress_batch(_cat(keys, 32), _e.FMT_EDWARDS_Y)
        if len(keys) and not ok.all():
            raise SignatureError("PointDecompression")
        return [VerifyingKey(keys[i], pts[i].tobytes(), VerifyingKey._from_bytes_token) for i in range(len(keys))]

    def as_bytes(self):
        return self.compressed

    def __bytes__(self):
        return self.compressed


def verify_batch(messages, signatures, verifying_keys, engine=None, z_mode=_e.Z_TRANSCRIPT):
    """ed25519-dalek/src/batch.rs:146: returns None on success, raises SignatureError otherwise
    (ArrayLength / ScalarFormat / Verify in the reference's precedence; PointDecompression for a key
    that would have failed VerifyingKey::from_bytes, verifying.rs:167).  verifying_keys: 32-byte strings, or
    VerifyingKey objects (then their cached points are used, as in the reference)."""
    eng = engine or default_engine()
    keys = list(verifying_keys)
    pk_points = None
    if keys and all(isinstance(k, VerifyingKey) for k in keys):
        pk_points = _cat([k.point for k in keys], 160)
    st = eng.verify_batch(list(messages), list(signatures), [bytes(k) for k in keys], z_mode, pk_points=pk_points)
    if st == _e.OK:
        return None
    raise SignatureError({_e.ARRAY_LENGTH: "ArrayLength", _e.SCALAR_FORMAT: "ScalarFormat", _e.VERIFY: "Verify",
                          _e.NONE: "PointDecompression"}[st])


def x25519_public_keys(secrets, engine=None):
    """PublicKey::from(&StaticSecret | &EphemeralSecret) (x25519-dalek/src/x25519.rs:105-109, :255-259), batched:
    EdwardsPoint::mul_base_clamped(secret).to_montgomery() through the fixed-base tables, not the ladder."""
    eng = engine or default_engine()
    out = eng.x25519_base_batch(_cat(secrets, 32))
    return [out[i].tobytes() for i in range(out.shape[0])]


def verify_each(messages, signatures, verifying_keys, strict=False, engine=None):
    """Per-signature VerifyingKey::verify (verifying.rs:565) or verify_strict (:359): a list with None for
    Ok(()) and a SignatureError for every failing signature."""
    eng = engine or default_engine()
    st = eng.verify_each(list(messages), list(signatures), list(verifying_keys), strict)
    names = {_e.SCALAR_FORMAT: "ScalarFormat", _e.VERIFY: "Verify", _e.NONE: "PointDecompression"}
    return [None if c == _e.OK else SignatureError(names[int(c)]) for c in st]


def verify_each_prehashed(prehashed_messages, signatures, verifying_keys, context=None, strict=False, engine=None):
    """VerifyingKey::verify_prehashed (verifying.rs:284) / verify_prehashed_strict (:424) per signature -- Ed25519ph.  prehashed_messages: hashlib.sha512
    objects (the reference takes a Digest state and finalises it) or their 64-byte digests; context: None (= empty, as in the reference) or up to 255
    bytes, one for the batch.  -> a list with None for Ok(()) and a SignatureError otherwise.  A context beyond 255 bytes raises
    SignatureError("PrehashedContextLength") (the reference debug-asserts on verification and returns that error when signing)."""
    eng = engine or default_engine()
    st = eng.verify_each_prehashed([_digest64(m) for m in prehashed_messages], list(signatures), list(verifying_keys), context or b"", strict)
    if isinstance(st, int):
        raise SignatureError("PrehashedContextLength")
    names = {_e.SCALAR_FORMAT: "ScalarFormat", _e.VERIFY: "Verify", _e.NONE: "PointDecompression"}
    return [None if c == _e.OK else SignatureError(names[int(c)]) for c in st]


def sign_batch_prehashed(secret_keys, prehashed_messages, context=None, engine=None):
    """SigningKey::from_bytes(sk).sign_prehashed(digest, context) for every pair (signing.rs:312 -> :917-976): -> (verifying keys, signatures);
    raises SignatureError("PrehashedContextLength") for a context beyond 255 bytes (signing.rs:931-933)."""
    eng = engine or default_engine()
    r = eng.sign_batch_prehashed(list(secret_keys), [_digest64(m) for m in prehashed_messages], context or b"")
    if isinstance(r, int):
        raise SignatureError("PrehashedContextLength")
    pks, sigs = r
    return [pks[i].tobytes() for i in range(len(secret_keys))], [sigs[i].tobytes() for i in range(len(secret_keys))]


def _digest64(m):
    d = m.digest() if hasattr(m, "digest") else bytes(m)
    if len(d) != 64:
        raise SignatureError("prehashed message: a 512-bit digest is required")
    return d


def sign_batch(secret_keys, messages, engine=None):
    """SigningKey::from_bytes(sk).sign(m) for every pair (signing.rs:878-905): -> (verifying keys, signatures)."""
    eng = engine or default_engine()
    pks, sigs = eng.sign_batch(list(secret_keys), list(messages))
    return [pks[i].tobytes() for i in range(len(messages))], [sigs[i].tobytes() for i in range(len(messages))]


def multiscalar_mul(scalars, points, engine=None):
    """EdwardsPoint::multiscalar_mul (edwards.rs:966-1000), points as CompressedEdwardsY: regular schedule."""
    if len(scalars) != len(points):
        raise AssertionError("multiscalar_mul: scalars and points must have equal length")
    eng = engine or default_engine()
    st, out = eng.msm_consttime(_cat(scalars, 32), _cat(points, 32), _e.FMT_EDWARDS_Y, _e.FMT_EDWARDS_Y)
    return None if st == _e.NONE else out


class VartimeEdwardsPrecomputation:
    """edwards.rs:1037-1076 (VartimePrecomputedMultiscalarMul, traits.rs:304-419); points as CompressedEdwardsY."""

    def __init__(self, static_points, engine=None):
        self.eng = engine or default_engine()
        self.h = self.eng.precomp_create(_cat(static_points, 32), _e.FMT_EDWARDS_Y)

    def __len__(self):
        return self.eng.precomp_len(self.h)

    def is_empty(self):
        return len(self) == 0

    def vartime_mixed_multiscalar_mul(self, static_scalars, dynamic_scalars, dynamic_points):
        if len(dynamic_scalars) != len(dynamic_points):
            raise AssertionError("dynamic scalars and points must have equal length")   # precomputed_straus.rs:87
        st, out = self.eng.precomp_msm_vartime(self.h, _cat(static_scalars, 32), _cat(dynamic_scalars, 32), _cat(dynamic_points, 32),
                                               _e.FMT_EDWARDS_Y, _e.FMT_EDWARDS_Y)
        return None if st == _e.NONE else out

    def vartime_multiscalar_mul(self, static_scalars):
        return self.vartime_mixed_multiscalar_mul(static_scalars, [], [])

    def close(self):
        if self.h:
            self.eng.precomp_destroy(self.h)
            self.h = None
