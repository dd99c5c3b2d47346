"""Independent big-int cross-oracle (pure Python, affine formulas straight from RFC 7748 / RFC 8032).

Shares no code and no representation with oracle/ (C, radix-2^51 limbs) or the HIP engine
(radix-2^25.5 limbs): a disagreement between any two of the three is a bug in one of them.
Slow -- use for small cases only.
"""
import hashlib

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def inv(x):
    return pow(x, P - 2, P)


def ed_add(p, q):
    (x1, y1), (x2, y2) = p, q
    t = D * x1 * x2 * y1 * y2 % P
    return ((x1 * y2 + x2 * y1) * inv(1 + t) % P, (y1 * y2 + x1 * x2) * inv(1 - t) % P)


def ed_mul(k, p):
    r = (0, 1)
    while k:
        if k & 1:
            r = ed_add(r, p)
        p = ed_add(p, p)
        k >>= 1
    return r


def ed_neg(p):
    return ((-p[0]) % P, p[1])


def recover_x(y, sign):
    """RFC 8032 5.1.3 but WITHOUT the y<p / x=0&&sign rejections (the reference's ZIP-215 rules)."""
    y %= P
    u, v = (y * y - 1) % P, (D * y * y + 1) % P
    x2 = u * inv(v) % P
    x = pow(x2, (P + 3) // 8, P)
    if (x * x - x2) % P != 0:
        x = x * SQRT_M1 % P
    if (x * x - x2) % P != 0:
        return None
    if x & 1:
        x = P - x
    if sign:
        x = (P - x) % P
    return x


def ed_decompress(b):
    n = int.from_bytes(b, "little")
    y, sign = n & (2**255 - 1), n >> 255
    x = recover_x(y, sign)
    return None if x is None else (x, y % P)


def ed_compress(p):
    x, y = p
    return ((y % P) | ((x & 1) << 255)).to_bytes(32, "little")


BY = 4 * inv(5) % P
BX = recover_x(BY, 0)
B = (BX, BY)


def x25519(k, u):
    """RFC 7748 section 5."""
    kb = bytearray(k)
    kb[0] &= 248; kb[31] &= 127; kb[31] |= 64
    kn = int.from_bytes(kb, "little")
    x1 = int.from_bytes(u, "little") & (2**255 - 1)
    x1 %= P
    x2, z2, x3, z3, swap = 1, 0, x1, 1, 0
    for t in reversed(range(255)):
        kt = (kn >> t) & 1
        swap ^= kt
        if swap:
            x2, x3, z2, z3 = x3, x2, z3, z2
        swap = kt
        A = (x2 + z2) % P; AA = A * A % P
        Bv = (x2 - z2) % P; BB = Bv * Bv % P
        E = (AA - BB) % P
        Cv = (x3 + z3) % P; Dv = (x3 - z3) % P
        DA = Dv * A % P; CB = Cv * Bv % P
        x3 = (DA + CB) ** 2 % P
        z3 = x1 * (DA - CB) ** 2 % P
        x2 = AA * BB % P
        z2 = E * (AA + 121665 * E) % P
    if swap:
        x2, x3, z2, z3 = x3, x2, z3, z2
    return (x2 * inv(z2) % P).to_bytes(32, "little")


def sha512_modl(*parts):
    h = hashlib.sha512()
    for p in parts:
        h.update(p)
    return int.from_bytes(h.digest(), "little") % L


def ed25519_pubkey(sk):
    h = bytearray(hashlib.sha512(sk).digest()[:32])
    h[0] &= 248; h[31] &= 127; h[31] |= 64
    return ed_compress(ed_mul(int.from_bytes(h, "little"), B))


def ed25519_sign(sk, msg):
    hh = hashlib.sha512(sk).digest()
    a = bytearray(hh[:32]); a[0] &= 248; a[31] &= 127; a[31] |= 64
    a = int.from_bytes(a, "little")
    A = ed_compress(ed_mul(a, B))
    r = sha512_modl(hh[32:], msg)
    R = ed_compress(ed_mul(r, B))
    k = sha512_modl(R, A, msg)
    return R + ((r + k * a) % L).to_bytes(32, "little")


def ed25519_verify_cofactorless(pk, msg, sig):
    """The reference's single-signature rule (verifying.rs:549-556): s canonical, A decodes (ZIP-215
    decoding), and compress([s]B - [k]A) == R bytes."""
    A = ed_decompress(pk)
    if A is None:
        return False
    s = int.from_bytes(sig[32:], "little")
    if s >= L:
        return False
    k = sha512_modl(sig[:32], pk, msg)
    Rc = ed_add(ed_mul(s, B), ed_neg(ed_mul(k, A)))
    return ed_compress(Rc) == sig[:32]
