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


# ---------------------------------------------------------------------------------------------------------------------
# STROBE-128/1600 + Merlin, written from the public specifications (FIPS 202; STROBE v1.0.2, strobe.sourceforge.io/specs;
# merlin.cool/transcript) in their GENERAL form -- every operation goes through one `operate(flags, data, more)` that derives
# its duplex behaviour from the flag bits (cbefore / cafter), the Keccak round constants and rotation offsets are GENERATED
# (LFSR / (x, y) -> (y, 2x + 3y) walk), and the state is a byte array.  It shares nothing with oracle/hashes.h or
# csrc/transcript_host.h (which special-case the four operations they need, keep the state as 25 lanes and take their
# constants from a table): it is the third opinion on the z_i of verify_batch, including the KEY operation of the RNG
# finalisation (transcript.rs:157-173) that Merlin's published conformance vector never exercises.
# ---------------------------------------------------------------------------------------------------------------------
def _keccak_constants():
    rc, r = [], 1
    for _ in range(24):
        c = 0
        for j in range(7):                          # rc(t) of FIPS 202 algorithm 5: an LFSR over x^8 + x^6 + x^5 + x^4 + 1
            if r & 1:
                c |= 1 << ((1 << j) - 1)
            r <<= 1
            if r & 0x100:
                r ^= 0x171
        rc.append(c)
    rot = [[0] * 5 for _ in range(5)]
    x, y = 1, 0
    for t in range(24):                             # FIPS 202 algorithm 2 (rho)
        rot[x][y] = ((t + 1) * (t + 2) // 2) % 64
        x, y = y, (2 * x + 3 * y) % 5
    return rc, rot


_KRC, _KROT = _keccak_constants()
_M64 = (1 << 64) - 1


def keccak_f1600(state):
    """state: bytearray(200), permuted in place"""
    a = [[int.from_bytes(state[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & _M64 if n else v
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = rol(a[x][y], _KROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= _KRC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = a[x][y].to_bytes(8, "little")


def sponge(rate, suffix, msg, outlen):
    """plain Keccak sponge on keccak_f1600 (SHA3-256: rate 136, suffix 0x06; SHAKE128: rate 168, suffix 0x1f): pins the permutation"""
    st = bytearray(200)
    m = bytearray(msg) + bytes([suffix])
    m += bytes((-len(m)) % rate)
    m[-1] |= 0x80
    for off in range(0, len(m), rate):
        for i in range(rate):
            st[i] ^= m[off + i]
        keccak_f1600(st)
    out = b""
    while len(out) < outlen:
        out += bytes(st[:rate])
        if len(out) < outlen:
            keccak_f1600(st)
    return out[:outlen]


class Strobe128:
    """STROBE v1.0.2, security level 128, Keccak-f[1600]; section 6 of the specification ("Strobe-128/1600")."""
    I, A, C, T, M, K = 1, 2, 4, 8, 16, 32

    def __init__(self, proto):
        self.R = 200 - 128 // 4 - 2                                  # N - sec/4 - 2 = 166
        self.st = bytearray(200)
        self.pos = self.posbegin = 0
        self.I0 = None
        self.cur_flags = None
        # S = F(0x01 || R+2 || 0x01 || 0x00 || 0x01 || 12*8 || "STROBEv1.0.2")  (cSHAKE-style domain separation, section 5)
        dom = bytes([1, self.R + 2, 1, 0, 1, 12 * 8]) + b"STROBEv1.0.2"
        self.st[:len(dom)] = dom
        keccak_f1600(self.st)
        self.operate(self.A | self.M, proto)

    def _run_f(self):
        self.st[self.pos] ^= self.posbegin
        self.st[self.pos + 1] ^= 0x04
        self.st[self.R + 1] ^= 0x80
        keccak_f1600(self.st)
        self.pos = self.posbegin = 0

    def _duplex(self, data, cbefore, cafter, force_f):
        data = bytearray(data)
        for i in range(len(data)):
            if cbefore:
                data[i] ^= self.st[self.pos]
            self.st[self.pos] ^= data[i]
            if cafter:
                data[i] = self.st[self.pos]
            self.pos += 1
            if self.pos == self.R:
                self._run_f()
        if force_f and self.pos != 0:
            self._run_f()
        return bytes(data)

    def _begin_op(self, flags):
        if flags & self.T:                                           # (no transport operations in Merlin; kept for generality)
            if self.I0 is None:
                self.I0 = flags & self.I
            flags ^= self.I0
        old, self.posbegin = self.posbegin, self.pos + 1
        self._duplex(bytes([old, flags]), False, False, bool(flags & (self.C | self.K)))

    def operate(self, flags, data, more=False):
        """data: bytes to absorb, or an int = number of bytes to produce (operations with I set and A set: PRF)"""
        if more:
            assert flags == self.cur_flags
        else:
            self._begin_op(flags)
            self.cur_flags = flags
        if isinstance(data, int):
            data = bytes(data)
        cafter = (flags & (self.C | self.I | self.T)) == (self.C | self.T)
        cbefore = bool(flags & self.C) and not cafter
        processed = self._duplex(data, cbefore, cafter, False)
        if (flags & (self.I | self.A)) == (self.I | self.A):
            return processed                                         # output to the application (PRF, recv_CLR, ...)
        if (flags & (self.I | self.T)) == self.T:
            return processed                                         # output to the transport
        return None


class MerlinTranscript:
    """merlin.cool/transcript/ops.html; the reference's copy is ed25519-dalek/src/batch/transcript.rs:39-207"""

    def __init__(self, label):
        self.s = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, msg):
        S = self.s
        S.operate(S.M | S.A, label)
        S.operate(S.M | S.A, len(msg).to_bytes(4, "little"), more=True)
        S.operate(S.A, msg)

    def challenge_bytes(self, label, n):
        S = self.s
        S.operate(S.M | S.A, label)
        S.operate(S.M | S.A, n.to_bytes(4, "little"), more=True)
        return S.operate(S.I | S.A | S.C, n)

    def rng_finalize(self, random32):
        """build_rng().finalize(rng): meta-AD "rng", then KEY with the 32 bytes the external RNG supplied (transcript.rs:157-173)"""
        S = self.s
        S.operate(S.M | S.A, b"rng")
        S.operate(S.A | S.C, random32)
        return self

    def rng_fill(self, n):
        """TranscriptRng::fill_bytes (transcript.rs:200-206)"""
        S = self.s
        S.operate(S.M | S.A, n.to_bytes(4, "little"))
        return S.operate(S.I | S.A | S.C, n)


def batch_transcript_zs(hrams, ss):
    """The z_i of ed25519_dalek::verify_batch (batch.rs:168-222): hrams = [H(R||A||M)] (64 bytes each), ss = [s] (32 bytes each)
    -> list of 16-byte strings (little-endian u128).  ZeroRng leaves merlin's zero-initialised buffer unchanged (batch.rs:49-76)."""
    t = MerlinTranscript(b"ed25519 batch verification")
    for h in hrams:
        t.append_message(b"hram", h)
    for s in ss:
        t.append_message(b"sig.s", s)
    t.rng_finalize(bytes(32))
    return [t.rng_fill(16) for _ in hrams]


# ---- the device z-mode's derivation (include/c25519_hip.h, C25519_Z_DEVICE), restated from its written description --------------------------
# NOT the reference's derivation: a hash tree over (H(R||A||M) mod l, s) of every signature, whose root is expanded into the z_i.  A node is the first
# 32 bytes of a SHA-512 chaining value after a one-block domain tag (level, inputs of the level, batch size) and fixed-length data with NO padding -- not a
# standard hash call, so the compression function is written out here (FIPS 180-4, 6.4.2).
_SHA512_IV = [0x6a09e667f3bcc908, 0xbb67ae8584caa73b, 0x3c6ef372fe94f82b, 0xa54ff53a5f1d36f1, 0x510e527fade682d1, 0x9b05688c2b3e6c1f, 0x1f83d9abfb41bd6b, 0x5be0cd19137e2179]
_M64 = (1 << 64) - 1


def _sha512_k():
    """the 80 round constants: the first 64 bits of the fractional parts of the cube roots of the first 80 primes"""
    ks, p = [], 2
    while len(ks) < 80:
        if all(p % q for q in range(2, int(p ** 0.5) + 1)):
            lo, hi = 0, 1 << 80                              # floor(p^(1/3) * 2^64) by integer bisection
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if mid ** 3 <= p << 192:
                    lo = mid
                else:
                    hi = mid
            ks.append(lo & _M64)
        p += 1
    return ks


_SHA512_K = _sha512_k()
assert _SHA512_K[0] == 0x428a2f98d728ae22 and _SHA512_K[79] == 0x6c44198c4a475817


def sha512_compress(h, block):
    """one application of the SHA-512 compression function: h = 8 words, block = 128 bytes -> 8 words"""
    rotr = lambda x, r: ((x >> r) | (x << (64 - r))) & _M64
    w = [int.from_bytes(block[8 * i:8 * i + 8], "big") for i in range(16)]
    for t in range(16, 80):
        s0 = rotr(w[t - 15], 1) ^ rotr(w[t - 15], 8) ^ (w[t - 15] >> 7)
        s1 = rotr(w[t - 2], 19) ^ rotr(w[t - 2], 61) ^ (w[t - 2] >> 6)
        w.append((w[t - 16] + s0 + w[t - 7] + s1) & _M64)
    a, b, c, d, e, f, g, hh = h
    for t in range(80):
        t1 = (hh + (rotr(e, 14) ^ rotr(e, 18) ^ rotr(e, 41)) + ((e & f) ^ (~e & _M64 & g)) + _SHA512_K[t] + w[t]) & _M64
        t2 = ((rotr(a, 28) ^ rotr(a, 34) ^ rotr(a, 39)) + ((a & b) ^ (a & c) ^ (b & c))) & _M64
        a, b, c, d, e, f, g, hh = (t1 + t2) & _M64, a, b, c, (d + t1) & _M64, e, f, g
    return [(x + y) & _M64 for x, y in zip(h, [a, b, c, d, e, f, g, hh])]


def device_zs(hrams, ss):
    """hrams = [H(R||A||M)] (64 bytes each), ss = [s] (32 bytes each) -> the n sign-magnitude z_i of the device z-mode (16 bytes each: bit 127 = sign).
    v5 (round 6): every node of the tree is a plain unkeyed BLAKE2b-256 digest (RFC 7693; hashlib), the z_i are quarters of BLAKE2b-512(root || LE64(i / 4)) --
    written from the description in include/c25519_hip.h (C25519_Z_DEVICE), not from the kernels."""
    import hashlib
    n = len(hrams)
    L = 2**252 + 27742317777372353535851937790883648493

    def node(level, count, data):
        tag = b"c25519-hip/verify_batch/z-tree/v5"
        blk = tag + bytes(104 - len(tag)) + level.to_bytes(8, "little") + count.to_bytes(8, "little") + n.to_bytes(8, "little")
        return hashlib.blake2b(blk + data, digest_size=32).digest()

    leaves = [(int.from_bytes(h, "little") % L).to_bytes(32, "little") + s for h, s in zip(hrams, ss)]
    count, level = n, 0
    nodes = [node(0, count, b"".join(leaves[4 * j:4 * j + 4]).ljust(256, b"\0")) for j in range((n + 3) // 4)]      # level 0: four 64-byte records
    while len(nodes) > 1:                                    # upper levels: four children
        count = (count + 3) // 4
        level += 1
        nodes = [node(level, count, b"".join(nodes[4 * j:4 * j + 4]).ljust(128, b"\0")) for j in range((len(nodes) + 3) // 4)]
    out = []
    for i in range((n + 3) // 4):
        d = hashlib.blake2b(nodes[0] + i.to_bytes(8, "little")).digest()
        out += [d[16 * q:16 * q + 16] for q in range(4)]
    return out[:n]
