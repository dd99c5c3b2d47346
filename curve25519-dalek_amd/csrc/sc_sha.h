// Scalar arithmetic mod l and SHA-512 for the verify_batch pipeline (host + device).
//
//  * sc_*: integers mod l = 2^252 + 27742317777372353535851937790883648493 on 5 x 52-bit limbs with
//    Montgomery reduction (R = 2^260) -- the algorithm of the reference's Scalar52
//    (curve25519-dalek/src/backend/serial/u64/scalar.rs:24-345); constants are derived by
//    tools/gen_constants.py.  Needed per signature: one 512-bit reduction of H(R||A||M)
//    (scalar.rs:248), z*s and z*h (scalar.rs:317), and the running sum of z*s (batch.rs:225-233).
//  * sha512_*: FIPS 180-4, one message per lane (call site ed25519-dalek/src/batch.rs:185-189;
//    the reference uses the `sha2` crate).
#pragma once
#include "fe26.h"

namespace c25519 {

typedef unsigned __int128 u128;
struct sc52 { u64 v[5]; };

constexpr u64 SC_MASK52 = (1ull << 52) - 1;
C25519_HD sc52 sc_L() { sc52 r = {{0x0002631a5cf5d3edull, 0x000dea2f79cd6581ull, 0x000000000014def9ull, 0ull, 0x0000100000000000ull}}; return r; }
C25519_HD sc52 sc_R() { sc52 r = {{0x000f48bd6721e6edull, 0x0003bab5ac67e45aull, 0x000fffffeb35e51bull, 0x000fffffffffffffull, 0x00000fffffffffffull}}; return r; }
C25519_HD sc52 sc_RR() { sc52 r = {{0x0009d265e952d13bull, 0x000d63c715bea69full, 0x0005be65cb687604ull, 0x0003dceec73d217full, 0x000009411b7c309aull}}; return r; }
constexpr u64 SC_LFACTOR = 0x51da312547e1bull;

C25519_HD sc52 sc_zero() { sc52 r = {{0, 0, 0, 0, 0}}; return r; }

// 8 little-endian u32 words <-> limbs (scalar.rs:66-85, :121-158)
C25519_HD sc52 sc_from_words(const u32 w[8]) {
    u64 q[4];
    for (int i = 0; i < 4; i++) q[i] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
    sc52 s;
    s.v[0] = q[0] & SC_MASK52;
    s.v[1] = ((q[0] >> 52) | (q[1] << 12)) & SC_MASK52;
    s.v[2] = ((q[1] >> 40) | (q[2] << 24)) & SC_MASK52;
    s.v[3] = ((q[2] >> 28) | (q[3] << 36)) & SC_MASK52;
    s.v[4] = (q[3] >> 16) & ((1ull << 48) - 1);
    return s;
}
C25519_HD void sc_to_words(const sc52 &s, u32 w[8]) {
    u64 q[4];
    q[0] = s.v[0] | (s.v[1] << 52);
    q[1] = (s.v[1] >> 12) | (s.v[2] << 40);
    q[2] = (s.v[2] >> 24) | (s.v[3] << 28);
    q[3] = (s.v[3] >> 36) | (s.v[4] << 16);
    for (int i = 0; i < 4; i++) { w[2 * i] = (u32)q[i]; w[2 * i + 1] = (u32)(q[i] >> 32); }
}

// a - b mod l (scalar.rs:177-207)
C25519_HD sc52 sc_sub(const sc52 &a, const sc52 &b) {
    const sc52 l = sc_L();
    sc52 d;
    u64 borrow = 0;
    for (int i = 0; i < 5; i++) { borrow = a.v[i] - (b.v[i] + (borrow >> 63)); d.v[i] = borrow & SC_MASK52; }
    u64 under = borrow >> 63, carry = 0;
    for (int i = 0; i < 5; i++) { carry = (carry >> 52) + d.v[i] + (under ? l.v[i] : 0ull); d.v[i] = carry & SC_MASK52; }
    return d;
}
// a + b mod l (scalar.rs:161-174)
C25519_HD sc52 sc_add(const sc52 &a, const sc52 &b) {
    sc52 s;
    u64 carry = 0;
    for (int i = 0; i < 5; i++) { carry = a.v[i] + b.v[i] + (carry >> 52); s.v[i] = carry & SC_MASK52; }
    return sc_sub(s, sc_L());
}
C25519_HD sc52 sc_neg(const sc52 &a) { return sc_sub(sc_zero(), a); }

#define SCM(p, q) ((u128)(p) * (u128)(q))
// (scalar.rs:265-299) limbs / 2^260 mod l
C25519_HD sc52 sc_montgomery_reduce(const u128 z[9]) {
    const sc52 L = sc_L();
    const u64 *l = L.v;
    u128 carry, sum;
    u64 n0, n1, n2, n3, n4;
    sc52 r;
    sum = z[0]; n0 = ((u64)sum * SC_LFACTOR) & SC_MASK52; carry = (sum + SCM(n0, l[0])) >> 52;
    sum = carry + z[1] + SCM(n0, l[1]); n1 = ((u64)sum * SC_LFACTOR) & SC_MASK52; carry = (sum + SCM(n1, l[0])) >> 52;
    sum = carry + z[2] + SCM(n0, l[2]) + SCM(n1, l[1]); n2 = ((u64)sum * SC_LFACTOR) & SC_MASK52; carry = (sum + SCM(n2, l[0])) >> 52;
    sum = carry + z[3] + SCM(n1, l[2]) + SCM(n2, l[1]); n3 = ((u64)sum * SC_LFACTOR) & SC_MASK52; carry = (sum + SCM(n3, l[0])) >> 52;
    sum = carry + z[4] + SCM(n0, l[4]) + SCM(n2, l[2]) + SCM(n3, l[1]); n4 = ((u64)sum * SC_LFACTOR) & SC_MASK52; carry = (sum + SCM(n4, l[0])) >> 52;
    sum = carry + z[5] + SCM(n1, l[4]) + SCM(n3, l[2]) + SCM(n4, l[1]); r.v[0] = (u64)sum & SC_MASK52; carry = sum >> 52;
    sum = carry + z[6] + SCM(n2, l[4]) + SCM(n4, l[2]); r.v[1] = (u64)sum & SC_MASK52; carry = sum >> 52;
    sum = carry + z[7] + SCM(n3, l[4]); r.v[2] = (u64)sum & SC_MASK52; carry = sum >> 52;
    sum = carry + z[8] + SCM(n4, l[4]); r.v[3] = (u64)sum & SC_MASK52; carry = sum >> 52;
    r.v[4] = (u64)carry;
    return sc_sub(r, L);
}
C25519_HD void sc_mul_internal(u128 z[9], const sc52 &x, const sc52 &y) {
    const u64 *a = x.v, *b = y.v;
    z[0] = SCM(a[0], b[0]);
    z[1] = SCM(a[0], b[1]) + SCM(a[1], b[0]);
    z[2] = SCM(a[0], b[2]) + SCM(a[1], b[1]) + SCM(a[2], b[0]);
    z[3] = SCM(a[0], b[3]) + SCM(a[1], b[2]) + SCM(a[2], b[1]) + SCM(a[3], b[0]);
    z[4] = SCM(a[0], b[4]) + SCM(a[1], b[3]) + SCM(a[2], b[2]) + SCM(a[3], b[1]) + SCM(a[4], b[0]);
    z[5] = SCM(a[1], b[4]) + SCM(a[2], b[3]) + SCM(a[3], b[2]) + SCM(a[4], b[1]);
    z[6] = SCM(a[2], b[4]) + SCM(a[3], b[3]) + SCM(a[4], b[2]);
    z[7] = SCM(a[3], b[4]) + SCM(a[4], b[3]);
    z[8] = SCM(a[4], b[4]);
}
#undef SCM
C25519_HD sc52 sc_montgomery_mul(const sc52 &a, const sc52 &b) { u128 z[9]; sc_mul_internal(z, a, b); return sc_montgomery_reduce(z); }
// a * b mod l (scalar.rs:302-306)
C25519_HD sc52 sc_mul(const sc52 &a, const sc52 &b) { return sc_montgomery_mul(sc_montgomery_mul(a, b), sc_RR()); }
// 256-bit value -> mod l (Scalar::reduce, scalar.rs:1159-1164)
C25519_HD sc52 sc_reduce256(const u32 w[8]) { return sc_montgomery_mul(sc_from_words(w), sc_R()); }
// from_canonical_bytes check (scalar.rs:259-263)
C25519_HD bool sc_is_canonical(const u32 w[8]) {
    if (w[7] >> 31) return false;
    u32 c[8];
    sc_to_words(sc_reduce256(w), c);
    u32 d = 0;
    for (int i = 0; i < 8; i++) d |= c[i] ^ w[i];
    return d == 0;
}
// 512-bit little-endian (16 words) -> mod l (scalar.rs:89-118)
C25519_HD sc52 sc_from_wide(const u32 w[16]) {
    u64 q[8];
    for (int i = 0; i < 8; i++) q[i] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
    sc52 lo, hi;
    lo.v[0] = q[0] & SC_MASK52;
    lo.v[1] = ((q[0] >> 52) | (q[1] << 12)) & SC_MASK52;
    lo.v[2] = ((q[1] >> 40) | (q[2] << 24)) & SC_MASK52;
    lo.v[3] = ((q[2] >> 28) | (q[3] << 36)) & SC_MASK52;
    lo.v[4] = ((q[3] >> 16) | (q[4] << 48)) & SC_MASK52;
    hi.v[0] = (q[4] >> 4) & SC_MASK52;
    hi.v[1] = ((q[4] >> 56) | (q[5] << 8)) & SC_MASK52;
    hi.v[2] = ((q[5] >> 44) | (q[6] << 20)) & SC_MASK52;
    hi.v[3] = ((q[6] >> 32) | (q[7] << 32)) & SC_MASK52;
    hi.v[4] = q[7] >> 20;
    lo = sc_montgomery_mul(lo, sc_R());
    hi = sc_montgomery_mul(hi, sc_RR());
    return sc_add(hi, lo);
}

// ---- SHA-512 ---------------------------------------------------------------------------------------
#ifdef __HIP_DEVICE_COMPILE__
__device__ __constant__ static const u64 SHA512_K[80] = C25519_SHA512_K;
#else
static const u64 SHA512_K[80] = C25519_SHA512_K;
#endif
C25519_HD u64 rotr64(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
C25519_HD void sha512_init(u64 h[8]) { const u64 iv[8] = C25519_SHA512_IV; for (int i = 0; i < 8; i++) h[i] = iv[i]; }
// one compression; w[16] = the block as big-endian u64 words (destroyed)
C25519_HD void sha512_compress(u64 h[8], u64 w[16]) {
    u64 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll 1
    for (int i = 0; i < 80; i += 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (i > 0) {
                u64 w15 = w[(j + 1) & 15], w2 = w[(j + 14) & 15];
                u64 s0 = rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7);
                u64 s1 = rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6);
                w[j] = w[j] + s0 + w[(j + 9) & 15] + s1;
            }
            u64 S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41);
            u64 ch = (e & f) ^ (~e & g);
            u64 t1 = hh + S1 + ch + SHA512_K[i + j] + w[j];
            u64 S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39);
            u64 maj = (a & b) ^ (a & c) ^ (b & c);
            u64 t2 = S0 + maj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
C25519_HD u64 bswap64(u64 x) {
    x = ((x & 0x00ff00ff00ff00ffull) << 8) | ((x >> 8) & 0x00ff00ff00ff00ffull);
    x = ((x & 0x0000ffff0000ffffull) << 16) | ((x >> 16) & 0x0000ffff0000ffffull);
    return (x << 32) | (x >> 32);
}
// digest as 16 little-endian u32 words = the byte string interpreted as a 512-bit LE integer
C25519_HD void sha512_digest_words(const u64 h[8], u32 w[16]) {
    for (int i = 0; i < 8; i++) { u64 le = bswap64(h[i]); w[2 * i] = (u32)le; w[2 * i + 1] = (u32)(le >> 32); }
}

// Streaming absorber over a byte-granular source, one message per lane: `prefix` words first
// (npre bytes, multiple of 8), then `len` message bytes from `msg`.
struct sha512_stream {
    u64 h[8], w[16];
    u32 fill;   // bytes in the current block
    u64 total;
    C25519_HD void init() { sha512_init(h); for (int i = 0; i < 16; i++) w[i] = 0; fill = 0; total = 0; }
    C25519_HD void put_be64(u64 v) {   // absorb 8 bytes given as a big-endian word; fill must be 8-aligned
        // dynamic index into w[]: written as a select chain so w stays in registers
        u32 slot = fill >> 3;
#pragma unroll
        for (int i = 0; i < 16; i++) if ((u32)i == slot) w[i] = v;
        fill += 8; total += 8;
        if (fill == 128) { sha512_compress(h, w); fill = 0; }
    }
    C25519_HD void put_byte(u32 b) {
        u32 slot = fill >> 3, sh = 56 - 8 * (fill & 7);
#pragma unroll
        for (int i = 0; i < 16; i++) if ((u32)i == slot) w[i] = ((fill & 7) == 0 ? 0ull : w[i]) | ((u64)b << sh);
        fill += 1; total += 1;
        if (fill == 128) { sha512_compress(h, w); fill = 0; }
    }
    // len message bytes: whole big-endian words wherever the block position is 8-aligned (one 16-way select per 8 bytes instead of
    // per byte: the select chain is what keeps w[] in registers, and it is 40 VALU instructions each time -- byte-wise absorption of a
    // 32-byte message and its padding was ~2000 instructions beside a ~5000-instruction compression)
    C25519_HD void put_bytes(const uint8_t *m, u64 len) {
        while (len && (fill & 7)) { put_byte(*m++); len--; }
        const bool aligned = (((uintptr_t)m) & 3) == 0;
        while (len >= 8) {
            u64 v;
            if (aligned) {
                const u32 a = reinterpret_cast<const u32 *>(m)[0], b = reinterpret_cast<const u32 *>(m)[1];      // little-endian loads
                v = bswap64((u64)a | ((u64)b << 32));
            } else {
                v = 0;
                for (int j = 0; j < 8; j++) v = (v << 8) | m[j];
            }
            put_be64(v);
            m += 8; len -= 8;
        }
        while (len) { put_byte(*m++); len--; }
    }
    C25519_HD void finish() {
        u64 bits = total * 8;
        put_byte(0x80);                                   // (the rest of this word is already zero: a word is cleared when its first byte arrives)
        fill = (fill + 7u) & ~7u;
        if (fill == 128) { sha512_compress(h, w); fill = 0; }
        while (fill != 112) put_be64(0);
        put_be64(0); put_be64(bits);
    }
};

}  // namespace c25519
