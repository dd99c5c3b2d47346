/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's scalar arithmetic mod l:
 *   curve25519-dalek/src/backend/serial/u64/scalar.rs   (Scalar52: 5 x 52-bit limbs, Montgomery)
 *   curve25519-dalek/src/scalar.rs                      (Scalar wrapper, digit recodings)
 */
#ifndef ORC_SC52_H
#define ORC_SC52_H
#include <stdint.h>
#include <string.h>
#include "constants.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t v[5]; } sc52;
#define ORC_MASK52 ((((uint64_t)1) << 52) - 1)

static const sc52 SC_ZERO = {{0, 0, 0, 0, 0}};
static inline sc52 sc_const(const uint64_t l[5]) { sc52 r; memcpy(r.v, l, 40); return r; }

/* u64/scalar.rs:66-85 */
static inline sc52 sc_from_bytes(const uint8_t b[32]) {
    uint64_t w[4]; memcpy(w, b, 32);
    sc52 s;
    s.v[0] = w[0] & ORC_MASK52;
    s.v[1] = ((w[0] >> 52) | (w[1] << 12)) & ORC_MASK52;
    s.v[2] = ((w[1] >> 40) | (w[2] << 24)) & ORC_MASK52;
    s.v[3] = ((w[2] >> 28) | (w[3] << 36)) & ORC_MASK52;
    s.v[4] = (w[3] >> 16) & ((((uint64_t)1) << 48) - 1);
    return s;
}

/* u64/scalar.rs:121-158 */
static inline void sc_to_bytes(uint8_t out[32], sc52 s) {
    uint64_t w[4];
    w[0] = s.v[0] | (s.v[1] << 52);
    w[1] = (s.v[1] >> 12) | (s.v[2] << 40);
    w[2] = (s.v[2] >> 24) | (s.v[3] << 28);
    w[3] = (s.v[3] >> 36) | (s.v[4] << 16);
    memcpy(out, w, 32);
}

/* u64/scalar.rs:177-207 (sub + conditional_add_l) */
static inline sc52 sc_sub(sc52 a, sc52 b) {
    sc52 d; uint64_t borrow = 0;
    for (int i = 0; i < 5; i++) {
        borrow = a.v[i] - (b.v[i] + (borrow >> 63));
        d.v[i] = borrow & ORC_MASK52;
    }
    uint64_t under = borrow >> 63, carry = 0;
    for (int i = 0; i < 5; i++) {
        carry = (carry >> 52) + d.v[i] + (under ? ORC_SC_L[i] : 0);
        d.v[i] = carry & ORC_MASK52;
    }
    return d;
}

/* u64/scalar.rs:161-174 */
static inline sc52 sc_add(sc52 a, sc52 b) {
    sc52 s; uint64_t carry = 0;
    for (int i = 0; i < 5; i++) {
        carry = a.v[i] + b.v[i] + (carry >> 52);
        s.v[i] = carry & ORC_MASK52;
    }
    return sc_sub(s, sc_const(ORC_SC_L));
}

/* u64/scalar.rs:222-237 */
static inline void sc_mul_internal(u128 z[9], sc52 x, sc52 y) {
    const uint64_t *a = x.v, *b = y.v;
#define M(p, q) ((u128)(p) * (u128)(q))
    z[0] = M(a[0], b[0]);
    z[1] = M(a[0], b[1]) + M(a[1], b[0]);
    z[2] = M(a[0], b[2]) + M(a[1], b[1]) + M(a[2], b[0]);
    z[3] = M(a[0], b[3]) + M(a[1], b[2]) + M(a[2], b[1]) + M(a[3], b[0]);
    z[4] = M(a[0], b[4]) + M(a[1], b[3]) + M(a[2], b[2]) + M(a[3], b[1]) + M(a[4], b[0]);
    z[5] = M(a[1], b[4]) + M(a[2], b[3]) + M(a[3], b[2]) + M(a[4], b[1]);
    z[6] = M(a[2], b[4]) + M(a[3], b[3]) + M(a[4], b[2]);
    z[7] = M(a[3], b[4]) + M(a[4], b[3]);
    z[8] = M(a[4], b[4]);
}

/* u64/scalar.rs:265-299 -- limbs / 2^260 mod l */
static inline sc52 sc_montgomery_reduce(const u128 limbs[9]) {
    const uint64_t *l = ORC_SC_L;
    u128 carry, sum; uint64_t n0, n1, n2, n3, n4, r0, r1, r2, r3, r4;
#define PART1(S, N) do { sum = (S); N = ((uint64_t)sum * ORC_SC_LFACTOR) & ORC_MASK52; carry = (sum + M(N, l[0])) >> 52; } while (0)
#define PART2(S, W) do { sum = (S); W = (uint64_t)sum & ORC_MASK52; carry = sum >> 52; } while (0)
    PART1(limbs[0], n0);
    PART1(carry + limbs[1] + M(n0, l[1]), n1);
    PART1(carry + limbs[2] + M(n0, l[2]) + M(n1, l[1]), n2);
    PART1(carry + limbs[3] + M(n1, l[2]) + M(n2, l[1]), n3);
    PART1(carry + limbs[4] + M(n0, l[4]) + M(n2, l[2]) + M(n3, l[1]), n4);
    PART2(carry + limbs[5] + M(n1, l[4]) + M(n3, l[2]) + M(n4, l[1]), r0);
    PART2(carry + limbs[6] + M(n2, l[4]) + M(n4, l[2]), r1);
    PART2(carry + limbs[7] + M(n3, l[4]), r2);
    PART2(carry + limbs[8] + M(n4, l[4]), r3);
    r4 = (uint64_t)carry;
#undef PART1
#undef PART2
#undef M
    sc52 r = {{r0, r1, r2, r3, r4}};
    return sc_sub(r, sc_const(ORC_SC_L));
}

/* u64/scalar.rs:317-320, :302-306 */
static inline sc52 sc_montgomery_mul(sc52 a, sc52 b) { u128 z[9]; sc_mul_internal(z, a, b); return sc_montgomery_reduce(z); }
static inline sc52 sc_mul(sc52 a, sc52 b) {
    sc52 ab = sc_montgomery_mul(a, b);
    return sc_montgomery_mul(ab, sc_const(ORC_SC_RR));
}

/* u64/scalar.rs:89-118 -- 512-bit -> mod l */
static inline sc52 sc_from_bytes_wide(const uint8_t b[64]) {
    uint64_t w[8]; memcpy(w, b, 64);
    sc52 lo, hi;
    lo.v[0] = w[0] & ORC_MASK52;
    lo.v[1] = ((w[0] >> 52) | (w[1] << 12)) & ORC_MASK52;
    lo.v[2] = ((w[1] >> 40) | (w[2] << 24)) & ORC_MASK52;
    lo.v[3] = ((w[2] >> 28) | (w[3] << 36)) & ORC_MASK52;
    lo.v[4] = ((w[3] >> 16) | (w[4] << 48)) & ORC_MASK52;
    hi.v[0] = (w[4] >> 4) & ORC_MASK52;
    hi.v[1] = ((w[4] >> 56) | (w[5] << 8)) & ORC_MASK52;
    hi.v[2] = ((w[5] >> 44) | (w[6] << 20)) & ORC_MASK52;
    hi.v[3] = ((w[6] >> 32) | (w[7] << 32)) & ORC_MASK52;
    hi.v[4] = w[7] >> 20;
    lo = sc_montgomery_mul(lo, sc_const(ORC_SC_R));
    hi = sc_montgomery_mul(hi, sc_const(ORC_SC_RR));
    return sc_add(hi, lo);
}

/* scalar.rs:235-246 (from_bytes_mod_order = reduce: montgomery_mul by R then from_montgomery,
   scalar.rs:1160-1166) */
static inline sc52 sc_from_bytes_mod_order(const uint8_t b[32]) {
    sc52 x = sc_from_bytes(b);
    u128 z[9]; sc_mul_internal(z, x, sc_const(ORC_SC_R));
    return sc_montgomery_reduce(z);
}

/* scalar.rs:259-263 -- high bit clear and s == s mod l */
static inline int sc_is_canonical_bytes(const uint8_t b[32]) {
    if (b[31] >> 7) return 0;
    uint8_t c[32]; sc_to_bytes(c, sc_from_bytes_mod_order(b));
    return memcmp(b, c, 32) == 0;
}

/* scalar.rs:955-1007 -- width-w NAF, input must have bit 255 clear */
static inline void sc_non_adjacent_form(int8_t naf[256], const uint8_t s[32], unsigned w) {
    memset(naf, 0, 256);
    uint64_t x[5] = {0, 0, 0, 0, 0};
    memcpy(x, s, 32);
    uint64_t width = 1ULL << w, window_mask = width - 1;
    unsigned pos = 0; uint64_t carry = 0;
    while (pos < 256) {
        unsigned idx = pos / 64, bit = pos % 64;
        uint64_t buf = (bit < 64 - w) ? (x[idx] >> bit) : ((x[idx] >> bit) | (x[idx + 1] << (64 - bit)));
        uint64_t window = carry + (buf & window_mask);
        if ((window & 1) == 0) { pos += 1; continue; }
        if (window < width / 2) { carry = 0; naf[pos] = (int8_t)window; }
        else { carry = 1; naf[pos] = (int8_t)((int8_t)window - (int8_t)width); }
        pos += w;
    }
}

/* scalar.rs:1019-1051 -- radix 16, digits in [-8,8), top digit <= 8 */
static inline void sc_as_radix_16(int8_t out[64], const uint8_t s[32]) {
    for (int i = 0; i < 32; i++) { out[2 * i] = s[i] & 15; out[2 * i + 1] = (s[i] >> 4) & 15; }
    for (int i = 0; i < 63; i++) {
        int8_t carry = (int8_t)((out[i] + 8) >> 4);
        out[i] -= (int8_t)(carry << 4);
        out[i + 1] += carry;
    }
}

/* scalar.rs:1056-1071 */
static inline unsigned sc_radix_2w_size_hint(unsigned w) { return w == 8 ? 33 : (256 + w - 1) / w; }

/* scalar.rs:1093-1150 -- signed radix 2^w, w in 4..8 */
static inline void sc_as_radix_2w(int8_t digits[64], const uint8_t s[32], unsigned w) {
    if (w == 4) { sc_as_radix_16(digits, s); return; }
    memset(digits, 0, 64);
    uint64_t x[4]; memcpy(x, s, 32);
    uint64_t radix = 1ULL << w, mask = radix - 1, carry = 0;
    unsigned count = (256 + w - 1) / w;
    for (unsigned i = 0; i < count; i++) {
        unsigned off = i * w, idx = off / 64, bit = off % 64;
        uint64_t buf = (bit < 64 - w || idx == 3) ? (x[idx] >> bit) : ((x[idx] >> bit) | (x[idx + 1] << (64 - bit)));
        uint64_t coef = carry + (buf & mask);
        carry = (coef + radix / 2) >> w;
        digits[i] = (int8_t)((int64_t)coef - (int64_t)(carry << w));
    }
    if (w == 8) digits[count] += (int8_t)carry;
    else digits[count - 1] += (int8_t)(carry << w);
}

/* scalar.rs:1407-1412 */
static inline void sc_clamp_integer(uint8_t b[32]) { b[0] &= 248; b[31] &= 127; b[31] |= 64; }

#endif
