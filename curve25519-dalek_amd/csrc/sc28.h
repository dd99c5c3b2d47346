// Scalar arithmetic mod l = 2^252 + c on the DEVICE, designed for a 32 x 32 -> 64 bit multiplier (v_mad_u64_u32).
//
// What the reference does (backend/serial/u64/scalar.rs:24-345, restated on 5 x 52-bit limbs in sc_sha.h for the host): every
// product is a Montgomery multiplication followed by a second one with RR -- two 5 x 5 limb products and two Montgomery
// reductions on 64 x 64 -> 128 bit multipliers.  A GPU has no such multiplier: each of those limb products is four
// v_mad_u64_u32 plus glue, ~400 multiplier instructions per sc_mul, all in one dependent chain.  Here instead:
//
//   * radix 2^28, 10 limbs (values below 2^280).  Limb products are < 2^56, a column of up to ten of them is < 2^60: every
//     partial product is ONE v_mad_u64_u32 into a 64-bit column sum, no carry tracking inside a column.
//   * 2^252 = 2^(9 * 28): the split of a value at bit 252 is a limb boundary, so the special form of the group order
//     l = 2^252 + c (c = 27742317777372353535851937790883648493 < 2^125, five limbs) gives the reduction directly:
//         x = hi * 2^252 + lo  =  lo - c * hi   (mod l).
//     One fold costs 5 * (limbs of hi) multiplier instructions and shrinks x by ~127 bits; to stay in unsigned limbs a multiple
//     2^(28 sh) * l >= c * hi is added first.  A 512-bit hash takes four folds (50 + 30 + 10 + 5 products), a 10 x 5 or 10 x 10
//     limb product three or four -- 95 to 195 multiplier instructions per multiplication-and-reduction, no Montgomery form, no
//     conversion constants, and the columns of a product are independent chains.
//   * from_canonical_bytes (scalar.rs:259-263) is a word-wise comparison with l, not a reduction.
//
// Used by the verify_batch / sign / per-signature-verify kernels (msm.hip, single.hip).  sc_sha.h's 5 x 52 form stays for the
// host (table generation, the Scalar::invert_batch chain).  Host + device (tests/test_fe26_host.py fuzzes it against big integers).
#pragma once
#include "fe26.h"

namespace c25519 {

constexpr u32 SC28_MASK = (1u << 28) - 1u;
struct sc28 { u32 v[10]; };                 // canonical: the integer is < l (so v[9] <= 1)

// c = l - 2^252 as five 28-bit limbs; l as eight 32-bit words
C25519_HD void sc28_c(u32 c[5]) { c[0] = 0xcf5d3edu; c[1] = 0x12631a5u; c[2] = 0x79cd658u; c[3] = 0xf9dea2fu; c[4] = 0x14deu; }
C25519_HD void sc28_l_words(u32 w[8]) {
    w[0] = 0x5cf5d3edu; w[1] = 0x5812631au; w[2] = 0xa2f79cd6u; w[3] = 0x14def9deu; w[4] = 0; w[5] = 0; w[6] = 0; w[7] = 0x10000000u;
}

// NW little-endian 32-bit words -> NL 28-bit limbs (NL * 28 >= NW * 32)
template <int NW, int NL>
C25519_HD void sc28_limbs_from_words(const u32 *w, u32 *out) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = 28 * i, wi = bit >> 5, sh = bit & 31;
        u64 two = wi < NW ? (u64)w[wi] : 0ull;
        if (wi + 1 < NW) two |= (u64)w[wi + 1] << 32;
        out[i] = (u32)(two >> sh) & SC28_MASK;
    }
}
// ten 28-bit limbs of a value < 2^256 -> eight words
C25519_HD void sc28_to_words(const sc28 &a, u32 w[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        // word i = bits [32 i, 32 i + 32): limb j0 = floor(32 i / 28) from bit (32 i - 28 j0), and the next one or two limbs
        const int bit = 32 * i, j0 = bit / 28, sh = bit - 28 * j0;
        u64 v = (u64)a.v[j0] >> sh;
        if (j0 + 1 < 10) v |= (u64)a.v[j0 + 1] << (28 - sh);
        if (j0 + 2 < 10) v |= (u64)a.v[j0 + 2] << (56 - sh);
        w[i] = (u32)v;
    }
}
C25519_HD sc28 sc28_from_words(const u32 w[8]) { sc28 r; sc28_limbs_from_words<8, 10>(w, r.v); return r; }
C25519_HD sc28 sc28_zero() { sc28 r; for (int i = 0; i < 10; i++) r.v[i] = 0; return r; }

// s < l, word-wise (scalar.rs:259-263 from_canonical_bytes: bit 255 clear and s == s mod l)
C25519_HD bool sc28_words_canonical(const u32 w[8]) {
    u32 l[8];
    sc28_l_words(l);
    u32 borrow = 0;                          // s - l borrows out  <=>  s < l
#pragma unroll
    for (int i = 0; i < 8; i++) { const u64 d = (u64)w[i] - l[i] - borrow; borrow = (u32)(d >> 63); }
    return borrow != 0;
}

// One fold.  x: 9 + NH limbs (28-bit, unsigned); out (SH + 10 limbs) = lo + 2^(28 SH) * l - c * hi, which must be >= 0:
// the caller picks SH with 2^(28 SH) * l >= c * hi for every admissible x.
template <int NH, int SH>
C25519_HD void sc28_fold(const u32 *x, u32 *out) {
    constexpr int NO = SH + 10;
    u32 c[5];
    sc28_c(c);
    u64 T[NH + 4];                           // columns of c * hi
#pragma unroll
    for (int k = 0; k < NH + 4; k++) T[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < NH; j++) T[i + j] += (u64)c[i] * x[9 + j];
    long long carry = 0;
#pragma unroll
    for (int k = 0; k < NO; k++) {
        long long acc = carry;
        if (k < 9) acc += (long long)x[k];
        if (k >= SH && k < SH + 5) acc += (long long)c[k - SH];
        if (k == SH + 9) acc += 1;
        if (k < NH + 4) acc -= (long long)T[k];
        out[k] = (u32)((u64)acc & SC28_MASK);
        carry = acc >> 28;                   // arithmetic shift: the borrow travels up
    }
}
// the last step: x < 2^254 in ten limbs -> [0, l)
C25519_HD sc28 sc28_finish(const u32 x[10]) {
    u32 c[5];
    sc28_c(c);
    const u32 hi = x[9];                     // <= 3
    u32 t[10];
    long long carry = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        long long acc = carry + (long long)x[k];
        if (k < 5) acc -= (long long)((u64)c[k] * hi);
        t[k] = (u32)((u64)acc & SC28_MASK);
        carry = acc >> 28;
    }
    const bool neg = carry < 0;              // lo - c * hi in (-2^127, 2^252): negative -> add l once
    const lanemask nm = lane_mask(neg);      // (explicit lane-mask selects: fe26.h sel_u32)
    sc28 r;
    u32 cy = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const u32 add = k < 5 ? sel_u32(0u, c[k], nm) : 0u;
        const u32 s = t[k] + add + cy;
        r.v[k] = s & SC28_MASK;
        cy = s >> 28;
    }
    // (neg: the nine limbs t hold value + 2^252 -- the borrow left at limb 9 -- so value + l = t + c, with a possible carry into limb 9)
    r.v[9] = cy;
    return r;
}
// 512-bit little-endian value (16 words) mod l: Scalar::from_bytes_mod_order_wide (scalar.rs:248, u64/scalar.rs:89-118)
C25519_HD sc28 sc28_from_wide(const u32 w[16]) {
    u32 x[19], a[15], b[11], d[10];
    sc28_limbs_from_words<16, 19>(w, x);     // hi < 2^260
    sc28_fold<10, 5>(x, a);                  // c hi < 2^385 <= 2^140 l;  a < 2^252 + 2^140 l < 2^393
    sc28_fold<6, 1>(a, b);                   // hi < 2^141, c hi < 2^266 <= 2^28 l;  b < 2^252 + 2^28 l < 2^281
    sc28_fold<2, 0>(b, d);                   // hi < 2^29,  c hi < 2^154 <= l;       d < 2^252 + l < 2^254
    return sc28_finish(d);
}
// a * b mod l for a < 2^140 (five limbs: a 128-bit z_i) and b < 2^280
C25519_HD sc28 sc28_mul_5x10(const u32 a[5], const u32 b[10]) {
    u64 col[14];
#pragma unroll
    for (int k = 0; k < 14; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 10; j++) col[i + j] += (u64)a[i] * b[j];
    u32 x[15], t[11], d[10];
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) { const u64 v = col[k] + carry; x[k] = (u32)v & SC28_MASK; carry = v >> 28; }
    x[14] = (u32)carry;                      // the product is < 2^(128 + 253) = 2^381 for the callers: hi < 2^141
    sc28_fold<6, 1>(x, t);
    sc28_fold<2, 0>(t, d);
    return sc28_finish(d);
}
// a * b mod l for a, b < 2^280 with a * b < 2^512 (RFC 8032 signing: k * a with a clamped, unreduced; S = r + k a)
C25519_HD sc28 sc28_mul(const u32 a[10], const u32 b[10]) {
    u64 col[19];
#pragma unroll
    for (int k = 0; k < 19; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++)
#pragma unroll
        for (int j = 0; j < 10; j++) col[i + j] += (u64)a[i] * b[j];
    u32 x[19], p[15], t[11], d[10];
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < 19; k++) { const u64 v = col[k] + carry; x[k] = (u32)v & SC28_MASK; carry = v >> 28; }
    // (callers guarantee a * b < 2^512, so nothing is carried out of limb 18 beyond its 8 bits: hi < 2^260)
    sc28_fold<10, 5>(x, p);
    sc28_fold<6, 1>(p, t);
    sc28_fold<2, 0>(t, d);
    return sc28_finish(d);
}
// (a + b) mod l and (l - a) mod l for canonical operands (scalar.rs:161-207)
C25519_HD sc28 sc28_add(const sc28 &a, const sc28 &b) {
    u32 s[10], cy = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) { const u32 v = a.v[k] + b.v[k] + cy; s[k] = v & SC28_MASK; cy = v >> 28; }
    // a + b < 2 l < 2^254: subtract l if a + b >= l
    u32 c[5];
    sc28_c(c);
    u32 d[10];
    long long carry = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        long long acc = carry + (long long)s[k] - (long long)(k < 5 ? c[k] : 0u) - (k == 9 ? 1 : 0);
        d[k] = (u32)((u64)acc & SC28_MASK);
        carry = acc >> 28;
    }
    const bool keep = carry < 0;             // a + b < l
    sc28 r;
    const lanemask km = lane_mask(keep);
#pragma unroll
    for (int k = 0; k < 10; k++) r.v[k] = sel_u32(d[k], s[k], km);
    return r;
}
C25519_HD sc28 sc28_neg(const sc28 &a) {
    u32 c[5];
    sc28_c(c);
    u32 any = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) any |= a.v[k];
    sc28 r;
    const lanemask am = lane_mask(any != 0);
    long long carry = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        long long acc = carry + (long long)(k < 5 ? c[k] : 0u) + (k == 9 ? 1 : 0) - (long long)a.v[k];
        r.v[k] = sel_u32(0u, (u32)((u64)acc & SC28_MASK), am);
        carry = acc >> 28;
    }
    return r;
}

}  // namespace c25519
