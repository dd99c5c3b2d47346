/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into or called by the product path.
 *
 * CPU restatement (plain C, unsigned __int128) of the reference's serial u64 field backend:
 *   curve25519-dalek/src/backend/serial/u64/field.rs   (FieldElement51: 5 x u64, radix 2^51)
 *   curve25519-dalek/src/field.rs                      (invert, pow_p58, sqrt_ratio_i, batch invert)
 * Each function cites the reference lines whose behaviour it follows.
 */
#ifndef ORC_FE51_H
#define ORC_FE51_H
#include <stdint.h>
#include <string.h>
#include "constants.h"

typedef unsigned __int128 u128;
typedef struct { uint64_t v[5]; } fe;

#define ORC_MASK51 ((((uint64_t)1) << 51) - 1)

static const fe FE_ZERO = {{0, 0, 0, 0, 0}};
static const fe FE_ONE = {{1, 0, 0, 0, 0}};

static inline fe fe_from_limbs(const uint64_t l[5]) { fe r; memcpy(r.v, l, 40); return r; }

/* u64/field.rs:290-323 -- weak reduction, carries computed in parallel */
static inline fe fe_reduce(fe a) {
    uint64_t c0 = a.v[0] >> 51, c1 = a.v[1] >> 51, c2 = a.v[2] >> 51, c3 = a.v[3] >> 51, c4 = a.v[4] >> 51;
    a.v[0] &= ORC_MASK51; a.v[1] &= ORC_MASK51; a.v[2] &= ORC_MASK51; a.v[3] &= ORC_MASK51; a.v[4] &= ORC_MASK51;
    a.v[0] += c4 * 19; a.v[1] += c0; a.v[2] += c1; a.v[3] += c2; a.v[4] += c3;
    return a;
}

/* u64/field.rs:58-73 -- limb-wise, never reduces */
static inline fe fe_add(fe a, fe b) {
    for (int i = 0; i < 5; i++) a.v[i] += b.v[i];
    return a;
}

/* u64/field.rs:82-102 -- add 16p limb-wise, then weak-reduce */
static inline fe fe_sub(fe a, fe b) {
    fe r;
    r.v[0] = (a.v[0] + 36028797018963664ULL) - b.v[0];
    r.v[1] = (a.v[1] + 36028797018963952ULL) - b.v[1];
    r.v[2] = (a.v[2] + 36028797018963952ULL) - b.v[2];
    r.v[3] = (a.v[3] + 36028797018963952ULL) - b.v[3];
    r.v[4] = (a.v[4] + 36028797018963952ULL) - b.v[4];
    return fe_reduce(r);
}

/* u64/field.rs:276-287 */
static inline fe fe_neg(fe a) { return fe_sub(FE_ZERO, a); }

/* u64/field.rs:111-214 -- 5x5 schoolbook with the x19 fold and the serial carry chain */
static inline fe fe_mul(fe x, fe y) {
    const uint64_t *a = x.v, *b = y.v;
    uint64_t b1_19 = b[1] * 19, b2_19 = b[2] * 19, b3_19 = b[3] * 19, b4_19 = b[4] * 19;
#define M(p, q) ((u128)(p) * (u128)(q))
    u128 c0 = M(a[0], b[0]) + M(a[4], b1_19) + M(a[3], b2_19) + M(a[2], b3_19) + M(a[1], b4_19);
    u128 c1 = M(a[1], b[0]) + M(a[0], b[1]) + M(a[4], b2_19) + M(a[3], b3_19) + M(a[2], b4_19);
    u128 c2 = M(a[2], b[0]) + M(a[1], b[1]) + M(a[0], b[2]) + M(a[4], b3_19) + M(a[3], b4_19);
    u128 c3 = M(a[3], b[0]) + M(a[2], b[1]) + M(a[1], b[2]) + M(a[0], b[3]) + M(a[4], b4_19);
    u128 c4 = M(a[4], b[0]) + M(a[3], b[1]) + M(a[2], b[2]) + M(a[1], b[3]) + M(a[0], b[4]);
    fe out;
    c1 += (uint64_t)(c0 >> 51); out.v[0] = (uint64_t)c0 & ORC_MASK51;
    c2 += (uint64_t)(c1 >> 51); out.v[1] = (uint64_t)c1 & ORC_MASK51;
    c3 += (uint64_t)(c2 >> 51); out.v[2] = (uint64_t)c2 & ORC_MASK51;
    c4 += (uint64_t)(c3 >> 51); out.v[3] = (uint64_t)c3 & ORC_MASK51;
    uint64_t carry = (uint64_t)(c4 >> 51); out.v[4] = (uint64_t)c4 & ORC_MASK51;
    out.v[0] += carry * 19;
    out.v[1] += out.v[0] >> 51; out.v[0] &= ORC_MASK51;
    return out;
}

/* u64/field.rs:454-559 -- k squarings, 15 products each */
static inline fe fe_pow2k(fe x, unsigned k) {
    uint64_t a[5];
    memcpy(a, x.v, 40);
    do {
        uint64_t a3_19 = 19 * a[3], a4_19 = 19 * a[4];
        u128 c0 = M(a[0], a[0]) + 2 * (M(a[1], a4_19) + M(a[2], a3_19));
        u128 c1 = M(a[3], a3_19) + 2 * (M(a[0], a[1]) + M(a[2], a4_19));
        u128 c2 = M(a[1], a[1]) + 2 * (M(a[0], a[2]) + M(a[4], a3_19));
        u128 c3 = M(a[4], a4_19) + 2 * (M(a[0], a[3]) + M(a[1], a[2]));
        u128 c4 = M(a[2], a[2]) + 2 * (M(a[0], a[4]) + M(a[1], a[3]));
        c1 += (uint64_t)(c0 >> 51); a[0] = (uint64_t)c0 & ORC_MASK51;
        c2 += (uint64_t)(c1 >> 51); a[1] = (uint64_t)c1 & ORC_MASK51;
        c3 += (uint64_t)(c2 >> 51); a[2] = (uint64_t)c2 & ORC_MASK51;
        c4 += (uint64_t)(c3 >> 51); a[3] = (uint64_t)c3 & ORC_MASK51;
        uint64_t carry = (uint64_t)(c4 >> 51); a[4] = (uint64_t)c4 & ORC_MASK51;
        a[0] += carry * 19;
        a[1] += a[0] >> 51; a[0] &= ORC_MASK51;
    } while (--k);
    fe r; memcpy(r.v, a, 40); return r;
}
#undef M

/* u64/field.rs:562-574 */
static inline fe fe_sq(fe a) { return fe_pow2k(a, 1); }
static inline fe fe_sq2(fe a) {
    fe s = fe_pow2k(a, 1);
    for (int i = 0; i < 5; i++) s.v[i] *= 2;
    return s;
}

/* u64/field.rs:338-363 -- bit 255 is masked away, values >= p are accepted */
static inline fe fe_from_bytes(const uint8_t b[32]) {
    uint64_t w[4];
    memcpy(w, b, 32);
    fe r;
    r.v[0] = w[0] & ORC_MASK51;
    r.v[1] = ((w[0] >> 51) | (w[1] << 13)) & ORC_MASK51;
    r.v[2] = ((w[1] >> 38) | (w[2] << 26)) & ORC_MASK51;
    r.v[3] = ((w[2] >> 25) | (w[3] << 39)) & ORC_MASK51;
    r.v[4] = (w[3] >> 12) & ORC_MASK51;
    return r;
}

/* u64/field.rs:368-450 -- canonical encoding */
static inline void fe_to_bytes(uint8_t s[32], fe a) {
    a = fe_reduce(a);
    uint64_t *l = a.v;
    uint64_t q = (l[0] + 19) >> 51;
    q = (l[1] + q) >> 51; q = (l[2] + q) >> 51; q = (l[3] + q) >> 51; q = (l[4] + q) >> 51;
    l[0] += 19 * q;
    l[1] += l[0] >> 51; l[0] &= ORC_MASK51;
    l[2] += l[1] >> 51; l[1] &= ORC_MASK51;
    l[3] += l[2] >> 51; l[2] &= ORC_MASK51;
    l[4] += l[3] >> 51; l[3] &= ORC_MASK51;
    l[4] &= ORC_MASK51;
    uint64_t w[4];
    w[0] = l[0] | (l[1] << 51);
    w[1] = (l[1] >> 13) | (l[2] << 38);
    w[2] = (l[2] >> 26) | (l[3] << 25);
    w[3] = (l[3] >> 39) | (l[4] << 12);
    memcpy(s, w, 32);
}

/* field.rs:156-170 */
static inline int fe_is_negative(fe a) { uint8_t b[32]; fe_to_bytes(b, a); return b[0] & 1; }
static inline int fe_is_zero(fe a) {
    uint8_t b[32]; fe_to_bytes(b, a);
    uint8_t acc = 0; for (int i = 0; i < 32; i++) acc |= b[i];
    return acc == 0;
}
/* u64/field.rs:225-255 (ConstantTimeEq compares canonical bytes) */
static inline int fe_eq(fe a, fe b) {
    uint8_t x[32], y[32]; fe_to_bytes(x, a); fe_to_bytes(y, b);
    return memcmp(x, y, 32) == 0;
}
static inline fe fe_select(fe a, fe b, int choose_b) { return choose_b ? b : a; }
static inline fe fe_cneg(fe a, int neg) { return neg ? fe_neg(a) : a; }

/* field.rs:176-210 -- (x^(2^250-1), x^11) */
static inline void fe_pow22501(fe x, fe *t19_out, fe *t3_out) {
    fe t0 = fe_sq(x);
    fe t1 = fe_sq(fe_sq(t0));
    fe t2 = fe_mul(x, t1);
    fe t3 = fe_mul(t0, t2);
    fe t4 = fe_sq(t3);
    fe t5 = fe_mul(t2, t4);
    fe t6 = fe_pow2k(t5, 5);
    fe t7 = fe_mul(t6, t5);
    fe t8 = fe_pow2k(t7, 10);
    fe t9 = fe_mul(t8, t7);
    fe t10 = fe_pow2k(t9, 20);
    fe t11 = fe_mul(t10, t9);
    fe t12 = fe_pow2k(t11, 10);
    fe t13 = fe_mul(t12, t7);
    fe t14 = fe_pow2k(t13, 50);
    fe t15 = fe_mul(t14, t13);
    fe t16 = fe_pow2k(t15, 100);
    fe t17 = fe_mul(t16, t15);
    fe t18 = fe_pow2k(t17, 50);
    *t19_out = fe_mul(t18, t13);
    *t3_out = t3;
}

/* field.rs:283-292 -- x^(p-2); 0 -> 0 */
static inline fe fe_invert(fe x) {
    fe t19, t3; fe_pow22501(x, &t19, &t3);
    return fe_mul(fe_pow2k(t19, 5), t3);
}

/* field.rs:297-306 -- x^((p-5)/8) */
static inline fe fe_pow_p58(fe x) {
    fe t19, t3; fe_pow22501(x, &t19, &t3);
    return fe_mul(x, fe_pow2k(t19, 2));
}

/* field.rs:320-366 -- returns was_nonzero_square, *r = the non-negative root */
static inline int fe_sqrt_ratio_i(fe *r_out, fe u, fe v) {
    fe v3 = fe_mul(fe_sq(v), v);
    fe v7 = fe_mul(fe_sq(v3), v);
    fe r = fe_mul(fe_mul(u, v3), fe_pow_p58(fe_mul(u, v7)));
    fe check = fe_mul(v, fe_sq(r));
    fe i = fe_from_limbs(ORC_SQRT_M1);
    fe neg_u = fe_neg(u);
    int correct_sign = fe_eq(check, u);
    int flipped_sign = fe_eq(check, neg_u);
    int flipped_sign_i = fe_eq(check, fe_mul(neg_u, i));
    fe r_prime = fe_mul(i, r);
    r = fe_select(r, r_prime, flipped_sign | flipped_sign_i);
    r = fe_cneg(r, fe_is_negative(r));
    *r_out = r;
    return correct_sign | flipped_sign;
}

/* field.rs:368-376 */
static inline int fe_invsqrt(fe *r_out, fe v) { return fe_sqrt_ratio_i(r_out, FE_ONE, v); }

/* field.rs:225-273 -- Montgomery's trick; zero inputs stay zero */
static inline void fe_invert_batch(fe *inputs, fe *scratch, size_t n) {
    fe acc = FE_ONE;
    for (size_t i = 0; i < n; i++) {
        scratch[i] = acc;
        if (!fe_is_zero(inputs[i])) acc = fe_mul(acc, inputs[i]);
    }
    acc = fe_invert(acc);
    for (size_t i = n; i-- > 0;) {
        fe tmp = fe_mul(acc, inputs[i]);
        if (!fe_is_zero(inputs[i])) {
            inputs[i] = fe_mul(acc, scratch[i]);
            acc = tmp;
        }
    }
}

#endif
