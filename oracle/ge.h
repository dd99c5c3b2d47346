/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's serial curve models, lookup tables and scalar-multiplication
 * algorithms (the "serial" backend behind curve25519-dalek/src/backend.rs):
 *   backend/serial/curve_models.rs          point models and mixed-model formulas
 *   window.rs                               LookupTable / NafLookupTable5 / NafLookupTable8
 *   backend/serial/scalar_mul/{..}.rs         pippenger, straus (vartime), variable_base, vartime_double_base
 *   edwards.rs                              EdwardsPoint API, (de)compression, basepoint table
 */
#ifndef ORC_GE_H
#define ORC_GE_H
#include <stdlib.h>
#include "fe51.h"
#include "sc52.h"

typedef struct { fe X, Y, Z, T; } ge_p3;      /* EdwardsPoint            edwards.rs:390-395 */
typedef struct { fe X, Y, Z; } ge_p2;         /* ProjectivePoint         curve_models.rs:154 */
typedef struct { fe X, Y, Z, T; } ge_p1p1;    /* CompletedPoint          curve_models.rs:169 */
typedef struct { fe ypx, ymx, xy2d; } ge_aniels;   /* AffineNielsPoint   curve_models.rs:184 */
typedef struct { fe YpX, YmX, Z, T2d; } ge_pniels; /* ProjectiveNielsPoint curve_models.rs:206 */

static inline ge_p3 ge_identity(void) { ge_p3 r = {FE_ZERO, FE_ONE, FE_ONE, FE_ZERO}; return r; }
static inline ge_p2 ge_p2_identity(void) { ge_p2 r = {FE_ZERO, FE_ONE, FE_ONE}; return r; }
static inline ge_aniels ge_aniels_identity(void) { ge_aniels r = {FE_ONE, FE_ONE, FE_ZERO}; return r; }
static inline ge_pniels ge_pniels_identity(void) { ge_pniels r = {FE_ONE, FE_ONE, FE_ONE, FE_ZERO}; return r; }

static inline ge_p3 ge_basepoint(void) {
    ge_p3 b = {fe_from_limbs(ORC_BASEPOINT_X), fe_from_limbs(ORC_BASEPOINT_Y), FE_ONE, fe_from_limbs(ORC_BASEPOINT_T)};
    return b;
}

/* conversions: curve_models.rs:333-373, edwards.rs:528-560 */
static inline ge_p2 ge_p3_to_p2(ge_p3 p) { ge_p2 r = {p.X, p.Y, p.Z}; return r; }
static inline ge_p3 ge_p2_to_p3(ge_p2 p) {
    ge_p3 r = {fe_mul(p.X, p.Z), fe_mul(p.Y, p.Z), fe_sq(p.Z), fe_mul(p.X, p.Y)}; return r;
}
static inline ge_p2 ge_p1p1_to_p2(ge_p1p1 p) {
    ge_p2 r = {fe_mul(p.X, p.T), fe_mul(p.Y, p.Z), fe_mul(p.Z, p.T)}; return r;
}
static inline ge_p3 ge_p1p1_to_p3(ge_p1p1 p) {
    ge_p3 r = {fe_mul(p.X, p.T), fe_mul(p.Y, p.Z), fe_mul(p.Z, p.T), fe_mul(p.X, p.Y)}; return r;
}
static inline ge_pniels ge_p3_to_pniels(ge_p3 p) {
    ge_pniels r = {fe_add(p.Y, p.X), fe_sub(p.Y, p.X), p.Z, fe_mul(p.T, fe_from_limbs(ORC_EDWARDS_D2))}; return r;
}
static inline ge_aniels ge_p3_to_aniels(ge_p3 p) { /* edwards.rs:549-560 */
    fe recip = fe_invert(p.Z);
    fe x = fe_mul(p.X, recip), y = fe_mul(p.Y, recip);
    ge_aniels r = {fe_add(y, x), fe_sub(y, x), fe_mul(fe_mul(x, y), fe_from_limbs(ORC_EDWARDS_D2))};
    return r;
}

/* curve_models.rs:381-397 */
static inline ge_p1p1 ge_p2_dbl(ge_p2 p) {
    fe XX = fe_sq(p.X), YY = fe_sq(p.Y), ZZ2 = fe_sq2(p.Z);
    fe XpY_sq = fe_sq(fe_add(p.X, p.Y));
    fe YYpXX = fe_add(YY, XX), YYmXX = fe_sub(YY, XX);
    ge_p1p1 r = {fe_sub(XpY_sq, YYpXX), YYpXX, YYmXX, fe_sub(ZZ2, YYmXX)};
    return r;
}

/* curve_models.rs:411-451 */
static inline ge_p1p1 ge_add_pniels(ge_p3 p, ge_pniels q) {
    fe PP = fe_mul(fe_add(p.Y, p.X), q.YpX), MM = fe_mul(fe_sub(p.Y, p.X), q.YmX);
    fe TT2d = fe_mul(p.T, q.T2d), ZZ = fe_mul(p.Z, q.Z), ZZ2 = fe_add(ZZ, ZZ);
    ge_p1p1 r = {fe_sub(PP, MM), fe_add(PP, MM), fe_add(ZZ2, TT2d), fe_sub(ZZ2, TT2d)};
    return r;
}
static inline ge_p1p1 ge_sub_pniels(ge_p3 p, ge_pniels q) {
    fe PM = fe_mul(fe_add(p.Y, p.X), q.YmX), MP = fe_mul(fe_sub(p.Y, p.X), q.YpX);
    fe TT2d = fe_mul(p.T, q.T2d), ZZ = fe_mul(p.Z, q.Z), ZZ2 = fe_add(ZZ, ZZ);
    ge_p1p1 r = {fe_sub(PM, MP), fe_add(PM, MP), fe_sub(ZZ2, TT2d), fe_add(ZZ2, TT2d)};
    return r;
}
/* curve_models.rs:455-494 */
static inline ge_p1p1 ge_add_aniels(ge_p3 p, ge_aniels q) {
    fe PP = fe_mul(fe_add(p.Y, p.X), q.ypx), MM = fe_mul(fe_sub(p.Y, p.X), q.ymx);
    fe Txy2d = fe_mul(p.T, q.xy2d), Z2 = fe_add(p.Z, p.Z);
    ge_p1p1 r = {fe_sub(PP, MM), fe_add(PP, MM), fe_add(Z2, Txy2d), fe_sub(Z2, Txy2d)};
    return r;
}
static inline ge_p1p1 ge_sub_aniels(ge_p3 p, ge_aniels q) {
    fe PM = fe_mul(fe_add(p.Y, p.X), q.ymx), MP = fe_mul(fe_sub(p.Y, p.X), q.ypx);
    fe Txy2d = fe_mul(p.T, q.xy2d), Z2 = fe_add(p.Z, p.Z);
    ge_p1p1 r = {fe_sub(PM, MP), fe_add(PM, MP), fe_sub(Z2, Txy2d), fe_add(Z2, Txy2d)};
    return r;
}

/* edwards.rs:795-872 */
static inline ge_p3 ge_add(ge_p3 a, ge_p3 b) { return ge_p1p1_to_p3(ge_add_pniels(a, ge_p3_to_pniels(b))); }
static inline ge_p3 ge_sub(ge_p3 a, ge_p3 b) { return ge_p1p1_to_p3(ge_sub_pniels(a, ge_p3_to_pniels(b))); }
static inline ge_p3 ge_neg(ge_p3 a) { ge_p3 r = {fe_neg(a.X), a.Y, a.Z, fe_neg(a.T)}; return r; }
static inline ge_p3 ge_dbl(ge_p3 a) { return ge_p1p1_to_p3(ge_p2_dbl(ge_p3_to_p2(a))); }
static inline ge_aniels ge_aniels_neg(ge_aniels a) { ge_aniels r = {a.ymx, a.ypx, fe_neg(a.xy2d)}; return r; }
static inline ge_pniels ge_pniels_neg(ge_pniels a) { ge_pniels r = {a.YmX, a.YpX, a.Z, fe_neg(a.T2d)}; return r; }

/* edwards.rs:1370-1380 */
static inline ge_p3 ge_mul_by_pow_2(ge_p3 p, unsigned k) {
    ge_p2 s = ge_p3_to_p2(p);
    for (unsigned i = 0; i + 1 < k; i++) s = ge_p1p1_to_p2(ge_p2_dbl(s));
    return ge_p1p1_to_p3(ge_p2_dbl(s));
}

/* edwards.rs:501-511, traits.rs:45 */
static inline int ge_eq(ge_p3 a, ge_p3 b) {
    return fe_eq(fe_mul(a.X, b.Z), fe_mul(b.X, a.Z)) & fe_eq(fe_mul(a.Y, b.Z), fe_mul(b.Y, a.Z));
}
static inline int ge_is_identity(ge_p3 a) { return ge_eq(a, ge_identity()); }
static inline int ge_is_small_order(ge_p3 a) { return ge_is_identity(ge_mul_by_pow_2(a, 3)); } /* edwards.rs:1405 */

/* edwards.rs:211-258 -- ZIP-215 style decoding: non-canonical y accepted, sign applied blindly */
static inline int ge_decompress(ge_p3 *out, const uint8_t b[32]) {
    fe Y = fe_from_bytes(b), Z = FE_ONE, YY = fe_sq(Y);
    fe u = fe_sub(YY, Z);
    fe v = fe_add(fe_mul(YY, fe_from_limbs(ORC_EDWARDS_D)), Z);
    fe X; int ok = fe_sqrt_ratio_i(&X, u, v);
    if (!ok) return 0;
    X = fe_cneg(X, b[31] >> 7);
    out->X = X; out->Y = Y; out->Z = Z; out->T = fe_mul(X, Y);
    return 1;
}

/* edwards.rs:615, edwards/affine.rs:71-75 */
static inline void ge_affine_compress(uint8_t s[32], fe x, fe y) {
    fe_to_bytes(s, y);
    s[31] ^= (uint8_t)(fe_is_negative(x) << 7);
}
static inline void ge_compress(uint8_t s[32], ge_p3 p) {
    fe recip = fe_invert(p.Z);
    ge_affine_compress(s, fe_mul(p.X, recip), fe_mul(p.Y, recip));
}

/* window.rs:97-116 (radix-16 LookupTable: P,2P,..,8P) and :54-76 (select) */
static inline void ge_table_pniels(ge_pniels t[8], ge_p3 P) {
    t[0] = ge_p3_to_pniels(P);
    for (int j = 0; j < 7; j++) t[j + 1] = ge_p3_to_pniels(ge_p1p1_to_p3(ge_add_pniels(P, t[j])));
}
static inline void ge_table_aniels(ge_aniels t[8], ge_p3 P) {
    t[0] = ge_p3_to_aniels(P);
    for (int j = 0; j < 7; j++) t[j + 1] = ge_p3_to_aniels(ge_p1p1_to_p3(ge_add_aniels(P, t[j])));
}
static inline ge_pniels ge_select_pniels(const ge_pniels t[8], int8_t x) {
    int xabs = x < 0 ? -x : x;
    ge_pniels r = ge_pniels_identity();
    for (int j = 1; j <= 8; j++) if (xabs == j) r = t[j - 1];
    return x < 0 ? ge_pniels_neg(r) : r;
}
static inline ge_aniels ge_select_aniels(const ge_aniels t[8], int8_t x) {
    int xabs = x < 0 ? -x : x;
    ge_aniels r = ge_aniels_identity();
    for (int j = 1; j <= 8; j++) if (xabs == j) r = t[j - 1];
    return x < 0 ? ge_aniels_neg(r) : r;
}
/* window.rs:201-211 -- odd multiples A,3A,..,15A */
static inline void ge_naf5_table(ge_pniels t[8], ge_p3 A) {
    t[0] = ge_p3_to_pniels(A);
    ge_p3 A2 = ge_dbl(A);
    for (int i = 0; i < 7; i++) t[i + 1] = ge_p3_to_pniels(ge_p1p1_to_p3(ge_add_pniels(A2, t[i])));
}
/* window.rs:266-276 -- odd multiples A,3A,..,127A in affine Niels form */
static inline void ge_naf8_table_aniels(ge_aniels t[64], ge_p3 A) {
    t[0] = ge_p3_to_aniels(A);
    ge_p3 A2 = ge_dbl(A);
    for (int i = 0; i < 63; i++) t[i + 1] = ge_p3_to_aniels(ge_p1p1_to_p3(ge_add_aniels(A2, t[i])));
}

/* ---- scalar multiplication algorithms ------------------------------------------------ */

/* backend/serial/scalar_mul/variable_base.rs:11-47 */
static inline ge_p3 ge_variable_base_mul(ge_p3 P, const uint8_t s[32]) {
    ge_pniels table[8]; ge_table_pniels(table, P);
    int8_t d[64]; sc_as_radix_16(d, s);
    ge_p1p1 t1 = ge_add_pniels(ge_identity(), ge_select_pniels(table, d[63]));
    for (int i = 62; i >= 0; i--) {
        ge_p2 t2 = ge_p1p1_to_p2(t1);
        t1 = ge_p2_dbl(t2); t2 = ge_p1p1_to_p2(t1);
        t1 = ge_p2_dbl(t2); t2 = ge_p1p1_to_p2(t1);
        t1 = ge_p2_dbl(t2); t2 = ge_p1p1_to_p2(t1);
        t1 = ge_p2_dbl(t2);
        t1 = ge_add_pniels(ge_p1p1_to_p3(t1), ge_select_pniels(table, d[i]));
    }
    return ge_p1p1_to_p3(t1);
}

/* edwards.rs:1125-1141 (create) and :1192-1209 (mul_base), radix-16 instance :1248-1254.
   The 32x8 table is regenerated by the `create` logic, not transcribed. */
typedef struct { ge_aniels t[32][8]; } ge_basepoint_table;
static inline void ge_basepoint_table_create(ge_basepoint_table *tab, ge_p3 B) {
    ge_p3 P = B;
    for (int i = 0; i < 32; i++) {
        ge_table_aniels(tab->t[i], P);
        P = ge_mul_by_pow_2(P, 8);
    }
}
static inline ge_p3 ge_mul_base_table(const ge_basepoint_table *tab, const uint8_t s[32]) {
    int8_t a[64]; sc_as_radix_16(a, s);
    ge_p3 P = ge_identity();
    for (int i = 1; i < 64; i += 2) P = ge_p1p1_to_p3(ge_add_aniels(P, ge_select_aniels(tab->t[i / 2], a[i])));
    P = ge_mul_by_pow_2(P, 4);
    for (int i = 0; i < 64; i += 2) P = ge_p1p1_to_p3(ge_add_aniels(P, ge_select_aniels(tab->t[i / 2], a[i])));
    return P;
}

/* backend/serial/scalar_mul/straus.rs:159-200 -- vartime, NAF-5, shared doublings */
static inline ge_p3 ge_straus_vartime(const uint8_t *scalars, const ge_p3 *points, size_t n) {
    int8_t (*nafs)[256] = malloc(n ? n * 256 : 1);
    ge_pniels (*tabs)[8] = malloc(n ? n * sizeof(ge_pniels[8]) : 1);
    for (size_t k = 0; k < n; k++) { sc_non_adjacent_form(nafs[k], scalars + 32 * k, 5); ge_naf5_table(tabs[k], points[k]); }
    ge_p2 r = ge_p2_identity();
    for (int i = 255; i >= 0; i--) {
        ge_p1p1 t = ge_p2_dbl(r);
        for (size_t k = 0; k < n; k++) {
            int d = nafs[k][i];
            if (d > 0) t = ge_add_pniels(ge_p1p1_to_p3(t), tabs[k][d / 2]);
            else if (d < 0) t = ge_sub_pniels(ge_p1p1_to_p3(t), tabs[k][(-d) / 2]);
        }
        r = ge_p1p1_to_p2(t);
    }
    free(nafs); free(tabs);
    return ge_p2_to_p3(r);
}

/* backend/serial/scalar_mul/pippenger.rs:67-160 -- bucket method, signed radix 2^w, w = 6|7|8 */
static inline ge_p3 ge_pippenger_vartime(const uint8_t *scalars, const ge_p3 *points, size_t n) {
    unsigned w = n < 500 ? 6 : (n < 800 ? 7 : 8);
    size_t max_digit = (size_t)1 << w, digits_count = sc_radix_2w_size_hint(w), buckets_count = max_digit / 2;
    int8_t (*digits)[64] = malloc(n ? n * 64 : 1);
    ge_pniels *pn = malloc(n ? n * sizeof(ge_pniels) : 1);
    for (size_t k = 0; k < n; k++) { sc_as_radix_2w(digits[k], scalars + 32 * k, w); pn[k] = ge_p3_to_pniels(points[k]); }
    ge_p3 *buckets = malloc(buckets_count * sizeof(ge_p3));
    ge_p3 total = ge_identity();
    for (size_t di = digits_count; di-- > 0;) {
        for (size_t b = 0; b < buckets_count; b++) buckets[b] = ge_identity();
        for (size_t k = 0; k < n; k++) {
            int digit = digits[k][di];
            if (digit > 0) { size_t b = (size_t)(digit - 1); buckets[b] = ge_p1p1_to_p3(ge_add_pniels(buckets[b], pn[k])); }
            else if (digit < 0) { size_t b = (size_t)(-digit - 1); buckets[b] = ge_p1p1_to_p3(ge_sub_pniels(buckets[b], pn[k])); }
        }
        ge_p3 isum = buckets[buckets_count - 1], sum = buckets[buckets_count - 1];
        for (size_t i = buckets_count - 1; i-- > 0;) { isum = ge_add(isum, buckets[i]); sum = ge_add(sum, isum); }
        if (di == digits_count - 1) total = sum;
        else total = ge_add(ge_mul_by_pow_2(total, w), sum);
    }
    free(digits); free(pn); free(buckets);
    return total;
}

/* edwards.rs:1002-1031 -- size dispatch: Straus below 190 terms, Pippenger otherwise */
static inline ge_p3 ge_multiscalar_mul_vartime(const uint8_t *scalars, const ge_p3 *points, size_t n) {
    return n < 190 ? ge_straus_vartime(scalars, points, n) : ge_pippenger_vartime(scalars, points, n);
}

/* backend/serial/scalar_mul/vartime_double_base.rs:23-72 -- aA + bB, NAF-5 on A, NAF-8 on the
   static odd-multiples-of-B table (regenerated here by the NafLookupTable8::from logic) */
static inline ge_p3 ge_vartime_double_base_mul(const uint8_t a[32], ge_p3 A, const uint8_t b[32], const ge_aniels *tableB) {
    int8_t a_naf[256], b_naf[256];
    sc_non_adjacent_form(a_naf, a, 5); sc_non_adjacent_form(b_naf, b, 8);
    int i = 255;
    for (int j = 255; j >= 0; j--) { i = j; if (a_naf[i] != 0 || b_naf[i] != 0) break; }
    ge_pniels tableA[8]; ge_naf5_table(tableA, A);
    ge_p2 r = ge_p2_identity();
    for (;;) {
        ge_p1p1 t = ge_p2_dbl(r);
        if (a_naf[i] > 0) t = ge_add_pniels(ge_p1p1_to_p3(t), tableA[a_naf[i] / 2]);
        else if (a_naf[i] < 0) t = ge_sub_pniels(ge_p1p1_to_p3(t), tableA[(-a_naf[i]) / 2]);
        if (b_naf[i] > 0) t = ge_add_aniels(ge_p1p1_to_p3(t), tableB[b_naf[i] / 2]);
        else if (b_naf[i] < 0) t = ge_sub_aniels(ge_p1p1_to_p3(t), tableB[(-b_naf[i]) / 2]);
        r = ge_p1p1_to_p2(t);
        if (i == 0) break;
        i--;
    }
    return ge_p2_to_p3(r);
}

#endif
