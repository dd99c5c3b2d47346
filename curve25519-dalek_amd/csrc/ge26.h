// Edwards / Montgomery point formulas for gfx950, one point per lane, on the fe26 field layer.
//
// Device counterpart of the reference's serial curve models
// (curve25519-dalek/src/backend/serial/curve_models.rs:154-494) and the formulas in
// edwards.rs / montgomery.rs that the hot path inlines.  The operation ORDER of each formula is
// re-derived for the unsigned radix-2^25.5 limbs (which operand of each product may be "wide",
// where a carry pass is needed) -- see the bound comments; results are compared only through
// canonical encodings / projective equality, exactly like the reference's tests.
#pragma once
#include "fe26.h"
#include "constants_gen.h"

namespace c25519 {

struct ge_p3 { feT X, Y, Z, T; };          // EdwardsPoint, edwards.rs:390-395
struct ge_p2 { feT X, Y, Z; };             // ProjectivePoint, curve_models.rs:154
// CompletedPoint (curve_models.rs:169).  Every producer below yields X, Y, Z loose and only T wide, so
// the conversions can always put T first in its two products and need no carry pass.
struct ge_p1p1 { feL X, Y, Z; feW T; };
struct ge_aniels { feT ypx, ymx, xy2d; };  // AffineNielsPoint, curve_models.rs:184
struct ge_cached { feL YpX, YmX; feT Z; feL T2d; };  // ProjectiveNielsPoint, curve_models.rs:206

C25519_HD feT fe_const(const u32 (&c)[10]) { feT r; for (int i = 0; i < 10; i++) r.v[i] = c[i]; return r; }
C25519_HD feT fe_d() { const u32 c[10] = C25519_EDWARDS_D_26; return fe_const(c); }
C25519_HD feT fe_d2() { const u32 c[10] = C25519_EDWARDS_D2_26; return fe_const(c); }
C25519_HD feT fe_d_inv() { const u32 c[10] = C25519_EDWARDS_D_INV_26; return fe_const(c); }
C25519_HD feT fe_sqrtm1() { const u32 c[10] = C25519_SQRT_M1_26; return fe_const(c); }
C25519_HD feT fe_invsqrt_a_minus_d() { const u32 c[10] = C25519_INVSQRT_A_MINUS_D_26; return fe_const(c); }

C25519_HD ge_p3 ge_identity() { ge_p3 r; r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.T = fe_zero(); return r; }
C25519_HD ge_p3 ge_basepoint() {
    const u32 x[10] = C25519_BASEPOINT_X_26, y[10] = C25519_BASEPOINT_Y_26, t[10] = C25519_BASEPOINT_T_26;
    ge_p3 r; r.X = fe_const(x); r.Y = fe_const(y); r.Z = fe_one(); r.T = fe_const(t); return r;
}

// Loop-carried points: LLVM hoists the zero-extension of the u32 limbs across the loop PHI and then carries
// them as 64-bit values, which turns every product with such a limb into a 64 x 32 multiply (two
// v_mad_u64_u32 and two moves instead of one: +7 % multiplier work in k_mul_base_wide).  An empty asm on every
// limb at the loop boundary keeps them 32-bit registers.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void fe_pin(feT &a) { for (int i = 0; i < 10; i++) asm("" : "+v"(a.v[i])); }
#else
C25519_HD void fe_pin(feT &) {}
#endif
C25519_HD void ge_pin(ge_p3 &p) { fe_pin(p.X); fe_pin(p.Y); fe_pin(p.Z); fe_pin(p.T); }

// CompletedPoint -> EdwardsPoint, curve_models.rs:365-373 (4 M).  The wide operand goes first.
C25519_HD ge_p3 ge_p1p1_to_p3(const ge_p1p1 &p) {
    ge_p3 r;
    r.X = fe_mul(p.T, p.X);
    r.Y = fe_mul(p.Z, p.Y);
    r.Z = fe_mul(p.T, p.Z);
    r.T = fe_mul(p.X, p.Y);
    return r;
}
// CompletedPoint -> ProjectivePoint, curve_models.rs:353-359 (3 M)
C25519_HD ge_p2 ge_p1p1_to_p2(const ge_p1p1 &p) {
    ge_p2 r;
    r.X = fe_mul(p.T, p.X);
    r.Y = fe_mul(p.Z, p.Y);
    r.Z = fe_mul(p.T, p.Z);
    return r;
}

// Doubling, curve_models.rs:381-397 (4 S).  Works on (X,Y,Z) of either a p2 or a p3.
C25519_HD ge_p1p1 ge_dbl(const feT &X, const feT &Y, const feT &Z) {
    feT XX = fe_sq(X), YY = fe_sq(Y), ZZ = fe_sq(Z);
    feL ZZ2 = fe_twice(ZZ);
    feT S = fe_sq(fe_add(X, Y));
    feL YYpXX = fe_add(YY, XX), YYmXX = fe_sub(YY, XX);
    ge_p1p1 r;
    r.X = fe_sub(S, fe_carry(YYpXX));   // tight - tight -> loose
    r.Y = YYpXX;
    r.Z = YYmXX;
    r.T = fe_sub_w(ZZ2, YYmXX);         // loose - loose -> wide
    return r;
}
C25519_HD ge_p3 ge_dbl_p3(const ge_p3 &p) { return ge_p1p1_to_p3(ge_dbl(p.X, p.Y, p.Z)); }

// edwards.rs:1370-1380 mul_by_pow_2
C25519_HD ge_p3 ge_mul_by_pow_2(const ge_p3 &p, int k) {
    ge_p2 s; s.X = p.X; s.Y = p.Y; s.Z = p.Z;
    for (int i = 0; i < k - 1; i++) s = ge_p1p1_to_p2(ge_dbl(s.X, s.Y, s.Z));
    return ge_p1p1_to_p3(ge_dbl(s.X, s.Y, s.Z));
}

// Extended + AffineNiels (curve_models.rs:455-472), 3 M.  Subtraction (:476-494) is addition of the
// negated operand: -(y+x, y-x, 2dxy) = (y-x, y+x, -2dxy), applied to the PACKED table words by
// aniels_words_cneg before unpacking (24 VALU ops instead of 80 selects on limbs and outputs).
C25519_HD ge_p1p1 ge_madd(const ge_p3 &p, const ge_aniels &q) {
    feL YpX = fe_add(p.Y, p.X), YmX = fe_sub(p.Y, p.X);
    feT PP = fe_mul(YpX, q.ypx), MM = fe_mul(YmX, q.ymx);
    feT TT = fe_mul(p.T, q.xy2d);
    feL Z2 = fe_twice(p.Z);
    ge_p1p1 r;
    r.X = fe_sub(PP, MM);
    r.Y = fe_add(PP, MM);
    r.Z = fe_add_lt(Z2, TT);
    r.T = fe_sub_w(Z2, TT);
    return r;
}
// P + (neg ? -Q : Q) straight to extended coordinates, for table entries kept as LIMBS (no unpacking, no negation
// arithmetic): -Q = (y-x, y+x, -2dxy) swaps the first two entries, and the sign of TT = T * 2dxy only decides which of
// Z2 + TT / Z2 - TT plays Z and which plays T of the completed point.  Both are needed anyway; Z3 = (Z2+TT)(Z2-TT)
// either way, and the wide one of the pair is always put first in its product, so 40 selects replace the
// conditional subtraction p - v and the three unpackings.  7 M like ge_madd + ge_p1p1_to_p3.
C25519_HD ge_p3 ge_madd_signed_p3(const ge_p3 &p, const ge_aniels &q, bool neg) {
    feT qa, qb;
    const lanemask nm = lane_mask(neg);
    for (int i = 0; i < 10; i++) { qa.v[i] = sel_u32(q.ypx.v[i], q.ymx.v[i], nm); qb.v[i] = sel_u32(q.ymx.v[i], q.ypx.v[i], nm); }
    feL YpX = fe_add(p.Y, p.X), YmX = fe_sub(p.Y, p.X);
    feT PP = fe_mul(YpX, qa), MM = fe_mul(YmX, qb);
    feT TT = fe_mul(p.T, q.xy2d);
    feL Z2 = fe_twice(p.Z);
    feL X = fe_sub(PP, MM), Y = fe_add(PP, MM);
    feL zp = fe_add_lt(Z2, TT);            // Z of the completed point for +Q, T for -Q   (loose)
    feW zm = fe_sub_w(Z2, TT);             // T of the completed point for +Q, Z for -Q   (wide)
    feW fx, fy;                            // the factor of X (the completed T) and of Y (the completed Z)
    for (int i = 0; i < 10; i++) { fx.v[i] = sel_u32(zm.v[i], zp.v[i], nm); fy.v[i] = sel_u32(zp.v[i], zm.v[i], nm); }
    ge_p3 r;
    r.X = fe_mul(fx, X);
    r.Y = fe_mul(fy, Y);
    r.Z = fe_mul(zm, zp);
    r.T = fe_mul(X, Y);
    return r;
}

// The same addition with the sign LAZILY on the accumulator (r6, last; fe26x.h ge_madd_lazy_p3_lockstep has the reasoning): the caller keeps whether the stored point is
// the true sum or its negative; `flip` (all ones / zero) says that it changes sides -- X and T negated, one v_xad_u32 per limb -- before Q is added as it is.
C25519_HD ge_p3 ge_madd_lazy_p3(const ge_p3 &p, const ge_aniels &q, u32 flip) {
    const feL Xs = fe_cond_neg(p.X, flip), Ts = fe_cond_neg(p.T, flip);
    feW YpX = fe_add_w(feL(p.Y), Xs), YmX = fe_sub_w(feL(p.Y), Xs);
    feT PP = fe_mul(YpX, q.ypx), MM = fe_mul(YmX, q.ymx);
    feT TT = fe_mul(feW(Ts), q.xy2d);
    feL Z2 = fe_twice(p.Z);
    feL X = fe_sub(PP, MM), Y = fe_add(PP, MM);
    feL zp = fe_add_lt(Z2, TT);
    feW zm = fe_sub_w(Z2, TT);
    ge_p3 r;
    r.X = fe_mul(zm, X);
    r.Y = fe_mul(feW(zp), Y);
    r.Z = fe_mul(zm, zp);
    r.T = fe_mul(feW(X), Y);
    return r;
}

// (neg ? -Q : Q) as an extended point, for the FIRST addition of a chain (identity + Q): with (y+x, y-x, 2dxy) at hand,
// (X : Y : Z : T) = (2x : 2y : 2 : 2xy) is (y+x) - (y-x), (y+x) + (y-x), 2 and 2dxy / d -- one multiplication by the
// constant 1/d instead of the 7 M of a mixed addition onto the identity.  -Q swaps the first two entries, which negates X,
// and negates T.
C25519_HD ge_p3 ge_from_aniels_signed(const ge_aniels &q, bool neg) {
    feT qa, qb;
    const lanemask nm = lane_mask(neg);
    for (int i = 0; i < 10; i++) { qa.v[i] = sel_u32(q.ypx.v[i], q.ymx.v[i], nm); qb.v[i] = sel_u32(q.ymx.v[i], q.ypx.v[i], nm); }
    feT t = fe_mul(q.xy2d, fe_d_inv());
    feT tn = fe_carry(fe_neg(t));
    ge_p3 r;
    r.X = fe_carry(fe_sub(qa, qb));
    r.Y = fe_carry(fe_add(qa, qb));
    r.Z = fe_small(2);
    for (int i = 0; i < 10; i++) r.T.v[i] = sel_u32(t.v[i], tn.v[i], nm);
    return r;
}

// w[0..7] = y+x, w[8..15] = y-x, w[16..23] = 2dxy as canonical 255-bit words.  neg: swap the first
// two and replace the third by p - v (v = 0 gives p, a non-canonical but valid representative of 0).
C25519_HD void aniels_words_cneg(u32 w[24], bool neg) {
    const lanemask nm = lane_mask(neg);
    for (int i = 0; i < 8; i++) { u32 a = w[i], b = w[8 + i]; w[i] = sel_u32(a, b, nm); w[8 + i] = sel_u32(b, a, nm); }
    const u32 P0 = 0xffffffedu, PM = 0xffffffffu, P7 = 0x7fffffffu;
    u64 borrow = 0;
    for (int i = 0; i < 8; i++) {
        u64 pi = (i == 0) ? P0 : (i == 7 ? P7 : PM);
        u64 d = pi - (u64)w[16 + i] - borrow;
        borrow = (d >> 63) & 1;
        w[16 + i] = sel_u32(w[16 + i], (u32)d, nm);
    }
}
C25519_HD ge_aniels aniels_from_words(const u32 w[24]) {
    ge_aniels A;
    A.ypx = fe_from_words(w); A.ymx = fe_from_words(w + 8); A.xy2d = fe_from_words(w + 16);
    return A;
}

// EdwardsPoint -> ProjectiveNiels (edwards.rs:528-535), 1 M
C25519_HD ge_cached ge_p3_to_cached(const ge_p3 &p) {
    ge_cached r;
    r.YpX = fe_add(p.Y, p.X); r.YmX = fe_sub(p.Y, p.X); r.Z = p.Z; r.T2d = fe_mul(p.T, fe_d2());
    return r;
}
// -(Y+X, Y-X, Z, 2dT) = (Y-X, Y+X, Z, -2dT), chosen per lane (curve_models.rs:501-512 Neg)
C25519_HD ge_cached ge_cached_cneg(const ge_cached &c, bool neg) {
    ge_cached r;
    r.YpX = fe_select(c.YpX, c.YmX, neg); r.YmX = fe_select(c.YmX, c.YpX, neg); r.Z = c.Z;
    feL nt; nt.v[0] = 0x7ffffdau - c.T2d.v[0];
    for (int i = 1; i < 10; i++) nt.v[i] = ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - c.T2d.v[i];   // 2p - T2d (T2d tight or 2p-tight)
    r.T2d = fe_select(c.T2d, nt, neg);
    return r;
}
// Extended + ProjectiveNiels (curve_models.rs:411-429), 4 M
C25519_HD ge_p1p1 ge_add_cached(const ge_p3 &p, const ge_cached &q) {
    feL YpX = fe_add(p.Y, p.X), YmX = fe_sub(p.Y, p.X);
    feT PP = fe_mul(YpX, q.YpX), MM = fe_mul(YmX, q.YmX);
    feT TT = fe_mul(p.T, q.T2d), ZZ = fe_mul(p.Z, q.Z);
    feL ZZ2 = fe_twice(ZZ);
    ge_p1p1 r;
    r.X = fe_sub(PP, MM);
    r.Y = fe_add(PP, MM);
    r.Z = fe_add_lt(ZZ2, TT);
    r.T = fe_sub_w(ZZ2, TT);
    return r;
}
// EdwardsPoint + EdwardsPoint (edwards.rs:795-800), 9 M
C25519_HD ge_p3 ge_add(const ge_p3 &a, const ge_p3 &b) { return ge_p1p1_to_p3(ge_add_cached(a, ge_p3_to_cached(b))); }
C25519_HD ge_p3 ge_neg(const ge_p3 &a) {
    ge_p3 r; r.X = fe_carry(fe_neg(a.X)); r.Y = a.Y; r.Z = a.Z; r.T = fe_carry(fe_neg(a.T)); return r;
}

// edwards.rs:501-511 ct_eq and traits.rs:45 is_identity
C25519_HD bool ge_eq(const ge_p3 &a, const ge_p3 &b) {
    return fe_eq(fe_mul(a.X, b.Z), fe_mul(b.X, a.Z)) & fe_eq(fe_mul(a.Y, b.Z), fe_mul(b.Y, a.Z));
}
C25519_HD bool ge_is_identity(const ge_p3 &a) { return fe_is_zero(a.X) & fe_eq(a.Y, a.Z); }

// CompressedEdwardsY::decompress, edwards.rs:211-258 on top of sqrt_ratio_i, field.rs:320-366.
// Returns validity; X,Y (Z = 1) in *out.  ZIP-215 rules: non-canonical y accepted, sign applied
// without checking x = 0.
C25519_HD bool fe_sqrt_ratio_i(feT &r_out, const feT &u, const feT &v) {
    feT v3 = fe_mul(fe_sq(v), v);
    feT v7 = fe_mul(fe_sq(v3), v);
    feT r = fe_mul(fe_mul(u, v3), fe_pow_p58(fe_mul(u, v7)));
    feT check = fe_mul(v, fe_sq(r));
    feT i = fe_sqrtm1();
    feT neg_u = fe_carry(fe_neg(u));
    bool correct = fe_eq(check, u), flipped = fe_eq(check, neg_u), flipped_i = fe_eq(check, fe_mul(neg_u, i));
    feT r_prime = fe_mul(i, r);
    r = fe_select(r, r_prime, flipped | flipped_i);
    r = fe_cneg(r, fe_is_negative(r) != 0);
    r_out = r;
    return correct | flipped;
}
C25519_HD bool ge_decompress(ge_p3 &out, const u32 w[8]) {
    feT Y = fe_from_words(w), Z = fe_one(), YY = fe_sq(Y);
    feT u = fe_carry(fe_sub(YY, Z));
    feT v = fe_carry(fe_add(fe_mul(YY, fe_d()), Z));
    feT X; bool ok = fe_sqrt_ratio_i(X, u, v);
    X = fe_cneg(X, (w[7] >> 31) != 0);
    out.X = X; out.Y = Y; out.Z = Z; out.T = fe_mul(X, Y);
    return ok;
}

// affine (x, y) -> compressed words, edwards/affine.rs:71-75
C25519_HD void ge_affine_compress(const feT &x, const feT &y, u32 w[8]) {
    fe_to_words(y, w);
    w[7] ^= fe_is_negative(x) << 31;
}


// ---- Ristretto (ristretto.rs:266-345 decompress, :500-533 compress, :822-829 equality) ----------
C25519_HD bool fe_invsqrt(feT &r, const feT &v) { return fe_sqrt_ratio_i(r, fe_one(), v); }
C25519_HD bool ris_decompress(ge_p3 &out, const u32 w[8]) {
    feT s = fe_from_words(w);
    u32 chk[8];
    fe_to_words(s, chk);
    u32 diff = 0;
    for (int i = 0; i < 8; i++) diff |= chk[i] ^ w[i];          // canonical iff re-encoding matches
    bool s_bad = (diff != 0) | ((chk[0] & 1u) != 0);             // ... and s non-negative
    feT one = fe_one(), ss = fe_sq(s);
    feT u1 = fe_carry(fe_sub(one, ss)), u2 = fe_carry(fe_add(one, ss)), u2_sqr = fe_sq(u2);
    feT neg_d_u1sq = fe_carry(fe_neg(fe_mul(fe_d(), fe_sq(u1))));
    feT v = fe_carry(fe_sub(neg_d_u1sq, u2_sqr));
    feT I;
    bool ok = fe_invsqrt(I, fe_mul(v, u2_sqr));
    feT Dx = fe_mul(I, u2), Dy = fe_mul(I, fe_mul(Dx, v));
    feT x = fe_mul(fe_add(s, s), Dx);
    x = fe_cneg(x, fe_is_negative(x) != 0);
    feT y = fe_mul(u1, Dy), t = fe_mul(x, y);
    out.X = x; out.Y = y; out.Z = one; out.T = t;
    return !s_bad & ok & (fe_is_negative(t) == 0) & !fe_is_zero(y);
}
C25519_HD void ris_compress(const ge_p3 &P, u32 w[8]) {
    feT X = P.X, Y = P.Y;
    feT u1 = fe_mul(fe_add(P.Z, P.Y), fe_sub(P.Z, P.Y)), u2 = fe_mul(X, Y);
    feT invsqrt;
    fe_invsqrt(invsqrt, fe_mul(u1, fe_sq(u2)));
    feT i1 = fe_mul(invsqrt, u1), i2 = fe_mul(invsqrt, u2);
    feT z_inv = fe_mul(i1, fe_mul(i2, P.T)), den_inv = i2;
    feT iX = fe_mul(X, fe_sqrtm1()), iY = fe_mul(Y, fe_sqrtm1());
    feT ench = fe_mul(i1, fe_invsqrt_a_minus_d());
    bool rotate = fe_is_negative(fe_mul(P.T, z_inv)) != 0;
    X = fe_select(X, iY, rotate); Y = fe_select(Y, iX, rotate); den_inv = fe_select(den_inv, ench, rotate);
    Y = fe_cneg(Y, fe_is_negative(fe_mul(X, z_inv)) != 0);
    feT s = fe_mul(fe_sub(P.Z, Y), den_inv);
    s = fe_cneg(s, fe_is_negative(s) != 0);
    fe_to_words(s, w);
}
C25519_HD bool ris_eq(const ge_p3 &a, const ge_p3 &b) {
    return fe_eq(fe_mul(a.X, b.Y), fe_mul(a.Y, b.X)) | fe_eq(fe_mul(a.X, b.X), fe_mul(a.Y, b.Y));
}

// ---- Montgomery ladder step, montgomery.rs:430-468 (5 M + 4 S + 1 small mul) --------------------
struct mont_pp { feT U, W; };
C25519_HD void mont_diff_add_and_double(mont_pp &P, mont_pp &Q, const feT &affine_PmQ) {
    feL t0 = fe_add(P.U, P.W), t1 = fe_sub(P.U, P.W), t2 = fe_add(Q.U, Q.W), t3 = fe_sub(Q.U, Q.W);
    feT t4 = fe_sq(t0), t5 = fe_sq(t1);
    feL t6 = fe_sub(t4, t5);
    feT t7 = fe_mul(t0, t3), t8 = fe_mul(t1, t2);
    feL t9 = fe_add(t7, t8), t10 = fe_sub(t7, t8);
    feT t11 = fe_sq(t9), t12 = fe_sq(t10);
    feT t13 = fe_mul_small(t6, 121666u);   // APLUS2_OVER_FOUR
    feT t14 = fe_mul(t4, t5);
    feL t15 = fe_add(t13, t5);
    feT t16 = fe_mul(t6, t15);
    feT t17 = fe_mul(affine_PmQ, t12);
    P.U = t14; P.W = t16; Q.U = t11; Q.W = t17;
}

}  // namespace c25519
