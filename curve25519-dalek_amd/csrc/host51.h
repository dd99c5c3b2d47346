// HOST-side point arithmetic for the one thing the host does in an MSM / verify_batch call: the Horner fold over the <= 56 window column
// sums (pippenger.rs:159: total.mul_by_pow_2(w) + column -- a serial chain of ~250 doublings, which a CPU core finishes faster than a GPU
// lane) and the additions of partial results across passes / contexts / ranks.
//
// Until round 4 the fold ran through ge26.h -- the DEVICE field layout (10 x u32, radix 2^25.5, products into u64) compiled for the host:
// 31 ns per field multiplication, 74 us per fold, 40 % of a 100-term call.  A 64-bit core has a 64 x 64 -> 128 multiplier, which is what the
// reference's FieldElement51 is designed for (u64/field.rs:43-52: five 51-bit limbs, 25 products per multiplication): this header is that
// layout for the host only -- ~11 ns per multiplication.  Values cross over from / to the device layout limb by limb
// (limb51[i] = limb26[2i] + 2^26 limb26[2i+1]).
//
// Bounds: every function returns limbs < 2^51 + 2^19 ("reduced") except h51_add (sum of two reduced: < 2^52 + 2^15); h51_mul / h51_sq accept
// limbs < 2^54 (25 products of < 2^54 * 19 * 2^54 = 2^112.3 each: five of them < 2^115 fit the 128-bit accumulator).
#pragma once
#include <stdint.h>
#include "ge26.h"

namespace c25519 {

typedef unsigned __int128 u128;
struct h51 { uint64_t v[5]; };
constexpr uint64_t H51_MASK = (1ull << 51) - 1;

static inline h51 h51_carry(const u128 t[5]) {
    h51 r;
    u128 c = t[0] >> 51; r.v[0] = (uint64_t)t[0] & H51_MASK;
    u128 x = t[1] + c; c = x >> 51; r.v[1] = (uint64_t)x & H51_MASK;
    x = t[2] + c; c = x >> 51; r.v[2] = (uint64_t)x & H51_MASK;
    x = t[3] + c; c = x >> 51; r.v[3] = (uint64_t)x & H51_MASK;
    x = t[4] + c; c = x >> 51; r.v[4] = (uint64_t)x & H51_MASK;
    const u128 f = (u128)r.v[0] + c * 19;                    // c < 2^65 for the admitted inputs: the wrap-around stays in 128 bits
    r.v[0] = (uint64_t)f & H51_MASK; r.v[1] += (uint64_t)(f >> 51);   // < 2^51 + 2^19
    return r;
}
static inline h51 h51_mul(const h51 &a, const h51 &b) {
    const uint64_t b1 = b.v[1] * 19, b2 = b.v[2] * 19, b3 = b.v[3] * 19, b4 = b.v[4] * 19;
    u128 t[5];
    t[0] = (u128)a.v[0] * b.v[0] + (u128)a.v[1] * b4 + (u128)a.v[2] * b3 + (u128)a.v[3] * b2 + (u128)a.v[4] * b1;
    t[1] = (u128)a.v[0] * b.v[1] + (u128)a.v[1] * b.v[0] + (u128)a.v[2] * b4 + (u128)a.v[3] * b3 + (u128)a.v[4] * b2;
    t[2] = (u128)a.v[0] * b.v[2] + (u128)a.v[1] * b.v[1] + (u128)a.v[2] * b.v[0] + (u128)a.v[3] * b4 + (u128)a.v[4] * b3;
    t[3] = (u128)a.v[0] * b.v[3] + (u128)a.v[1] * b.v[2] + (u128)a.v[2] * b.v[1] + (u128)a.v[3] * b.v[0] + (u128)a.v[4] * b4;
    t[4] = (u128)a.v[0] * b.v[4] + (u128)a.v[1] * b.v[3] + (u128)a.v[2] * b.v[2] + (u128)a.v[3] * b.v[1] + (u128)a.v[4] * b.v[0];
    return h51_carry(t);
}
static inline h51 h51_sq(const h51 &a) {
    const uint64_t a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], a4 = a.v[4];
    const uint64_t d0 = 2 * a0, d1 = 2 * a1, d2 = 2 * a2, a3_19 = 19 * a3, a4_19 = 19 * a4;
    u128 t[5];
    t[0] = (u128)a0 * a0 + (u128)d1 * a4_19 + (u128)d2 * a3_19;
    t[1] = (u128)d0 * a1 + (u128)d2 * a4_19 + (u128)a3 * a3_19;
    t[2] = (u128)d0 * a2 + (u128)a1 * a1 + (u128)(2 * a3) * a4_19;
    t[3] = (u128)d0 * a3 + (u128)d1 * a2 + (u128)a4 * a4_19;
    t[4] = (u128)d0 * a4 + (u128)d1 * a3 + (u128)a2 * a2;
    return h51_carry(t);
}
static inline h51 h51_add(const h51 &a, const h51 &b) { h51 r; for (int i = 0; i < 5; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
// a - b for a, b with limbs < 2^54: a + 16 p - b, then one carry pass
static inline h51 h51_sub(const h51 &a, const h51 &b) {
    u128 t[5];
    t[0] = (u128)(a.v[0] + 36028797018963664ull - b.v[0]);      // 16 * (2^51 - 19)
    for (int i = 1; i < 5; i++) t[i] = (u128)(a.v[i] + 36028797018963952ull - b.v[i]);   // 16 * (2^51 - 1)
    return h51_carry(t);
}
static inline h51 h51_twice(const h51 &a) { h51 r; for (int i = 0; i < 5; i++) r.v[i] = 2 * a.v[i]; return r; }
// device layout (10 limbs at bits 0, 26, 51, ...; tight) <-> host layout
static inline h51 h51_from_fe(const feT &f) { h51 r; for (int i = 0; i < 5; i++) r.v[i] = (uint64_t)f.v[2 * i] + ((uint64_t)f.v[2 * i + 1] << 26); return r; }
static inline feT h51_to_fe(const h51 &a) {
    u128 t[5];
    for (int i = 0; i < 5; i++) t[i] = a.v[i];
    const h51 c = h51_carry(t);                               // limbs < 2^51 + 2^19
    feW w;
    for (int i = 0; i < 5; i++) { w.v[2 * i] = (u32)c.v[i] & M26; w.v[2 * i + 1] = (u32)(c.v[i] >> 26); }
    return fe_carry(w);
}
static inline h51 h51_d2() { return h51_from_fe(fe_d2()); }

static inline h51 h51_pow2k(h51 a, int k) { for (int i = 0; i < k; i++) a = h51_sq(a); return a; }
// z^(p - 2) (field.rs:307-322 invert through :176-219 pow22501: 254 squarings + 11 multiplications); 0 -> 0
static inline h51 h51_invert(const h51 &z) {
    const h51 z2 = h51_sq(z), z9 = h51_mul(z, h51_pow2k(z2, 2)), z11 = h51_mul(z2, z9);
    const h51 t5 = h51_mul(z9, h51_sq(z11));                                  // 2^5 - 1
    const h51 t10 = h51_mul(h51_pow2k(t5, 5), t5), t20 = h51_mul(h51_pow2k(t10, 10), t10), t40 = h51_mul(h51_pow2k(t20, 20), t20);
    const h51 t50 = h51_mul(h51_pow2k(t40, 10), t10), t100 = h51_mul(h51_pow2k(t50, 50), t50), t200 = h51_mul(h51_pow2k(t100, 100), t100);
    const h51 t250 = h51_mul(h51_pow2k(t200, 50), t50);
    return h51_mul(h51_pow2k(t250, 5), z11);                                   // 2^255 - 21
}

struct hp3 { h51 X, Y, Z, T; };
static inline hp3 hp3_identity() { hp3 r; for (int i = 0; i < 5; i++) r.X.v[i] = r.Y.v[i] = r.Z.v[i] = r.T.v[i] = 0; r.Y.v[0] = 1; r.Z.v[0] = 1; return r; }
static inline hp3 hp3_from(const ge_p3 &p) { hp3 r; r.X = h51_from_fe(p.X); r.Y = h51_from_fe(p.Y); r.Z = h51_from_fe(p.Z); r.T = h51_from_fe(p.T); return r; }
static inline ge_p3 hp3_to(const hp3 &p) { ge_p3 r; r.X = h51_to_fe(p.X); r.Y = h51_to_fe(p.Y); r.Z = h51_to_fe(p.Z); r.T = h51_to_fe(p.T); return r; }
// EdwardsPoint + EdwardsPoint (edwards.rs:795-800 through curve_models.rs:411-429 and :365-373), 9 M
static inline hp3 hp3_add(const hp3 &p, const hp3 &q) {
    const h51 PP = h51_mul(h51_add(p.Y, p.X), h51_add(q.Y, q.X)), MM = h51_mul(h51_sub(p.Y, p.X), h51_sub(q.Y, q.X));
    const h51 TT = h51_mul(p.T, h51_mul(q.T, h51_d2())), ZZ2 = h51_twice(h51_mul(p.Z, q.Z));
    const h51 X = h51_sub(PP, MM), Y = h51_add(PP, MM), Z = h51_add(ZZ2, TT), T = h51_sub(ZZ2, TT);      // completed point (X : Y : Z : T)
    hp3 r;
    r.X = h51_mul(X, T); r.Y = h51_mul(Y, Z); r.Z = h51_mul(Z, T); r.T = h51_mul(X, Y);
    return r;
}
// k doublings (edwards.rs:1370-1380 mul_by_pow_2; curve_models.rs:381-397): the intermediate doublings skip T
static inline hp3 hp3_mul_by_pow_2(const hp3 &p, int k) {
    h51 X = p.X, Y = p.Y, Z = p.Z;
    hp3 r = p;
    for (int i = 0; i < k; i++) {
        const h51 XX = h51_sq(X), YY = h51_sq(Y), ZZ2 = h51_twice(h51_sq(Z)), S = h51_sq(h51_add(X, Y));
        const h51 YpX = h51_add(YY, XX), YmX = h51_sub(YY, XX);
        const h51 cX = h51_sub(S, YpX), cT = h51_sub(ZZ2, YmX);                                       // completed: (cX : YpX : YmX : cT)
        X = h51_mul(cX, cT); Y = h51_mul(YpX, YmX); Z = h51_mul(YmX, cT);
        if (i == k - 1) { r.X = X; r.Y = Y; r.Z = Z; r.T = h51_mul(cX, YpX); }
    }
    return r;
}

}  // namespace c25519
