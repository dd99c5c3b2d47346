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


// ---- (r6) the doubling chain of the fold on AVX-512 IFMA: four field operations per vector instruction --------------------------------------------------------
// The Horner fold is ~253 doublings whatever the window width -- 21 of the 28 us a small or mid-size call spends on the host after the GPU is through (DESIGN.md
// section 4), at ~85 ns per doubling with the 64 x 64 -> 128 multiplier.  A doubling is four independent squarings followed by four independent products
// (curve_models.rs:381-397 + :365-373) -- exactly the shape the reference's own vector backends exploit (docs/parallel-formulas.md; backend/vector/ifma: four field
// elements in the lanes of a vector, 52-bit multiply-adds).  Here: lane j of five 256-bit vectors holds field element j in radix 2^51; vpmadd52luq / vpmadd52huq
// give the low / high 52 bits of a 52 x 52 product, so a limb product a_i b_j contributes lo to position i + j and 2 hi to position i + j + 1 (2^52 = 2 * 2^51).
// Operands of a multiplication must be below 2^52 (the instruction truncates): every sum / difference is followed by a carry pass, which leaves limbs below
// 2^51 + 2^17.  Used when the CPU has avx512ifma + avx512vl (checked once); the scalar path above remains for every other host and is the reference in
// tests/test_fe26_host.py::test_ifma_doubling_chain.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(C25519_NO_IFMA)
}  // namespace c25519
#include <immintrin.h>
namespace c25519 {
#define C25519_IFMA_FN __attribute__((target("avx512ifma,avx512vl,avx2"))) static inline
struct f51x4 { __m256i v[5]; };
C25519_IFMA_FN __m256i ifma_mask51() { return _mm256_set1_epi64x((long long)H51_MASK); }
// limbs below 2^63 -> limbs below 2^51 + 2^17 (one pass; the wrap-around of the top carry times 19 goes to limb 0 and is not propagated further)
C25519_IFMA_FN f51x4 f51x4_carry(f51x4 a) {
    const __m256i m = ifma_mask51();
    __m256i c = _mm256_srli_epi64(a.v[0], 51); a.v[0] = _mm256_and_si256(a.v[0], m);
    a.v[1] = _mm256_add_epi64(a.v[1], c); c = _mm256_srli_epi64(a.v[1], 51); a.v[1] = _mm256_and_si256(a.v[1], m);
    a.v[2] = _mm256_add_epi64(a.v[2], c); c = _mm256_srli_epi64(a.v[2], 51); a.v[2] = _mm256_and_si256(a.v[2], m);
    a.v[3] = _mm256_add_epi64(a.v[3], c); c = _mm256_srli_epi64(a.v[3], 51); a.v[3] = _mm256_and_si256(a.v[3], m);
    a.v[4] = _mm256_add_epi64(a.v[4], c); c = _mm256_srli_epi64(a.v[4], 51); a.v[4] = _mm256_and_si256(a.v[4], m);
    const __m256i c19 = _mm256_add_epi64(_mm256_add_epi64(_mm256_slli_epi64(c, 4), _mm256_slli_epi64(c, 1)), c);      // 19 c, c < 2^12
    a.v[0] = _mm256_add_epi64(a.v[0], c19);
    return a;
}
// column sums t[0..9] (t[k] = sum of lo at k + 2 x sum of hi at k - 1) -> reduced limbs
C25519_IFMA_FN f51x4 f51x4_reduce(const __m256i zl[9], const __m256i zh[9]) {
    __m256i t[10];
    t[0] = zl[0];
    _Pragma("GCC unroll 16") for (int k = 1; k < 9; k++) t[k] = _mm256_add_epi64(zl[k], _mm256_slli_epi64(zh[k - 1], 1));
    t[9] = _mm256_slli_epi64(zh[8], 1);
    f51x4 r;
    _Pragma("GCC unroll 16") for (int k = 0; k < 5; k++) {
        const __m256i h = t[k + 5];                                       // < 2^57: 19 h < 2^62
        r.v[k] = _mm256_add_epi64(t[k], _mm256_add_epi64(_mm256_add_epi64(_mm256_slli_epi64(h, 4), _mm256_slli_epi64(h, 1)), h));
    }
    return f51x4_carry(r);
}
// lane-wise product; limbs of a, b below 2^52
C25519_IFMA_FN f51x4 f51x4_mul(const f51x4 &a, const f51x4 &b) {
    __m256i zl[9], zh[9];
    const __m256i z = _mm256_setzero_si256();
    _Pragma("GCC unroll 16") for (int k = 0; k < 9; k++) { zl[k] = z; zh[k] = z; }
    _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++)
        _Pragma("GCC unroll 16") for (int j = 0; j < 5; j++) {
            zl[i + j] = _mm256_madd52lo_epu64(zl[i + j], a.v[i], b.v[j]);
            zh[i + j] = _mm256_madd52hi_epu64(zh[i + j], a.v[i], b.v[j]);
        }
    return f51x4_reduce(zl, zh);
}
// lane-wise square: the cross products once, doubled in the accumulator (2 a_j would not fit the 52-bit operand)
C25519_IFMA_FN f51x4 f51x4_sq(const f51x4 &a) {
    __m256i dl[9], dh[9], xl[9], xh[9];
    const __m256i z = _mm256_setzero_si256();
    _Pragma("GCC unroll 16") for (int k = 0; k < 9; k++) { dl[k] = z; dh[k] = z; xl[k] = z; xh[k] = z; }
    _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
        dl[2 * i] = _mm256_madd52lo_epu64(dl[2 * i], a.v[i], a.v[i]);
        dh[2 * i] = _mm256_madd52hi_epu64(dh[2 * i], a.v[i], a.v[i]);
        _Pragma("GCC unroll 16") for (int j = i + 1; j < 5; j++) {
            xl[i + j] = _mm256_madd52lo_epu64(xl[i + j], a.v[i], a.v[j]);
            xh[i + j] = _mm256_madd52hi_epu64(xh[i + j], a.v[i], a.v[j]);
        }
    }
    _Pragma("GCC unroll 16") for (int k = 0; k < 9; k++) { dl[k] = _mm256_add_epi64(dl[k], _mm256_slli_epi64(xl[k], 1)); dh[k] = _mm256_add_epi64(dh[k], _mm256_slli_epi64(xh[k], 1)); }
    return f51x4_reduce(dl, dh);
}
C25519_IFMA_FN f51x4 f51x4_load(const h51 &a, const h51 &b, const h51 &c, const h51 &d) {
    f51x4 r;
    _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) r.v[i] = _mm256_set_epi64x((long long)d.v[i], (long long)c.v[i], (long long)b.v[i], (long long)a.v[i]);
    return r;
}
C25519_IFMA_FN void f51x4_store(const f51x4 &a, h51 out[4]) {
    _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
        alignas(32) uint64_t t[4];
        _mm256_store_si256((__m256i *)t, a.v[i]);
        _Pragma("GCC unroll 16") for (int j = 0; j < 4; j++) out[j].v[i] = t[j];
    }
}
// 2 p in radix 2^51 (limbs 2^52 - 38, 2^52 - 2, ...): a + 2p - b for limbs of b below 2^52
C25519_IFMA_FN __m256i ifma_2p(int limb) { return _mm256_set1_epi64x(limb == 0 ? (long long)((1ull << 52) - 38) : (long long)((1ull << 52) - 2)); }
// k doublings of (X : Y : Z), T of the last one: hp3_mul_by_pow_2 with the four squarings and the four products of every doubling in the lanes of one vector
C25519_IFMA_FN hp3 hp3_mul_by_pow_2_ifma(const hp3 &p, int k) {
    if (k <= 0) return p;
    const h51 xy = h51_add(p.X, p.Y);
    f51x4 A = f51x4_carry(f51x4_load(p.X, p.Y, p.Z, xy));                 // (X, Y, Z, X + Y), limbs below 2^52
    f51x4 P = A;
    _Pragma("GCC unroll 16") for (int it = 0; it < k; it++) {
        const f51x4 Q = f51x4_sq(A);                                      // (XX, YY, ZZ, S)
        f51x4 W, U, M1, M2;
        _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
            // W = (YY + XX, YY + 2p - XX, 2 ZZ, S)
            const __m256i a0 = _mm256_permute4x64_epi64(Q.v[i], 0xE5);   // lanes (1, 1, 2, 3) = (YY, YY, ZZ, S)
            const __m256i b0 = _mm256_permute4x64_epi64(Q.v[i], 0xA0);   // lanes (0, 0, 2, 2) = (XX, XX, ZZ, ZZ)
            const __m256i neg = _mm256_sub_epi64(ifma_2p(i), b0);          // 2p - XX in lane 1
            __m256i add = _mm256_blend_epi32(b0, neg, 0x0C);             // lane 1 <- 2p - XX
            add = _mm256_blend_epi32(add, _mm256_setzero_si256(), 0xC0); // lane 3 <- 0
            W.v[i] = _mm256_add_epi64(a0, add);
        }
        W = f51x4_carry(W);                                               // (YpX, YmX, ZZ2, S)
        _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
            // U = (S + 2p - YpX, ZZ2 + 2p - YmX, ., .) = (cX, cT, ., .)
            const __m256i a1 = _mm256_permute4x64_epi64(W.v[i], 0x0B);   // lanes (3, 2, 0, 0) = (S, ZZ2, ., .)
            const __m256i b1 = _mm256_permute4x64_epi64(W.v[i], 0x04);   // lanes (0, 1, 0, 0) = (YpX, YmX, ., .)
            U.v[i] = _mm256_add_epi64(a1, _mm256_sub_epi64(ifma_2p(i), b1));
        }
        U = f51x4_carry(U);                                               // (cX, cT, ., .)
        _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
            const __m256i cx = _mm256_permute4x64_epi64(U.v[i], 0x00), ct = _mm256_permute4x64_epi64(U.v[i], 0x55);
            const __m256i w0110 = _mm256_permute4x64_epi64(W.v[i], 0x10);      // lanes (0, 0, 1, 0) = (., YpX, YmX, .)  [lane 1 = W0, lane 2 = W1]
            const __m256i w1x0 = _mm256_permute4x64_epi64(W.v[i], 0x04);       // lanes (0, 1, 0, 0) = (., YmX, ., YpX)  [lane 1 = W1, lane 3 = W0]
            M1.v[i] = _mm256_blend_epi32(cx, w0110, 0x3C);                // (cX, YpX, YmX, cX)
            M2.v[i] = _mm256_blend_epi32(ct, w1x0, 0xCC);                 // (cT, YmX, cT, YpX)
        }
        P = f51x4_mul(M1, M2);                                            // (X', Y', Z', T') = (cX cT, YpX YmX, YmX cT, cX YpX)
        if (it + 1 < k) {
            _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
                const __m256i yx = _mm256_permute4x64_epi64(P.v[i], 0x55);     // Y' everywhere
                const __m256i sum = _mm256_add_epi64(_mm256_permute4x64_epi64(P.v[i], 0x00), yx);      // X' + Y'
                A.v[i] = _mm256_blend_epi32(P.v[i], sum, 0xC0);           // (X', Y', Z', X' + Y')
            }
            A = f51x4_carry(A);
        }
    }
    h51 o[4];
    f51x4_store(P, o);
    hp3 r; r.X = o[0]; r.Y = o[1]; r.Z = o[2]; r.T = o[3];
    return r;
}
// ---- (r6, late) the WHOLE Horner fold in lane form: the running total stays (X, Y, Z, T) in the lanes of five vectors across doublings and additions ----------
// k doublings of P = (X, Y, Z, T) in lane form (the loop of hp3_mul_by_pow_2_ifma without the conversions around it)
C25519_IFMA_FN f51x4 p4_pow2(f51x4 P, int k) {
    _Pragma("GCC unroll 16") for (int it = 0; it < k; it++) {
        f51x4 A;
        _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
            const __m256i yx = _mm256_permute4x64_epi64(P.v[i], 0x55);
            const __m256i sum = _mm256_add_epi64(_mm256_permute4x64_epi64(P.v[i], 0x00), yx);
            A.v[i] = _mm256_blend_epi32(P.v[i], sum, 0xC0);                // (X, Y, Z, X + Y)
        }
        A = f51x4_carry(A);
        const f51x4 Q = f51x4_sq(A);                                      // (XX, YY, ZZ, S)
        f51x4 W, U, M1, M2;
        _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
            const __m256i a0 = _mm256_permute4x64_epi64(Q.v[i], 0xE5), b0 = _mm256_permute4x64_epi64(Q.v[i], 0xA0);
            const __m256i neg = _mm256_sub_epi64(ifma_2p(i), b0);
            __m256i add = _mm256_blend_epi32(b0, neg, 0x0C);
            add = _mm256_blend_epi32(add, _mm256_setzero_si256(), 0xC0);
            W.v[i] = _mm256_add_epi64(a0, add);
        }
        W = f51x4_carry(W);                                               // (YpX, YmX, ZZ2, S)
        _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
            const __m256i a1 = _mm256_permute4x64_epi64(W.v[i], 0x0B), b1 = _mm256_permute4x64_epi64(W.v[i], 0x04);
            U.v[i] = _mm256_add_epi64(a1, _mm256_sub_epi64(ifma_2p(i), b1));
        }
        U = f51x4_carry(U);                                               // (cX, cT, ., .)
        _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
            const __m256i cx = _mm256_permute4x64_epi64(U.v[i], 0x00), ct = _mm256_permute4x64_epi64(U.v[i], 0x55);
            const __m256i w0110 = _mm256_permute4x64_epi64(W.v[i], 0x10), w1x0 = _mm256_permute4x64_epi64(W.v[i], 0x04);
            M1.v[i] = _mm256_blend_epi32(cx, w0110, 0x3C);                // (cX, YpX, YmX, cX)
            M2.v[i] = _mm256_blend_epi32(ct, w1x0, 0xCC);                 // (cT, YmX, cT, YpX)
        }
        P = f51x4_mul(M1, M2);                                            // (X', Y', Z', T')
    }
    return P;
}
// the cached form of a point for p4_add: (Y - X, Y + X, 2 Z, 2 d T) in the lanes (curve_models.rs:365-373 / the reference's CachedPoint, backend/vector/ifma)
C25519_IFMA_FN f51x4 p4_cached(const hp3 &q) {
    const h51 d2 = h51_d2(), one = {{1, 0, 0, 0, 0}};
    f51x4 Q = f51x4_load(h51_sub(q.Y, q.X), h51_add(q.Y, q.X), h51_twice(q.Z), q.T);
    return f51x4_mul(f51x4_carry(Q), f51x4_load(one, one, one, d2));
}
// P + Q (complete addition, edwards.rs:795-800 through curve_models.rs:411-429 and :365-373; the same 9 M as hp3_add) with the four products of each of its two stages
// in the lanes of one vector
C25519_IFMA_FN f51x4 p4_add(const f51x4 &P, const f51x4 &Qc) {
    f51x4 D1;
    _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
        const __m256i a = _mm256_permute4x64_epi64(P.v[i], 0xE5);         // (Y, Y, Z, T)
        const __m256i b = _mm256_permute4x64_epi64(P.v[i], 0xA0);         // (X, X, Z, Z)
        const __m256i neg = _mm256_sub_epi64(ifma_2p(i), b);
        __m256i add = _mm256_blend_epi32(b, neg, 0x03);                   // lane 0 <- 2p - X, lane 1 = X
        add = _mm256_blend_epi32(add, _mm256_setzero_si256(), 0xF0);      // lanes 2, 3 <- 0
        D1.v[i] = _mm256_add_epi64(a, add);                               // (Y - X, Y + X, Z, T)
    }
    const f51x4 M = f51x4_mul(f51x4_carry(D1), Qc);                       // (A, B, D, C) = ((Y1-X1)(Y2-X2), (Y1+X1)(Y2+X2), 2 Z1 Z2, 2d T1 T2)
    f51x4 S, F;
    _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
        const __m256i sw = _mm256_permute4x64_epi64(M.v[i], 0xB1);        // (B, A, C, D)
        S.v[i] = _mm256_add_epi64(M.v[i], sw);                            // (H, H, G, G): H = B + A, G = D + C
        F.v[i] = _mm256_add_epi64(sw, _mm256_sub_epi64(ifma_2p(i), M.v[i]));      // (E, -E, -F, F): E = B - A, F = D - C
    }
    S = f51x4_carry(S); F = f51x4_carry(F);
    f51x4 L, R;
    _Pragma("GCC unroll 16") for (int i = 0; i < 5; i++) {
        const __m256i dl = _mm256_permute4x64_epi64(F.v[i], 0x30), sl = _mm256_permute4x64_epi64(S.v[i], 0xAA);
        L.v[i] = _mm256_blend_epi32(dl, sl, 0x0C);                        // (E, G, F, E)
        const __m256i dr = _mm256_permute4x64_epi64(F.v[i], 0xFF), sr = _mm256_permute4x64_epi64(S.v[i], 0x20);
        R.v[i] = _mm256_blend_epi32(sr, dr, 0x03);                        // (F, H, G, H)
    }
    return f51x4_mul(L, R);                                               // (E F, G H, F G, E H) = (X3, Y3, Z3, T3)
}
// sum_k 2^(pos_k) col_k by Horner, top window first: shift[k] = doublings in front of the addition of column k (shift of the first column handled: ignored)
C25519_IFMA_FN hp3 hp3_horner_ifma(const hp3 *cols, const int *shift, int n) {
    const hp3 id = hp3_identity();
    f51x4 P = f51x4_load(id.X, id.Y, id.Z, id.T);
    for (int k = 0; k < n; k++) {
        if (k) P = p4_pow2(P, shift[k]);
        P = p4_add(P, p4_cached(cols[k]));
    }
    h51 o[4];
    f51x4_store(P, o);
    hp3 r; r.X = o[0]; r.Y = o[1]; r.Z = o[2]; r.T = o[3];
    return r;
}
static inline bool host_has_ifma() {
    static const bool v = __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512vl");
    return v;
}
// the doubling chain of the fold: IFMA where the host has it
static inline hp3 hp3_pow2_fast(const hp3 &p, int k) { return host_has_ifma() ? hp3_mul_by_pow_2_ifma(p, k) : hp3_mul_by_pow_2(p, k); }
#else
static inline bool host_has_ifma() { return false; }
static inline hp3 hp3_pow2_fast(const hp3 &p, int k) { return hp3_mul_by_pow_2(p, k); }
#endif
// the fold: columns top window first, shift[k] doublings in front of column k (k >= 1)
static inline hp3 hp3_horner(const hp3 *cols, const int *shift, int n) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(C25519_NO_IFMA)
    if (host_has_ifma()) return hp3_horner_ifma(cols, shift, n);
#endif
    hp3 total = hp3_identity();
    for (int k = 0; k < n; k++) {
        if (k) total = hp3_mul_by_pow_2(total, shift[k]);
        total = hp3_add(total, cols[k]);
    }
    return total;
}

}  // namespace c25519
