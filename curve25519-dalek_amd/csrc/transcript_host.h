// Host-side restatement of the reference's batch-verification transcript
// (ed25519-dalek/src/batch.rs:168-222 over batch/transcript.rs:39-207): Merlin framing on
// STROBE-128/1600 (strobe-rs 0.13.0 / keccak 0.2.0 in the reference's Cargo.lock; restated from the
// public STROBE v1.0.2 and FIPS 202 specifications).  It is inherently sequential (one sponge absorbs
// 2n messages and is then squeezed n times), so z_mode 0 runs it on one host core; z_mode 1 replaces
// it with a parallel on-device derivation.
#pragma once
#include <stdint.h>
#include <string.h>
#include "constants_gen.h"
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

namespace c25519_tr {

static inline uint64_t rotl(uint64_t x, unsigned n) { return n ? (x << n) | (x >> (64 - n)) : x; }
// Keccak-f[1600] (FIPS 202 section 3.2-3.4) with the 25 lanes in locals and every index a constant: theta, then rho and pi
// as one renaming, then chi row by row.  The loop form of the same round costs 1.7 x as much on the host core that the
// whole-batch transcript is serialised on.
#define C25519_K_ROUND(rc)                                                                                            \
    {                                                                                                                 \
        const uint64_t c0 = a0 ^ a5 ^ a10 ^ a15 ^ a20, c1 = a1 ^ a6 ^ a11 ^ a16 ^ a21, c2 = a2 ^ a7 ^ a12 ^ a17 ^ a22, \
                       c3 = a3 ^ a8 ^ a13 ^ a18 ^ a23, c4 = a4 ^ a9 ^ a14 ^ a19 ^ a24;                               \
        const uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1), d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1); \
        /* b[y + 5 ((2x + 3y) mod 5)] = rotl(a[x + 5y] ^ d[x], ROT[x + 5y]) */                                        \
        const uint64_t b0 = a0 ^ d0, b1 = rotl(a6 ^ d1, 44), b2 = rotl(a12 ^ d2, 43), b3 = rotl(a18 ^ d3, 21), b4 = rotl(a24 ^ d4, 14);            \
        const uint64_t b5 = rotl(a3 ^ d3, 28), b6 = rotl(a9 ^ d4, 20), b7 = rotl(a10 ^ d0, 3), b8 = rotl(a16 ^ d1, 45), b9 = rotl(a22 ^ d2, 61);   \
        const uint64_t b10 = rotl(a1 ^ d1, 1), b11 = rotl(a7 ^ d2, 6), b12 = rotl(a13 ^ d3, 25), b13 = rotl(a19 ^ d4, 8), b14 = rotl(a20 ^ d0, 18); \
        const uint64_t b15 = rotl(a4 ^ d4, 27), b16 = rotl(a5 ^ d0, 36), b17 = rotl(a11 ^ d1, 10), b18 = rotl(a17 ^ d2, 15), b19 = rotl(a23 ^ d3, 56); \
        const uint64_t b20 = rotl(a2 ^ d2, 62), b21 = rotl(a8 ^ d3, 55), b22 = rotl(a14 ^ d4, 39), b23 = rotl(a15 ^ d0, 41), b24 = rotl(a21 ^ d1, 2); \
        a0 = b0 ^ (~b1 & b2) ^ (rc); a1 = b1 ^ (~b2 & b3); a2 = b2 ^ (~b3 & b4); a3 = b3 ^ (~b4 & b0); a4 = b4 ^ (~b0 & b1);                     \
        a5 = b5 ^ (~b6 & b7); a6 = b6 ^ (~b7 & b8); a7 = b7 ^ (~b8 & b9); a8 = b8 ^ (~b9 & b5); a9 = b9 ^ (~b5 & b6);                             \
        a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11);         \
        a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16);         \
        a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21);         \
    }
#define C25519_KECCAK_BODY                                                                                                                            \
    static const uint64_t RC[24] = C25519_KECCAK_RC;                                                                                                  \
    uint64_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6], a7 = a[7], a8 = a[8], a9 = a[9], a10 = a[10], a11 = a[11],  \
             a12 = a[12], a13 = a[13], a14 = a[14], a15 = a[15], a16 = a[16], a17 = a[17], a18 = a[18], a19 = a[19], a20 = a[20], a21 = a[21],        \
             a22 = a[22], a23 = a[23], a24 = a[24];                                                                                                    \
    for (int rnd = 0; rnd < 24; rnd += 2) { C25519_K_ROUND(RC[rnd]) C25519_K_ROUND(RC[rnd + 1]) }                                                     \
    a[0] = a0; a[1] = a1; a[2] = a2; a[3] = a3; a[4] = a4; a[5] = a5; a[6] = a6; a[7] = a7; a[8] = a8; a[9] = a9; a[10] = a10; a[11] = a11; a[12] = a12; \
    a[13] = a13; a[14] = a14; a[15] = a15; a[16] = a16; a[17] = a17; a[18] = a18; a[19] = a19; a[20] = a20; a[21] = a21; a[22] = a22; a[23] = a23; a[24] = a24;
static void keccak_f_generic(uint64_t a[25]) { C25519_KECCAK_BODY }
// (r4) The strict z-mode of verify_batch is ONE sequential sponge on one host core (batch.rs:195-222: ~1.7 permutations per signature), so the
// permutation IS the reference-exact mode's throughput.  The library is built for a generic x86-64; the same source compiled for BMI / BMI2 turns
// every chi term (~b & c) into one ANDN and every rotation into one RORX without a flags dependency.  Chosen once per process from CPUID; the
// bytes are the same (the known-answer tests under tests/ pin them against the independent STROBE of tests/pyref.py on whichever path the host takes).
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("bmi,bmi2"))) static void keccak_f_bmi2(uint64_t a[25]) { C25519_KECCAK_BODY }
// (r6, last) The permutation on 128-bit LANE PAIRS (AVX-512VL: VPTERNLOGQ, VPROLVQ and 32 xmm registers).  Register r[x][g] holds column x's lanes of rows {0, 1}, {2, 3} or
// {4, -} (upper half zero): 15 registers, no spills.  theta: one three-way XOR per column gives the two half-parities, a half swap + XOR broadcasts the column parity, and
// D[x] applies to both lanes of a register alike; rho: one VPROLVQ per register; pi moves row y's lanes to column y, so a new pair is two old registers' same halves --
// ONE in-lane unpack per register (or a move / byte shift for the single lane), never a cross-lane permute; chi: the three operands of a term sit in the same half of the
// registers of columns X, X + 1, X + 2 -- one VPTERNLOGQ (0xD2 = a ^ (~b & c)) per register.  86 vector instructions per round of which 20 in-lane shuffles, against 130
// scalar operations + ~100 moves: 156 against 178 ns on the GPU box's host (EPYC 9575F, Zen 5), 238 against 352 on a Xeon (profiles/r06_keccak_avx512.txt; the
// plane-per-zmm forms measured beside it lose to the scalar code on Zen 5: docs/lab/).  Same bytes as the scalar forms (tests/test_fe26_host.py compares every form the
// host can run with the spec-level permutation of tests/pyref.py; the transcript tests run on whichever keccak_pick() takes).
#define C25519_KP_CONSTANTS \
    static const uint64_t RC[24] = C25519_KECCAK_RC; \
    alignas(16) static const uint64_t RH[5][3][2] = {{{0, 36}, {3, 41}, {18, 0}}, {{1, 44}, {10, 45}, {2, 0}}, {{62, 6}, {43, 15}, {61, 0}}, {{28, 55}, {25, 21}, {56, 0}}, {{27, 20}, {39, 8}, {14, 0}}};
#define C25519_KP_LOAD(a) \
    __m128i r00 = _mm_set_epi64x((long long)a[5], (long long)a[0]), r01 = _mm_set_epi64x((long long)a[15], (long long)a[10]), r02 = _mm_loadl_epi64((const __m128i *)(a + 20)); \
    __m128i r10 = _mm_set_epi64x((long long)a[6], (long long)a[1]), r11 = _mm_set_epi64x((long long)a[16], (long long)a[11]), r12 = _mm_loadl_epi64((const __m128i *)(a + 21)); \
    __m128i r20 = _mm_set_epi64x((long long)a[7], (long long)a[2]), r21 = _mm_set_epi64x((long long)a[17], (long long)a[12]), r22 = _mm_loadl_epi64((const __m128i *)(a + 22)); \
    __m128i r30 = _mm_set_epi64x((long long)a[8], (long long)a[3]), r31 = _mm_set_epi64x((long long)a[18], (long long)a[13]), r32 = _mm_loadl_epi64((const __m128i *)(a + 23)); \
    __m128i r40 = _mm_set_epi64x((long long)a[9], (long long)a[4]), r41 = _mm_set_epi64x((long long)a[19], (long long)a[14]), r42 = _mm_loadl_epi64((const __m128i *)(a + 24));
#define C25519_KP_ROUNDS \
    for (int rnd = 0; rnd < 24; rnd++) { \
        __m128i c0 = _mm_ternarylogic_epi64(r00, r01, r02, 0x96); c0 = _mm_xor_si128(c0, _mm_shuffle_epi32(c0, 0x4E)); \
        __m128i c1 = _mm_ternarylogic_epi64(r10, r11, r12, 0x96); c1 = _mm_xor_si128(c1, _mm_shuffle_epi32(c1, 0x4E)); \
        __m128i c2 = _mm_ternarylogic_epi64(r20, r21, r22, 0x96); c2 = _mm_xor_si128(c2, _mm_shuffle_epi32(c2, 0x4E)); \
        __m128i c3 = _mm_ternarylogic_epi64(r30, r31, r32, 0x96); c3 = _mm_xor_si128(c3, _mm_shuffle_epi32(c3, 0x4E)); \
        __m128i c4 = _mm_ternarylogic_epi64(r40, r41, r42, 0x96); c4 = _mm_xor_si128(c4, _mm_shuffle_epi32(c4, 0x4E)); \
        const __m128i d0 = _mm_xor_si128(c4, _mm_rol_epi64(c1, 1)); \
        const __m128i d1 = _mm_xor_si128(c0, _mm_rol_epi64(c2, 1)); \
        const __m128i d2 = _mm_xor_si128(c1, _mm_rol_epi64(c3, 1)); \
        const __m128i d3 = _mm_xor_si128(c2, _mm_rol_epi64(c4, 1)); \
        const __m128i d4 = _mm_xor_si128(c3, _mm_rol_epi64(c0, 1)); \
        const __m128i t00 = _mm_rolv_epi64(_mm_xor_si128(r00, d0), _mm_load_si128((const __m128i *)RH[0][0])); \
        const __m128i t01 = _mm_rolv_epi64(_mm_xor_si128(r01, d0), _mm_load_si128((const __m128i *)RH[0][1])); \
        const __m128i t02 = _mm_rolv_epi64(_mm_xor_si128(r02, d0), _mm_load_si128((const __m128i *)RH[0][2])); \
        const __m128i t10 = _mm_rolv_epi64(_mm_xor_si128(r10, d1), _mm_load_si128((const __m128i *)RH[1][0])); \
        const __m128i t11 = _mm_rolv_epi64(_mm_xor_si128(r11, d1), _mm_load_si128((const __m128i *)RH[1][1])); \
        const __m128i t12 = _mm_rolv_epi64(_mm_xor_si128(r12, d1), _mm_load_si128((const __m128i *)RH[1][2])); \
        const __m128i t20 = _mm_rolv_epi64(_mm_xor_si128(r20, d2), _mm_load_si128((const __m128i *)RH[2][0])); \
        const __m128i t21 = _mm_rolv_epi64(_mm_xor_si128(r21, d2), _mm_load_si128((const __m128i *)RH[2][1])); \
        const __m128i t22 = _mm_rolv_epi64(_mm_xor_si128(r22, d2), _mm_load_si128((const __m128i *)RH[2][2])); \
        const __m128i t30 = _mm_rolv_epi64(_mm_xor_si128(r30, d3), _mm_load_si128((const __m128i *)RH[3][0])); \
        const __m128i t31 = _mm_rolv_epi64(_mm_xor_si128(r31, d3), _mm_load_si128((const __m128i *)RH[3][1])); \
        const __m128i t32 = _mm_rolv_epi64(_mm_xor_si128(r32, d3), _mm_load_si128((const __m128i *)RH[3][2])); \
        const __m128i t40 = _mm_rolv_epi64(_mm_xor_si128(r40, d4), _mm_load_si128((const __m128i *)RH[4][0])); \
        const __m128i t41 = _mm_rolv_epi64(_mm_xor_si128(r41, d4), _mm_load_si128((const __m128i *)RH[4][1])); \
        const __m128i t42 = _mm_rolv_epi64(_mm_xor_si128(r42, d4), _mm_load_si128((const __m128i *)RH[4][2])); \
        const __m128i n00 = _mm_unpacklo_epi64(t00, t30), n01 = _mm_unpacklo_epi64(t10, t40), n02 = _mm_move_epi64(t20); \
        const __m128i n10 = _mm_unpackhi_epi64(t10, t40), n11 = _mm_unpackhi_epi64(t20, t00), n12 = _mm_srli_si128(t30, 8); \
        const __m128i n20 = _mm_unpacklo_epi64(t21, t01), n21 = _mm_unpacklo_epi64(t31, t11), n22 = _mm_move_epi64(t41); \
        const __m128i n30 = _mm_unpackhi_epi64(t31, t11), n31 = _mm_unpackhi_epi64(t41, t21), n32 = _mm_srli_si128(t01, 8); \
        const __m128i n40 = _mm_unpacklo_epi64(t42, t22), n41 = _mm_unpacklo_epi64(t02, t32), n42 = _mm_move_epi64(t12); \
        r00 = _mm_ternarylogic_epi64(n00, n10, n20, 0xD2); \
        r01 = _mm_ternarylogic_epi64(n01, n11, n21, 0xD2); \
        r02 = _mm_ternarylogic_epi64(n02, n12, n22, 0xD2); \
        r10 = _mm_ternarylogic_epi64(n10, n20, n30, 0xD2); \
        r11 = _mm_ternarylogic_epi64(n11, n21, n31, 0xD2); \
        r12 = _mm_ternarylogic_epi64(n12, n22, n32, 0xD2); \
        r20 = _mm_ternarylogic_epi64(n20, n30, n40, 0xD2); \
        r21 = _mm_ternarylogic_epi64(n21, n31, n41, 0xD2); \
        r22 = _mm_ternarylogic_epi64(n22, n32, n42, 0xD2); \
        r30 = _mm_ternarylogic_epi64(n30, n40, n00, 0xD2); \
        r31 = _mm_ternarylogic_epi64(n31, n41, n01, 0xD2); \
        r32 = _mm_ternarylogic_epi64(n32, n42, n02, 0xD2); \
        r40 = _mm_ternarylogic_epi64(n40, n00, n10, 0xD2); \
        r41 = _mm_ternarylogic_epi64(n41, n01, n11, 0xD2); \
        r42 = _mm_ternarylogic_epi64(n42, n02, n12, 0xD2); \
        r00 = _mm_xor_si128(r00, _mm_loadl_epi64((const __m128i *)(RC + rnd))); \
    }
#define C25519_KP_STORE(a) \
    a[0] = (uint64_t)_mm_cvtsi128_si64(r00); a[5] = (uint64_t)_mm_extract_epi64(r00, 1); a[10] = (uint64_t)_mm_cvtsi128_si64(r01); a[15] = (uint64_t)_mm_extract_epi64(r01, 1); a[20] = (uint64_t)_mm_cvtsi128_si64(r02); \
    a[1] = (uint64_t)_mm_cvtsi128_si64(r10); a[6] = (uint64_t)_mm_extract_epi64(r10, 1); a[11] = (uint64_t)_mm_cvtsi128_si64(r11); a[16] = (uint64_t)_mm_extract_epi64(r11, 1); a[21] = (uint64_t)_mm_cvtsi128_si64(r12); \
    a[2] = (uint64_t)_mm_cvtsi128_si64(r20); a[7] = (uint64_t)_mm_extract_epi64(r20, 1); a[12] = (uint64_t)_mm_cvtsi128_si64(r21); a[17] = (uint64_t)_mm_extract_epi64(r21, 1); a[22] = (uint64_t)_mm_cvtsi128_si64(r22); \
    a[3] = (uint64_t)_mm_cvtsi128_si64(r30); a[8] = (uint64_t)_mm_extract_epi64(r30, 1); a[13] = (uint64_t)_mm_cvtsi128_si64(r31); a[18] = (uint64_t)_mm_extract_epi64(r31, 1); a[23] = (uint64_t)_mm_cvtsi128_si64(r32); \
    a[4] = (uint64_t)_mm_cvtsi128_si64(r40); a[9] = (uint64_t)_mm_extract_epi64(r40, 1); a[14] = (uint64_t)_mm_cvtsi128_si64(r41); a[19] = (uint64_t)_mm_extract_epi64(r41, 1); a[24] = (uint64_t)_mm_cvtsi128_si64(r42);
__attribute__((target("avx512f,avx512vl"))) static void keccak_f_pairs(uint64_t a[25]) { C25519_KP_CONSTANTS C25519_KP_LOAD(a) C25519_KP_ROUNDS C25519_KP_STORE(a) }
// The z squeeze of the batch transcript with the state staying in the 15 registers from one z to the next (transcript.rs:200-206: meta_ad(len16) + prf(16) per z; every z
// after the first finds the sponge 16 bytes into a fresh block, so an iteration is: header / length / padding bytes XORed into lanes 2, 3 and 20, the permutation, lanes 0
// and 1 out and zeroed -- the written-out form in c25519_transcript_zs below, minus 25 loads and 25 stores of the state per z).  Requires pos == 16, pos_begin == 0; leaves them so.
__attribute__((target("avx512f,avx512vl"))) static void keccak_squeeze_pairs(uint64_t a[25], uint8_t *zs, uint64_t n, uint64_t k2, uint64_t k3, uint64_t k20) {
    C25519_KP_CONSTANTS C25519_KP_LOAD(a)
    const __m128i x2 = _mm_set_epi64x(0, (long long)k2), x3 = _mm_set_epi64x(0, (long long)k3), x20 = _mm_set_epi64x(0, (long long)k20), keep_hi = _mm_set_epi64x(-1, 0);
    for (uint64_t i = 0; i < n; i++) {
        r20 = _mm_xor_si128(r20, x2); r30 = _mm_xor_si128(r30, x3); r02 = _mm_xor_si128(r02, x20);          // lanes 2, 3, 20 are the low halves of r[2][0], r[3][0], r[0][2]
        C25519_KP_ROUNDS
        _mm_storeu_si128((__m128i *)(zs + 16 * i), _mm_unpacklo_epi64(r00, r10));                             // lanes 0 and 1
        r00 = _mm_and_si128(r00, keep_hi); r10 = _mm_and_si128(r10, keep_hi);
    }
    C25519_KP_STORE(a)
}
#undef C25519_KP_CONSTANTS
#undef C25519_KP_LOAD
#undef C25519_KP_ROUNDS
#undef C25519_KP_STORE

typedef void (*keccak_fn)(uint64_t *);
static inline keccak_fn keccak_pick() {
    __builtin_cpu_init();
#if !defined(C25519_KECCAK_MAX_FORM) || C25519_KECCAK_MAX_FORM >= 2      // (the macro: tests/test_fe26_host.py builds one harness without the vector form, so that the scalar forms' transcript path runs on an AVX-512 host too)
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl")) return keccak_f_pairs;
#endif
    return (__builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2")) ? keccak_f_bmi2 : keccak_f_generic;
}
static inline void keccak_f(uint64_t a[25]) { static const keccak_fn f = keccak_pick(); f(a); }
static inline const char *keccak_impl() { const keccak_fn f = keccak_pick(); return f == keccak_f_pairs ? "avx512vl-pairs" : f == keccak_f_bmi2 ? "bmi2" : "generic"; }
#else
static inline void keccak_f(uint64_t a[25]) { keccak_f_generic(a); }
static inline const char *keccak_impl() { return "generic"; }
#endif
#undef C25519_KECCAK_BODY
#undef C25519_K_ROUND

struct strobe {
    enum { R = 166, FI = 1, FA = 2, FC = 4, FT = 8, FM = 16, FK = 32 };
    uint64_t st[25]; uint8_t pos, pos_begin;
    uint8_t *bytes() { return (uint8_t *)st; }
    void run_f() { uint8_t *b = bytes(); b[pos] ^= pos_begin; b[pos + 1] ^= 0x04; b[R + 1] ^= 0x80; keccak_f(st); pos = 0; pos_begin = 0; }
    // duplex in runs that end at the rate boundary (byte loops without the boundary test inside vectorise)
    void absorb(const uint8_t *d, size_t n) {
        while (n) {
            uint8_t *b = bytes() + pos;
            const size_t k = n < (size_t)(R - pos) ? n : (size_t)(R - pos);
            for (size_t i = 0; i < k; i++) b[i] ^= d[i];
            d += k; n -= k; pos = (uint8_t)(pos + k);
            if (pos == R) run_f();
        }
    }
    void overwrite(const uint8_t *d, size_t n) {
        while (n) {
            const size_t k = n < (size_t)(R - pos) ? n : (size_t)(R - pos);
            memcpy(bytes() + pos, d, k);
            d += k; n -= k; pos = (uint8_t)(pos + k);
            if (pos == R) run_f();
        }
    }
    void squeeze(uint8_t *d, size_t n) {
        while (n) {
            const size_t k = n < (size_t)(R - pos) ? n : (size_t)(R - pos);
            memcpy(d, bytes() + pos, k); memset(bytes() + pos, 0, k);
            d += k; n -= k; pos = (uint8_t)(pos + k);
            if (pos == R) run_f();
        }
    }
    void begin_op(uint8_t flags, bool more) {
        if (more) return;
        uint8_t hdr[2] = {pos_begin, flags};
        pos_begin = (uint8_t)(pos + 1);
        absorb(hdr, 2);
        if ((flags & (FC | FK)) && pos != 0) run_f();
    }
    void meta_ad(const uint8_t *d, size_t n, bool more) { begin_op(FM | FA, more); absorb(d, n); }
    void ad(const uint8_t *d, size_t n, bool more) { begin_op(FA, more); absorb(d, n); }
    void prf(uint8_t *d, size_t n) { begin_op(FI | FA | FC, false); squeeze(d, n); }
    void key(const uint8_t *d, size_t n) { begin_op(FA | FC, false); overwrite(d, n); }
    void init(const char *proto) {
        memset(st, 0, sizeof st); pos = 0; pos_begin = 0;
        const uint8_t hdr[6] = {1, R + 2, 1, 0, 1, 96};
        memcpy(bytes(), hdr, 6); memcpy(bytes() + 6, "STROBEv1.0.2", 12);
        keccak_f(st);
        meta_ad((const uint8_t *)proto, strlen(proto), false);
    }
    void append_message(const char *label, const uint8_t *m, uint32_t n) {   // transcript.rs:69-74
        uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
        const size_t ll = strlen(label), total = 2 + ll + 4 + 2 + (size_t)n;
        if ((size_t)pos + total < (size_t)R) {      // (r6) the whole framed message stays inside the block: no boundary test per piece (the duplex calls below are 12 short loops per signature)
            uint8_t *b = bytes() + pos;
            b[0] ^= pos_begin; b[1] ^= FM | FA;                                       // begin_op(FM | FA): the header names where the previous operation began
            for (size_t i = 0; i < ll; i++) b[2 + i] ^= (uint8_t)label[i];
            for (size_t i = 0; i < 4; i++) b[2 + ll + i] ^= len[i];
            b[6 + ll] ^= (uint8_t)(pos + 1); b[7 + ll] ^= FA;                         // begin_op(FA): the meta operation began at pos + 1
            for (size_t i = 0; i < n; i++) b[8 + ll + i] ^= m[i];
            pos_begin = (uint8_t)(pos + 6 + ll + 1); pos = (uint8_t)(pos + total);
            return;
        }
        meta_ad((const uint8_t *)label, strlen(label), false); meta_ad(len, 4, true); ad(m, n, false);
    }
};

}  // namespace c25519_tr

// hrams: n x 64, sigs: n x 64 (s = bytes 32..63), zs out: n x 16
static void c25519_transcript_zs(const uint8_t *hrams, const uint8_t *sigs, uint64_t n, uint8_t *zs) {
    c25519_tr::strobe t;
    t.init("Merlin v1.0");                                                          // transcript.rs:54-58
    t.append_message("dom-sep", (const uint8_t *)"ed25519 batch verification", 26);  // batch.rs:168
    for (uint64_t i = 0; i < n; i++) t.append_message("hram", hrams + 64 * i, 64);   // batch.rs:195-197
    for (uint64_t i = 0; i < n; i++) t.append_message("sig.s", sigs + 64 * i + 32, 32);  // :199-201
    uint8_t zeros[32] = {0};                                                         // ZeroRng, batch.rs:49-76
    t.meta_ad((const uint8_t *)"rng", 3, false); t.key(zeros, 32);                   // transcript.rs:157-173
    const uint8_t len16[4] = {16, 0, 0, 0};
    typedef c25519_tr::strobe S;
    // lanes 2 / 3 / 20 of the block a z after the first starts in (pos = 16, pos_begin = 0): meta_ad(len16) header {0, M|A} at bytes 16, 17 and the length at 18 .. 21; prf header
    // {17, I|A|C} at 22, 23; run_f at pos 24: pos_begin 23, the 0x04 of STROBE's padding at 25, 0x80 at R + 1 = 167
    const uint64_t k2 = ((uint64_t)(S::FM | S::FA) << 8) | (16ull << 16) | (17ull << 48) | ((uint64_t)(S::FI | S::FA | S::FC) << 56), k3 = 23ull | (0x04ull << 8), k20 = 0x80ull << 56;
    for (uint64_t i = 0; i < n; i++) {                                               // transcript.rs:200-206
        if (t.pos == 16 && t.pos_begin == 0) {     // (r6) every z after the first finds the sponge 16 bytes into a fresh block: the two operations written out, on whole lanes
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
            if (c25519_tr::keccak_pick() == c25519_tr::keccak_f_pairs) { c25519_tr::keccak_squeeze_pairs(t.st, zs + 16 * i, n - i, k2, k3, k20); break; }
#endif
            t.st[2] ^= k2; t.st[3] ^= k3; t.st[20] ^= k20;
            c25519_tr::keccak_f(t.st);
            memcpy(zs + 16 * i, t.st, 16); t.st[0] = 0; t.st[1] = 0;                   // pos = 16, pos_begin = 0 again
            continue;
        }
        t.meta_ad(len16, 4, false); t.prf(zs + 16 * i, 16);
    }
}
