// Field arithmetic over GF(2^255-19) for gfx950 (MI355X), one field element per lane.
//
// Replaces the reference's FieldElement51 (curve25519-dalek/src/backend/serial/u64/field.rs:43-575)
// on the device.  External wire form is the same canonical 32-byte little-endian encoding
// (from_bytes :338 ignores bit 255, to_bytes :368 is canonical).
//
// Register layout: the reference's 5 x u64 radix-2^51 limbs are held as 5 (lo26, hi25) u32 pairs,
// i.e. 10 x u32 limbs at bit positions 0,26,51,77,102,128,153,179,204,230 (radix 2^25.5) -- the same
// 10 VGPRs a 5 x u64 element occupies, but split at bit 26 instead of bit 32 so that every partial
// product is ONE v_mad_u64_u32 (32x32+64 -> 64) with no cross-half carry glue.  gfx950 has no
// 64x64 -> 128 multiplier: a radix-2^51 limb product lowers to 4 v_mad_u64_u32 + adds, so both
// layouts spend exactly 100 multiplier issues per field mul; this one has no glue.
//
// Limbs are UNSIGNED.  Subtraction adds a multiple of p limb-wise; to avoid a carry pass after
// every add/sub (the reference's Sub always reduces, field.rs:82-102) magnitudes are tracked in
// the TYPE:
//     feT  "tight"  even limbs <= 2^26 + 2^19, odd limbs <= 2^25 + 2^19   (output of mul/sq/carry)
//     feL  "loose"  even limbs <= 1.52 * 2^27,  odd <= 1.52 * 2^26        (one add/sub of tights)
//     feW  "wide"   even limbs <= 2^29,         odd <= 2^28               (sub of a loose, ...)
// mul(f: feW, g: feL) and sq(f: feL) are safe: the largest column sum is
// 124.5 * max_even(f) * max_even(g) < 2^64, and 19*g_i / 38*f_odd fit in 32 bits.
// tests/test_fe26_host.py compiles this header for the host with C25519_CHECK_BOUNDS and fuzzes
// every operation at the extreme of each bound against Python big-ints.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define C25519_HD __host__ __device__ __forceinline__
#else
#define C25519_HD inline
#endif

#ifdef C25519_CHECK_BOUNDS
#include <stdio.h>
#include <stdlib.h>
// the counterpart of the reference's debug asserts on limb magnitudes (u64/field.rs:162-166).  Host: report and abort.
// Device (make debug -> lib/libc25519hip_dbg.so): trap -- the kernel dies, the stream reports an error, the call fails.
#if defined(__HIP_DEVICE_COMPILE__)
#define C25519_BOUND(limbs, even_max, odd_max, what)                                              \
    do {                                                                                          \
        for (int _i = 0; _i < 10; _i++)                                                           \
            if ((uint64_t)(limbs)[_i] > ((_i & 1) ? (uint64_t)(odd_max) : (uint64_t)(even_max))) __builtin_trap(); \
    } while (0)
#else
#define C25519_BOUND(limbs, even_max, odd_max, what)                                              \
    do {                                                                                          \
        for (int _i = 0; _i < 10; _i++)                                                           \
            if ((uint64_t)(limbs)[_i] > ((_i & 1) ? (uint64_t)(odd_max) : (uint64_t)(even_max))) { \
                fprintf(stderr, "fe26 bound violated: %s limb %d = %llu\n", what, _i,             \
                        (unsigned long long)(limbs)[_i]);                                         \
                abort();                                                                          \
            }                                                                                     \
    } while (0)
#endif
#else
#define C25519_BOUND(limbs, even_max, odd_max, what) do { } while (0)
#endif

namespace c25519 {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr u32 M26 = (1u << 26) - 1;
constexpr u32 M25 = (1u << 25) - 1;

constexpr u64 T_EVEN = (1ull << 26) + (1ull << 19), T_ODD = (1ull << 25) + (1ull << 19);
constexpr u64 L_EVEN = 204010946ull /* 1.52*2^27 */, L_ODD = 102005473ull /* 1.52*2^26 */;
constexpr u64 W_EVEN = 1ull << 29, W_ODD = 1ull << 28;

struct feT { u32 v[10]; };
struct feL {
    u32 v[10];
    C25519_HD feL() {}
    C25519_HD feL(const feT &t) { for (int i = 0; i < 10; i++) v[i] = t.v[i]; }
};
struct feW {
    u32 v[10];
    C25519_HD feW() {}
    C25519_HD feW(const feT &t) { for (int i = 0; i < 10; i++) v[i] = t.v[i]; }
    C25519_HD feW(const feL &t) { for (int i = 0; i < 10; i++) v[i] = t.v[i]; }
};

C25519_HD feT fe_zero() { feT r; for (int i = 0; i < 10; i++) r.v[i] = 0; return r; }
C25519_HD feT fe_one() { feT r = fe_zero(); r.v[0] = 1; return r; }
C25519_HD feT fe_small(u32 x) { feT r = fe_zero(); r.v[0] = x; return r; }  // x < 2^26

// 2 x as an ADD: the shift LLVM makes of 2u * x (and of x + x) issues at a quarter of the rate of v_add_u32 on gfx950
// (profiles/r02_instruction_rates.txt), and a product has five of them, a square eight
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u32 fe_x2(u32 x) { u32 r; asm("v_add_u32_e32 %0, %1, %1" : "=v"(r) : "v"(x)); return r; }
#else
static inline u32 fe_x2(u32 x) { return 2u * x; }
#endif

// ---- additive ops (no carries) -------------------------------------------------------------
// field.rs:58-73 (Add never reduces)
C25519_HD feL fe_add(const feT &a, const feT &b) {
    feL r; for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
C25519_HD feL fe_twice(const feT &a) {
    feL r; for (int i = 0; i < 10; i++) r.v[i] = fe_x2(a.v[i]);
    return r;
}
// (2 * tight) + tight stays loose: 3 * (2^26 + 2^19) < 1.52 * 2^27
C25519_HD feL fe_add_lt(const feL &twice_tight, const feT &b) {
    feL r; for (int i = 0; i < 10; i++) r.v[i] = twice_tight.v[i] + b.v[i];
    C25519_BOUND(r.v, L_EVEN, L_ODD, "fe_add_lt");
    return r;
}
C25519_HD feW fe_add_w(const feL &a, const feL &b) {  // loose + loose
    feW r; for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    C25519_BOUND(r.v, W_EVEN, W_ODD, "fe_add_w");
    return r;
}
// a - b = a + 2p - b, b tight.   2p limbs: 2^27-38, 2^26-2, 2^27-2, 2^26-2, ...
C25519_HD feL fe_sub(const feT &a, const feT &b) {
    feL r;
    r.v[0] = a.v[0] + 0x7ffffdau - b.v[0];
    for (int i = 1; i < 10; i++) r.v[i] = a.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - b.v[i];
    C25519_BOUND(r.v, L_EVEN, L_ODD, "fe_sub");
    return r;
}
// a - b = a + 4p - b, a and b loose.   4p limbs: 2^28-76, 2^27-4, 2^28-4, ...
C25519_HD feW fe_sub_w(const feL &a, const feL &b) {
    feW r;
    r.v[0] = a.v[0] + 0xfffffb4u - b.v[0];
    for (int i = 1; i < 10; i++) r.v[i] = a.v[i] + ((i & 1) ? 0x7fffffcu : 0xffffffcu) - b.v[i];
    C25519_BOUND(r.v, W_EVEN, W_ODD, "fe_sub_w");
    return r;
}
C25519_HD feL fe_neg(const feT &a) { return fe_sub(fe_zero(), a); }
// m == ~0: 2p - a (a + 2p - a limb by limb, as fe_sub: loose); m == 0: a.  Two's complement: ~a + (2p limb + 1) -- one v_xad_u32 per limb on the device (the lazy sign of
// the accumulations: fe26x.h ge_madd_lazy_p3_lockstep, ge26.h ge_madd_lazy_p3)
C25519_HD feL fe_cond_neg(const feT &a, u32 m) {
    feL r;
    const u32 c0 = 0x7ffffdbu & m, ce = 0x7ffffffu & m, co = 0x3ffffffu & m;
    r.v[0] = (a.v[0] ^ m) + c0;
    for (int i = 1; i < 10; i++) r.v[i] = (a.v[i] ^ m) + ((i & 1) ? co : ce);
    C25519_BOUND(r.v, L_EVEN, L_ODD, "fe_cond_neg");
    return r;
}

// Weak reduction: all carries computed in parallel (like FieldElement51::reduce, field.rs:290-323).
// Any limbs < 2^32 in, tight out.
C25519_HD feT fe_carry(const feW &a) {
    u32 c[10];
    feT r;
    for (int i = 0; i < 10; i++) {
        c[i] = a.v[i] >> ((i & 1) ? 25 : 26);
        r.v[i] = a.v[i] & ((i & 1) ? M25 : M26);
    }
    r.v[0] += 19u * c[9];
    for (int i = 1; i < 10; i++) r.v[i] += c[i - 1];
    C25519_BOUND(r.v, T_EVEN, T_ODD, "fe_carry");
    return r;
}

// ---- multiplication --------------------------------------------------------------------------
// Serial carry chain over the ten 64-bit column sums (the 10-limb analogue of field.rs:175-209).
C25519_HD feT fe_carry64(u64 h[10]) {
    feT r;
    u64 c;
    c = h[0] >> 26; r.v[0] = (u32)h[0] & M26; h[1] += c;
    c = h[1] >> 25; r.v[1] = (u32)h[1] & M25; h[2] += c;
    c = h[2] >> 26; r.v[2] = (u32)h[2] & M26; h[3] += c;
    c = h[3] >> 25; r.v[3] = (u32)h[3] & M25; h[4] += c;
    c = h[4] >> 26; r.v[4] = (u32)h[4] & M26; h[5] += c;
    c = h[5] >> 25; r.v[5] = (u32)h[5] & M25; h[6] += c;
    c = h[6] >> 26; r.v[6] = (u32)h[6] & M26; h[7] += c;
    c = h[7] >> 25; r.v[7] = (u32)h[7] & M25; h[8] += c;
    c = h[8] >> 26; r.v[8] = (u32)h[8] & M26; h[9] += c;
    c = h[9] >> 25; r.v[9] = (u32)h[9] & M25;
    // c < 2^39; 19c < 2^44: fold into limb 0 and carry once more into limb 1
    u64 t = (u64)r.v[0] + 19ull * c;
    r.v[0] = (u32)t & M26;
    r.v[1] += (u32)(t >> 26);        // < 2^18
    C25519_BOUND(r.v, T_EVEN, T_ODD, "fe_carry64");
    return r;
}

// Chained form: column k+1 starts from the carry out of column k, so the carry rides in as the 64-bit addend of
// the column's first v_mad_u64_u32 instead of costing a separate 64-bit add (9 VOP3 issue slots per product).
// fe_finish64 takes columns that already contain their carry-in.
// Worth +4 % on the register-resident, high-occupancy kernels (comb, Montgomery ladder: kernels.hip sets
// C25519_CHAIN 1); the gather/latency-bound kernels want the ten independent column sums for ILP instead
// (k_var_base -15 %, k_reduce_level -19 %, k_long_segments -15 % when chained), hence off by default.
#ifndef C25519_CHAIN
#define C25519_CHAIN 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && C25519_CHAIN
// (r6, last) The pin is an INPUT of a volatile empty statement.  Until the end of round 6 it was `asm("" : "+v"(x))` -- a statement that DEFINES the register -- and behind
// every such statement the compiler's hazard recogniser put an s_nop (it must assume the worst of a statement it cannot see into): 0.45 - 0.75 s_nop per v_mad_u64_u32
// in every chained-form kernel (k_x25519 334 per ladder step, k_prep_compressed 2354, k_mul_base_ctp 1002).  Same vector instructions either way (tools/isa_stats.py);
// a lone wave's fe_mul 917 -> 677 cycles, fe_sq 572 -> 423, at eight waves per SIMD -2.5 % (profiles/r06_ab_pin_input.txt); k_mul_base_ctp 1.86 -> 1.77 ms per 2^20
// scalars, k_prep_compressed -3 %, small MSMs over encoded points -7 %.  -DC25519_PIN_DEF: the old form (A/B).
#ifndef C25519_PIN_DEF
#define C25519_PIN(x) asm volatile("" :: "v"(x))
#else
#define C25519_PIN(x) asm("" : "+v"(x))
#endif
#else
#define C25519_PIN(x)
#endif
#define C25519_MAD(acc, a, b) acc += (u64)(a) * (u64)(b); C25519_PIN(acc)

#if C25519_CHAIN
// (the empty asm pins "carry + first product" as one value, otherwise LLVM reassociates the carry to the end
// of the column sum and the separate add comes back)
#define C25519_COL(k, a, b) h[k] = (h[(k) - 1] >> (((k) & 1) ? 26 : 25)) + (u64)(a) * (u64)(b); C25519_PIN(h[k])
C25519_HD feT fe_finish64(const u64 h[10]) {
    feT r;
    for (int i = 0; i < 10; i++) r.v[i] = (u32)h[i] & ((i & 1) ? M25 : M26);
    u64 c = h[9] >> 25;
    u64 t = (u64)r.v[0] + 19ull * c;
    r.v[0] = (u32)t & M26;
    r.v[1] += (u32)(t >> 26);
    C25519_BOUND(r.v, T_EVEN, T_ODD, "fe_finish64");
    return r;
}
#else
#define C25519_COL(k, a, b) h[k] = (u64)(a) * (u64)(b)
#define fe_finish64 fe_carry64
#endif

// What it computes: FieldElement51::mul (u64/field.rs:111-214), the value mod p.  How: the 10-limb column schedule of the reference's
// FieldElement2625::mul (backend/serial/u32/field.rs:128-200 -- which products are doubled, which operand carries the x19: the
// multiplication table of radix 2^25.5, ref10 lineage): h_k = sum_{i+j=k} f_i g_j [x2 if i,j odd] + 19 sum_{i+j=k+10} f_i g_j [x2 if i,j odd].
// Unsigned limbs, bound classes in the type, v_mad_u64_u32 chaining and the carry forms around it are this library's.
C25519_HD feT fe_mul(const feW &f, const feL &g) {
    C25519_BOUND(f.v, W_EVEN, W_ODD, "fe_mul f");
    C25519_BOUND(g.v, L_EVEN, L_ODD, "fe_mul g");
    const u32 f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4];
    const u32 f5 = f.v[5], f6 = f.v[6], f7 = f.v[7], f8 = f.v[8], f9 = f.v[9];
    const u32 g0 = g.v[0], g1 = g.v[1], g2 = g.v[2], g3 = g.v[3], g4 = g.v[4];
    const u32 g5 = g.v[5], g6 = g.v[6], g7 = g.v[7], g8 = g.v[8], g9 = g.v[9];
    const u32 g1_19 = 19u * g1, g2_19 = 19u * g2, g3_19 = 19u * g3, g4_19 = 19u * g4, g5_19 = 19u * g5;
    const u32 g6_19 = 19u * g6, g7_19 = 19u * g7, g8_19 = 19u * g8, g9_19 = 19u * g9;
    const u32 f1_2 = fe_x2(f1), f3_2 = fe_x2(f3), f5_2 = fe_x2(f5), f7_2 = fe_x2(f7), f9_2 = fe_x2(f9);
    u64 h[10];
    h[0] = (u64)f0 * g0;
    C25519_MAD(h[0], f1_2, g9_19); C25519_MAD(h[0], f2, g8_19); C25519_MAD(h[0], f3_2, g7_19);
    C25519_MAD(h[0], f4, g6_19); C25519_MAD(h[0], f5_2, g5_19); C25519_MAD(h[0], f6, g4_19);
    C25519_MAD(h[0], f7_2, g3_19); C25519_MAD(h[0], f8, g2_19); C25519_MAD(h[0], f9_2, g1_19);
    C25519_COL(1, f0, g1);
    C25519_MAD(h[1], f1, g0); C25519_MAD(h[1], f2, g9_19); C25519_MAD(h[1], f3, g8_19);
    C25519_MAD(h[1], f4, g7_19); C25519_MAD(h[1], f5, g6_19); C25519_MAD(h[1], f6, g5_19);
    C25519_MAD(h[1], f7, g4_19); C25519_MAD(h[1], f8, g3_19); C25519_MAD(h[1], f9, g2_19);
    C25519_COL(2, f0, g2);
    C25519_MAD(h[2], f1_2, g1); C25519_MAD(h[2], f2, g0); C25519_MAD(h[2], f3_2, g9_19);
    C25519_MAD(h[2], f4, g8_19); C25519_MAD(h[2], f5_2, g7_19); C25519_MAD(h[2], f6, g6_19);
    C25519_MAD(h[2], f7_2, g5_19); C25519_MAD(h[2], f8, g4_19); C25519_MAD(h[2], f9_2, g3_19);
    C25519_COL(3, f0, g3);
    C25519_MAD(h[3], f1, g2); C25519_MAD(h[3], f2, g1); C25519_MAD(h[3], f3, g0);
    C25519_MAD(h[3], f4, g9_19); C25519_MAD(h[3], f5, g8_19); C25519_MAD(h[3], f6, g7_19);
    C25519_MAD(h[3], f7, g6_19); C25519_MAD(h[3], f8, g5_19); C25519_MAD(h[3], f9, g4_19);
    C25519_COL(4, f0, g4);
    C25519_MAD(h[4], f1_2, g3); C25519_MAD(h[4], f2, g2); C25519_MAD(h[4], f3_2, g1);
    C25519_MAD(h[4], f4, g0); C25519_MAD(h[4], f5_2, g9_19); C25519_MAD(h[4], f6, g8_19);
    C25519_MAD(h[4], f7_2, g7_19); C25519_MAD(h[4], f8, g6_19); C25519_MAD(h[4], f9_2, g5_19);
    C25519_COL(5, f0, g5);
    C25519_MAD(h[5], f1, g4); C25519_MAD(h[5], f2, g3); C25519_MAD(h[5], f3, g2);
    C25519_MAD(h[5], f4, g1); C25519_MAD(h[5], f5, g0); C25519_MAD(h[5], f6, g9_19);
    C25519_MAD(h[5], f7, g8_19); C25519_MAD(h[5], f8, g7_19); C25519_MAD(h[5], f9, g6_19);
    C25519_COL(6, f0, g6);
    C25519_MAD(h[6], f1_2, g5); C25519_MAD(h[6], f2, g4); C25519_MAD(h[6], f3_2, g3);
    C25519_MAD(h[6], f4, g2); C25519_MAD(h[6], f5_2, g1); C25519_MAD(h[6], f6, g0);
    C25519_MAD(h[6], f7_2, g9_19); C25519_MAD(h[6], f8, g8_19); C25519_MAD(h[6], f9_2, g7_19);
    C25519_COL(7, f0, g7);
    C25519_MAD(h[7], f1, g6); C25519_MAD(h[7], f2, g5); C25519_MAD(h[7], f3, g4);
    C25519_MAD(h[7], f4, g3); C25519_MAD(h[7], f5, g2); C25519_MAD(h[7], f6, g1);
    C25519_MAD(h[7], f7, g0); C25519_MAD(h[7], f8, g9_19); C25519_MAD(h[7], f9, g8_19);
    C25519_COL(8, f0, g8);
    C25519_MAD(h[8], f1_2, g7); C25519_MAD(h[8], f2, g6); C25519_MAD(h[8], f3_2, g5);
    C25519_MAD(h[8], f4, g4); C25519_MAD(h[8], f5_2, g3); C25519_MAD(h[8], f6, g2);
    C25519_MAD(h[8], f7_2, g1); C25519_MAD(h[8], f8, g0); C25519_MAD(h[8], f9_2, g9_19);
    C25519_COL(9, f0, g9);
    C25519_MAD(h[9], f1, g8); C25519_MAD(h[9], f2, g7); C25519_MAD(h[9], f3, g6);
    C25519_MAD(h[9], f4, g5); C25519_MAD(h[9], f5, g4); C25519_MAD(h[9], f6, g3);
    C25519_MAD(h[9], f7, g2); C25519_MAD(h[9], f8, g1); C25519_MAD(h[9], f9, g0);
    return fe_finish64(h);
}

// FieldElement51::pow2k body (u64/field.rs:454-559), the value mod p; the 55 products of FieldElement2625::square_inner
// (backend/serial/u32/field.rs:551-593), with the x2 that the reference applies to sums of products moved onto an operand (38 f_odd: fits
// 32 bits for loose limbs, the bound stated at the top of this file).
C25519_HD feT fe_sq(const feL &f) {
    C25519_BOUND(f.v, L_EVEN, L_ODD, "fe_sq f");
    const u32 f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4];
    const u32 f5 = f.v[5], f6 = f.v[6], f7 = f.v[7], f8 = f.v[8], f9 = f.v[9];
    const u32 f0_2 = fe_x2(f0), f1_2 = fe_x2(f1), f2_2 = fe_x2(f2), f3_2 = fe_x2(f3), f4_2 = fe_x2(f4);
    const u32 f5_2 = fe_x2(f5), f6_2 = fe_x2(f6), f7_2 = fe_x2(f7);
    const u32 f5_38 = 38u * f5, f6_19 = 19u * f6, f7_38 = 38u * f7, f8_19 = 19u * f8, f9_38 = 38u * f9;
    u64 h[10];
    h[0] = (u64)f0 * f0;
    C25519_MAD(h[0], f1_2, f9_38); C25519_MAD(h[0], f2_2, f8_19); C25519_MAD(h[0], f3_2, f7_38);
    C25519_MAD(h[0], f4_2, f6_19); C25519_MAD(h[0], f5, f5_38);
    C25519_COL(1, f0_2, f1);
    C25519_MAD(h[1], f2, f9_38); C25519_MAD(h[1], f3_2, f8_19); C25519_MAD(h[1], f4, f7_38);
    C25519_MAD(h[1], f5_2, f6_19);
    C25519_COL(2, f0_2, f2);
    C25519_MAD(h[2], f1_2, f1); C25519_MAD(h[2], f3_2, f9_38); C25519_MAD(h[2], f4_2, f8_19);
    C25519_MAD(h[2], f5_2, f7_38); C25519_MAD(h[2], f6, f6_19);
    C25519_COL(3, f0_2, f3);
    C25519_MAD(h[3], f1_2, f2); C25519_MAD(h[3], f4, f9_38); C25519_MAD(h[3], f5_2, f8_19);
    C25519_MAD(h[3], f6, f7_38);
    C25519_COL(4, f0_2, f4);
    C25519_MAD(h[4], f1_2, f3_2); C25519_MAD(h[4], f2, f2); C25519_MAD(h[4], f5_2, f9_38);
    C25519_MAD(h[4], f6_2, f8_19); C25519_MAD(h[4], f7, f7_38);
    C25519_COL(5, f0_2, f5);
    C25519_MAD(h[5], f1_2, f4); C25519_MAD(h[5], f2_2, f3); C25519_MAD(h[5], f6, f9_38);
    C25519_MAD(h[5], f7_2, f8_19);
    C25519_COL(6, f0_2, f6);
    C25519_MAD(h[6], f1_2, f5_2); C25519_MAD(h[6], f2_2, f4); C25519_MAD(h[6], f3_2, f3);
    C25519_MAD(h[6], f7_2, f9_38); C25519_MAD(h[6], f8, f8_19);
    C25519_COL(7, f0_2, f7);
    C25519_MAD(h[7], f1_2, f6); C25519_MAD(h[7], f2_2, f5); C25519_MAD(h[7], f3_2, f4);
    C25519_MAD(h[7], f8, f9_38);
    C25519_COL(8, f0_2, f8);
    C25519_MAD(h[8], f1_2, f7_2); C25519_MAD(h[8], f2_2, f6); C25519_MAD(h[8], f3_2, f5_2);
    C25519_MAD(h[8], f4, f4); C25519_MAD(h[8], f9, f9_38);
    C25519_COL(9, f0_2, f9);
    C25519_MAD(h[9], f1_2, f8); C25519_MAD(h[9], f2_2, f7); C25519_MAD(h[9], f3_2, f6);
    C25519_MAD(h[9], f4_2, f5);
    return fe_finish64(h);
}

// f * small constant c (c < 2^20), e.g. a24 = 121666 in the Montgomery ladder
// (montgomery.rs:454 does a full Mul by APLUS2_OVER_FOUR; 10 products suffice).
C25519_HD feT fe_mul_small(const feW &f, u32 c) {
    u64 h[10];
    for (int i = 0; i < 10; i++) h[i] = (u64)f.v[i] * c;
    return fe_carry64(h);
}

// field.rs:454 pow2k
C25519_HD feT fe_pow2k(feT x, int k) {
    for (int i = 0; i < k; i++) x = fe_sq(x);
    return x;
}

// ---- encode / decode ---------------------------------------------------------------------------
// field.rs:338-363: 8 little-endian u32 words -> limbs; bit 255 dropped; values >= p accepted.
C25519_HD feT fe_from_words(const u32 w[8]) {
    feT r;
    r.v[0] = w[0] & M26;
    r.v[1] = ((w[0] >> 26) | (w[1] << 6)) & M25;
    r.v[2] = ((w[1] >> 19) | (w[2] << 13)) & M26;
    r.v[3] = ((w[2] >> 13) | (w[3] << 19)) & M25;
    r.v[4] = (w[3] >> 6) & M26;
    r.v[5] = w[4] & M25;
    r.v[6] = ((w[4] >> 25) | (w[5] << 7)) & M26;
    r.v[7] = ((w[5] >> 19) | (w[6] << 13)) & M25;
    r.v[8] = ((w[6] >> 12) | (w[7] << 20)) & M26;
    r.v[9] = (w[7] >> 6) & M25;
    return r;
}

// Fully carried limbs (each < 2^26 / 2^25) of the canonical representative in [0, p).
// Same q-trick as field.rs:368-405.
C25519_HD void fe_canonical_limbs(const feW &a, u32 l[10]) {
    feT t = fe_carry(a);
    t = fe_carry(t);  // now value < 2^255 + tiny, limbs: l0 < 2^26 + 19, others exact
    for (int i = 0; i < 10; i++) l[i] = t.v[i];
    // sequential carry to make every limb exact (value < 2p guaranteed)
    u32 c;
    for (int i = 0; i < 9; i++) { c = l[i] >> ((i & 1) ? 25 : 26); l[i] &= (i & 1) ? M25 : M26; l[i + 1] += c; }
    c = l[9] >> 25; l[9] &= M25; l[0] += 19u * c;
    for (int i = 0; i < 9; i++) { c = l[i] >> ((i & 1) ? 25 : 26); l[i] &= (i & 1) ? M25 : M26; l[i + 1] += c; }
    // q = (value + 19) >> 255
    u32 q = (l[0] + 19u) >> 26;
    for (int i = 1; i < 10; i++) q = (l[i] + q) >> ((i & 1) ? 25 : 26);
    l[0] += 19u * q;
    for (int i = 0; i < 9; i++) { c = l[i] >> ((i & 1) ? 25 : 26); l[i] &= (i & 1) ? M25 : M26; l[i + 1] += c; }
    l[9] &= M25;
}

C25519_HD void fe_limbs_to_words(const u32 l[10], u32 w[8]) {
    w[0] = l[0] | (l[1] << 26);
    w[1] = (l[1] >> 6) | (l[2] << 19);
    w[2] = (l[2] >> 13) | (l[3] << 13);
    w[3] = (l[3] >> 19) | (l[4] << 6);
    w[4] = l[5] | (l[6] << 25);
    w[5] = (l[6] >> 7) | (l[7] << 19);
    w[6] = (l[7] >> 13) | (l[8] << 12);
    w[7] = (l[8] >> 20) | (l[9] << 6);
}

// field.rs:368-450 to_bytes (canonical), as 8 u32 words
C25519_HD void fe_to_words(const feW &a, u32 w[8]) {
    u32 l[10];
    fe_canonical_limbs(a, l);
    fe_limbs_to_words(l, w);
}

// canonical tight limbs (useful for storing table entries / equality)
C25519_HD feT fe_canon(const feW &a) {
    feT r; fe_canonical_limbs(a, r.v); return r;
}

// field.rs:156-170
C25519_HD u32 fe_is_negative(const feW &a) { u32 l[10]; fe_canonical_limbs(a, l); return l[0] & 1u; }
C25519_HD bool fe_is_zero(const feW &a) {
    u32 l[10]; fe_canonical_limbs(a, l);
    u32 acc = 0; for (int i = 0; i < 10; i++) acc |= l[i];
    return acc == 0;
}
C25519_HD bool fe_eq(const feW &a, const feW &b) {
    u32 x[10], y[10]; fe_canonical_limbs(a, x); fe_canonical_limbs(b, y);
    u32 acc = 0; for (int i = 0; i < 10; i++) acc |= x[i] ^ y[i];
    return acc == 0;
}

// branch-free selects (per-lane data-dependent choice): the condition becomes a 64-bit lane mask ONCE (lane_mask: a ballot, i.e. an SGPR pair)
// and every select is written as v_cndmask_b32_e64 with that pair, so the instruction form does not depend on where LLVM happens to keep a
// compare result.  Why this exists: the one-instruction probe of the VOP2 form (condition in VCC) reads 22.5 cycles per wave against 4.5 for
// the VOP3 form (profiles/r04_instruction_rates.txt) -- a penalty of BACK-TO-BACK VCC readers, as it turned out: with the ladder's 60 selects
// per step, the fixed-base recoding and the normaliser all moved to this form, k_x25519, k_mul_base<5, CT>, verify_batch and the MSM measure
// the same as before within 0.2 % (profiles/r04_ab_select_forms.txt).  Kept because it is never the slower form; not a speed-up.
// Constant time as before: a select executes identically whatever its mask holds.
#if defined(__HIP_DEVICE_COMPILE__)
// (the mask is a 64-bit ballot handed to v_cndmask_b32_e64 as an SGPR PAIR: wave64 only -- gfx950 / CDNA; a wave32 target must not build this)
#if !defined(__GFX9__)
#error "fe26.h lane masks are 64-bit SGPR pairs: wave64 targets (the GFX9 / CDNA family, gfx950) only"
#endif
typedef unsigned long long lanemask;
__device__ __forceinline__ lanemask lane_mask(bool c) { return __ballot(c); }
__device__ __forceinline__ u32 sel_u32(u32 a, u32 b, lanemask m) {       // lane's bit of m set -> b
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}
#else
typedef bool lanemask;
C25519_HD lanemask lane_mask(bool c) { return c; }
C25519_HD u32 sel_u32(u32 a, u32 b, lanemask m) { return m ? b : a; }
#endif
C25519_HD feT fe_select_m(const feT &a, const feT &b, lanemask m) { feT r; for (int i = 0; i < 10; i++) r.v[i] = sel_u32(a.v[i], b.v[i], m); return r; }
C25519_HD feL fe_select_m(const feL &a, const feL &b, lanemask m) { feL r; for (int i = 0; i < 10; i++) r.v[i] = sel_u32(a.v[i], b.v[i], m); return r; }
C25519_HD feW fe_select_m(const feW &a, const feW &b, lanemask m) { feW r; for (int i = 0; i < 10; i++) r.v[i] = sel_u32(a.v[i], b.v[i], m); return r; }
C25519_HD feT fe_select(const feT &a, const feT &b, bool choose_b) { return fe_select_m(a, b, lane_mask(choose_b)); }
C25519_HD feL fe_select(const feL &a, const feL &b, bool choose_b) { return fe_select_m(a, b, lane_mask(choose_b)); }
C25519_HD feW fe_select(const feW &a, const feW &b, bool choose_b) { return fe_select_m(a, b, lane_mask(choose_b)); }
C25519_HD void fe_cswap_m(feT &a, feT &b, lanemask m) {
    for (int i = 0; i < 10; i++) { const u32 x = a.v[i], y = b.v[i]; a.v[i] = sel_u32(x, y, m); b.v[i] = sel_u32(y, x, m); }
}
C25519_HD void fe_cswap(feT &a, feT &b, u32 swap) { fe_cswap_m(a, b, lane_mask(swap != 0)); }  // montgomery.rs:199 conditional_swap
C25519_HD feT fe_cneg(const feT &a, bool neg) { return fe_select(a, fe_carry(fe_neg(a)), neg); }

// ---- fixed addition chains (field.rs:176-306) --------------------------------------------------
C25519_HD void fe_pow22501(const feT &x, feT &t19, feT &t3) {
    feT t0 = fe_sq(x);
    feT t1 = fe_sq(fe_sq(t0));
    feT t2 = fe_mul(x, t1);
    t3 = fe_mul(t0, t2);
    feT t4 = fe_sq(t3);
    feT t5 = fe_mul(t2, t4);
    feT t6 = fe_pow2k(t5, 5);
    feT t7 = fe_mul(t6, t5);
    feT t8 = fe_pow2k(t7, 10);
    feT t9 = fe_mul(t8, t7);
    feT t10 = fe_pow2k(t9, 20);
    feT t11 = fe_mul(t10, t9);
    feT t12 = fe_pow2k(t11, 10);
    feT t13 = fe_mul(t12, t7);
    feT t14 = fe_pow2k(t13, 50);
    feT t15 = fe_mul(t14, t13);
    feT t16 = fe_pow2k(t15, 100);
    feT t17 = fe_mul(t16, t15);
    feT t18 = fe_pow2k(t17, 50);
    t19 = fe_mul(t18, t13);
}
// field.rs:283-292: x^(p-2); 0 -> 0
C25519_HD feT fe_invert(const feT &x) {
    feT t19, t3; fe_pow22501(x, t19, t3);
    return fe_mul(fe_pow2k(t19, 5), t3);
}
// field.rs:297-306: x^((p-5)/8)
C25519_HD feT fe_pow_p58(const feT &x) {
    feT t19, t3; fe_pow22501(x, t19, t3);
    return fe_mul(x, fe_pow2k(t19, 2));
}

}  // namespace c25519
