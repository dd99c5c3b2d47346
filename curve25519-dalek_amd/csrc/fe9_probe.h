// A NINE-limb representation of GF(2^255 - 19) for the gfx950 multiplier, as a PROBE (not used by any kernel): the round-4 verdict asked to price it
// with a microbenchmark instead of an argument.  Limbs of 29 / 28 / 28 bits at bit positions ceil(85 i / 3) = 0, 29, 57, 85, 114, 142, 170, 199, 227
// (radix 2^28.33): 81 products instead of the 100 of fe26.h.  19 x a 29-bit limb does not fit the 32-bit operand of v_mad_u64_u32, so the x19 fold
// cannot ride on an operand as it does in the 10-limb form: the 17 column sums stay UNFOLDED (each < 2^63 for limbs below 2^29.9), the upper eight are
// carried, folded with nine x19 multiply-adds, and the lower nine carried: 90 multiply-adds + 18 carry steps against 100 + 9 pre-scalings + 10.
// What must stay bit-exact is the value mod p (u64/field.rs:111-214), not the limb split.  Host-fuzzed against big integers at the bound extremes
// (tests/test_fe26_host.py); rates in profiles/r05_instruction_rates.txt (c25519_microbench 70 / 71).
#pragma once
#include "fe26.h"
namespace c25519 {
struct fe9 { u32 v[9]; };
C25519_HD u32 fe9_x2(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 r; asm("v_add_u32_e32 %0, %1, %1" : "=v"(r) : "v"(x)); return r;
#else
    return 2u * x;
#endif
}
C25519_HD constexpr int fe9_pos(int i) { return (85 * i + 2) / 3; }
C25519_HD constexpr int fe9_wid(int i) { return fe9_pos(i + 1) - fe9_pos(i); }
// products a_i b_j sit at pos_i + pos_j, which is pos_(i+j) or one bit above it: the latter for (i mod 3, j mod 3) in {(1,1), (1,2), (2,1)}
C25519_HD constexpr bool fe9_dbl(int i, int j) { return fe9_pos(i) + fe9_pos(j) != fe9_pos(i + j); }
// chained carries (the form fe26.h uses with C25519_CHAIN): column k + 1 starts from the carry out of column k, which rides in as the 64-bit addend of
// its first multiply-add; the empty asm keeps LLVM from re-associating the carry to the end of the column
#if defined(__HIP_DEVICE_COMPILE__)
#define FE9_PIN(x) asm("" : "+v"(x))
#else
#define FE9_PIN(x)
#endif
template <bool SQ>
C25519_HD fe9 fe9_product(const fe9 &f, const fe9 &g) {
    u32 g2[9], g4[9];
#pragma unroll
    for (int j = 0; j < 9; j++) { g2[j] = fe9_x2(g.v[j]); g4[j] = SQ ? fe9_x2(g2[j]) : 0u; }
    // operand of the product f_i g_j in column i + j: doubled where the limb positions leave a bit (fe9_dbl), doubled again for the
    // off-diagonal terms of a square (taken once, i < j)
    auto term = [&](int i, int j) -> u32 {
        const bool d = fe9_dbl(i, j);
        if (SQ && i != j) return d ? g4[j] : g2[j];
        return d ? g2[j] : g.v[j];
    };
    u32 hi[9];
    u64 acc = 0;
    // upper columns 9 .. 16 (column k sits at 255 + pos_(k-9) exactly: 85 * 9 / 3 = 255)
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = 1; i < 9; i++) {
            const int j = k - i;
            if (j < 1 || j > 8 || (SQ && j < i)) continue;
            acc += (u64)f.v[i] * (u64)term(i, j); FE9_PIN(acc);
        }
        hi[k - 9] = (u32)acc & ((1u << fe9_wid(k - 9)) - 1u);
        acc >>= fe9_wid(k - 9);
    }
    hi[8] = (u32)acc;                                        // the carry out of column 16: below 2^32 for carried inputs
    acc = 0;
    fe9 r;
    // lower columns 0 .. 8, each with its x19 fold of the carried upper column
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc += (u64)hi[k] * 19u; FE9_PIN(acc);
#pragma unroll
        for (int i = 0; i <= k; i++) {
            const int j = k - i;
            if (SQ && j < i) continue;
            acc += (u64)f.v[i] * (u64)term(i, j); FE9_PIN(acc);
        }
        r.v[k] = (u32)acc & ((1u << fe9_wid(k)) - 1u);
        acc >>= fe9_wid(k);
    }
    const u64 t = (u64)r.v[0] + 19ull * acc;
    r.v[0] = (u32)t & ((1u << fe9_wid(0)) - 1u);
    r.v[1] += (u32)(t >> fe9_wid(0));
    return r;
}
C25519_HD fe9 fe9_mul(const fe9 &f, const fe9 &g) { return fe9_product<false>(f, g); }
C25519_HD fe9 fe9_sq(const fe9 &f) { return fe9_product<true>(f, f); }
}  // namespace c25519
