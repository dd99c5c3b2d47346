// experiment 2: row form + early parity: the next round's column parities accumulate in five registers while the rows are stored
#pragma once
#include <stdint.h>
namespace kr2 {
static inline uint64_t rotl(uint64_t x, unsigned n) { return (x << n) | (x >> (64 - n)); }
#define KR2_ROW(first, E, o, i0, i1, i2, i3, i4, r0, r1, r2, r3, r4, D0, D1, D2, D3, D4, rc)               \
    {                                                                                                      \
        const uint64_t b0 = r0 ? rotl(A[i0] ^ D0, r0) : (A[i0] ^ D0), b1 = rotl(A[i1] ^ D1, r1), b2 = rotl(A[i2] ^ D2, r2), b3 = rotl(A[i3] ^ D3, r3), b4 = rotl(A[i4] ^ D4, r4); \
        const uint64_t e0 = b0 ^ (~b1 & b2) ^ (rc), e1 = b1 ^ (~b2 & b3), e2 = b2 ^ (~b3 & b4), e3 = b3 ^ (~b4 & b0), e4 = b4 ^ (~b0 & b1); \
        E[o + 0] = e0; E[o + 1] = e1; E[o + 2] = e2; E[o + 3] = e3; E[o + 4] = e4;                          \
        if (first) { n0 = e0; n1 = e1; n2 = e2; n3 = e3; n4 = e4; } else { n0 ^= e0; n1 ^= e1; n2 ^= e2; n3 ^= e3; n4 ^= e4; } \
    }
#define KR2_ROUND(A_, E_, rc)                                                                              \
    {                                                                                                      \
        const uint64_t *A = A_; uint64_t *E = E_;                                                          \
        const uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1), d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1); \
        uint64_t n0, n1, n2, n3, n4;                                                                       \
        KR2_ROW(1, E, 0, 0, 6, 12, 18, 24, 0, 44, 43, 21, 14, d0, d1, d2, d3, d4, rc)                      \
        KR2_ROW(0, E, 5, 3, 9, 10, 16, 22, 28, 20, 3, 45, 61, d3, d4, d0, d1, d2, 0)                       \
        KR2_ROW(0, E, 10, 1, 7, 13, 19, 20, 1, 6, 25, 8, 18, d1, d2, d3, d4, d0, 0)                        \
        KR2_ROW(0, E, 15, 4, 5, 11, 17, 23, 27, 36, 10, 15, 56, d4, d0, d1, d2, d3, 0)                     \
        KR2_ROW(0, E, 20, 2, 8, 14, 15, 21, 62, 55, 39, 41, 2, d2, d3, d4, d0, d1, 0)                      \
        c0 = n0; c1 = n1; c2 = n2; c3 = n3; c4 = n4;                                                       \
    }
__attribute__((target("bmi,bmi2"))) static void keccak_f_rows2(uint64_t a[25]) {
    static const uint64_t RC[24] = C25519_KECCAK_RC;
    uint64_t e[25];
    uint64_t c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20], c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21], c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22],
             c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23], c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
#define KR2_BAR __asm__ volatile("" : : "r"(e), "r"(a) : "memory");
    for (int rnd = 0; rnd < 24; rnd += 2) { KR2_BAR KR2_ROUND(a, e, RC[rnd]) KR2_BAR KR2_ROUND(e, a, RC[rnd + 1]) }
    KR2_BAR
}
}
