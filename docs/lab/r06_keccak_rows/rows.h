// experiment: row-by-row scalar Keccak with the state in memory (two buffers), ~186 instructions per round
#pragma once
#include <stdint.h>
namespace kr {
static inline uint64_t rotl(uint64_t x, unsigned n) { return (x << n) | (x >> (64 - n)); }
#define KR_ROW(E, o, i0, i1, i2, i3, i4, r0, r1, r2, r3, r4, D0, D1, D2, D3, D4, rc)                       \
    {                                                                                                      \
        const uint64_t b0 = r0 ? rotl(A[i0] ^ D0, r0) : (A[i0] ^ D0), b1 = rotl(A[i1] ^ D1, r1), b2 = rotl(A[i2] ^ D2, r2), b3 = rotl(A[i3] ^ D3, r3), b4 = rotl(A[i4] ^ D4, r4); \
        E[o + 0] = b0 ^ (~b1 & b2) ^ (rc); E[o + 1] = b1 ^ (~b2 & b3); E[o + 2] = b2 ^ (~b3 & b4); E[o + 3] = b3 ^ (~b4 & b0); E[o + 4] = b4 ^ (~b0 & b1); \
    }
#define KR_ROUND(A_, E_, rc)                                                                               \
    {                                                                                                      \
        const uint64_t *A = A_; uint64_t *E = E_;                                                          \
        const uint64_t c0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], c1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21], c2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], \
                       c3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23], c4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];  \
        const uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1), d3 = c2 ^ rotl(c4, 1), d4 = c3 ^ rotl(c0, 1); \
        KR_ROW(E, 0, 0, 6, 12, 18, 24, 0, 44, 43, 21, 14, d0, d1, d2, d3, d4, rc)                          \
        KR_ROW(E, 5, 3, 9, 10, 16, 22, 28, 20, 3, 45, 61, d3, d4, d0, d1, d2, 0)                           \
        KR_ROW(E, 10, 1, 7, 13, 19, 20, 1, 6, 25, 8, 18, d1, d2, d3, d4, d0, 0)                            \
        KR_ROW(E, 15, 4, 5, 11, 17, 23, 27, 36, 10, 15, 56, d4, d0, d1, d2, d3, 0)                         \
        KR_ROW(E, 20, 2, 8, 14, 15, 21, 62, 55, 39, 41, 2, d2, d3, d4, d0, d1, 0)                          \
    }
__attribute__((target("bmi,bmi2"))) static void keccak_f_rows(uint64_t a[25]) {
    static const uint64_t RC[24] = C25519_KECCAK_RC;
    uint64_t e[25];
#define KR_BAR __asm__ volatile("" : : "r"(e), "r"(a) : "memory");
    for (int rnd = 0; rnd < 24; rnd += 2) { KR_BAR KR_ROUND(a, e, RC[rnd]) KR_BAR KR_ROUND(e, a, RC[rnd + 1]) }
    KR_BAR
}
}
