// The host transcript of the strict z-mode (csrc/transcript_host.h) under AddressSanitizer + UBSan: the lane-pair permutation's loads / stores on an exactly
// 200-byte heap state, and append_message's in-block path + the written-out z squeeze at every block offset (tests/test_oracle_sanitized.py builds and runs this).
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "transcript_host.h"
int main() {
    // heap-allocated exact-size state: any read past lane 24 is an ASan error
    for (int it = 0; it < 2000; it++) {
        uint64_t *a = (uint64_t *)malloc(200), *b = (uint64_t *)malloc(200);
        for (int i = 0; i < 25; i++) a[i] = b[i] = ((uint64_t)rand() << 40) ^ rand();
        c25519_tr::keccak_f_generic(a); if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl")) c25519_tr::keccak_f_pairs(b); else c25519_tr::keccak_f(b);
        if (memcmp(a, b, 200)) { printf("MISMATCH\n"); return 1; }
        free(a); free(b);
    }
    for (uint64_t n : {0, 1, 2, 3, 50, 333, 1000}) {
        std::vector<uint8_t> h(n * 64 + 1), s(n * 64 + 1), z(n * 16 + 1);
        for (auto &x : h) x = rand(); for (auto &x : s) x = rand();
        c25519_transcript_zs(h.data(), s.data(), n, z.data());
    }
    printf("sanitized ok\n"); return 0;
}
