#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "transcript_host.h"
#include "pairs.h"
#include "rc_list.h"
#include "pairs2.h"
#include "pairs3.h"
#include "pairs4.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    uint64_t a[25], b[25]; srand(3);
    for (int it = 0; it < 10000; it++) { for (int i = 0; i < 25; i++) a[i] = ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ rand(); memcpy(b, a, 200); uint64_t a0[25]; memcpy(a0, a, 200);
        c25519_tr::keccak_f_generic(a); kp::keccak_f_pairs(b); { uint64_t c[25]; memcpy(c, a0, 200); kp::keccak_f_pairs2(c); if (memcmp(a, c, 200)) { printf("MISMATCH2 %d\n", it); return 1; } } if (memcmp(a, b, 200)) { printf("MISMATCH %d\n", it); return 1; } }
    uint64_t st[25]; const int N = 3000000;
    for (int rep = 0; rep < 3; rep++) {
        memset(st, 1, sizeof st); double t0 = now(); for (int i = 0; i < N; i++) kp::keccak_f_pairs(st); double t1 = now();
        printf("pairs: %.1f ns (%llx)   ", (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
        memset(st, 1, sizeof st); t0 = now(); for (int i = 0; i < N; i++) kp::keccak_f_pairs2(st); t1 = now();
        printf("pairs2: %.1f ns (%llx)   ", (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
        memset(st, 1, sizeof st); t0 = now(); for (int i = 0; i < N; i++) kp::keccak_f_pairs3(st); t1 = now();
        printf("pairs3: %.1f ns (%llx)   ", (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
        memset(st, 1, sizeof st); t0 = now(); for (int i = 0; i < N; i++) kp::keccak_f_pairs4(st); t1 = now();
        printf("pairs4: %.1f ns (%llx)   ", (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
        memset(st, 1, sizeof st); t0 = now(); for (int i = 0; i < N; i++) c25519_tr::keccak_f_bmi2(st); t1 = now();
        printf("bmi2: %.1f ns (%llx)\n", (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
    }
}
