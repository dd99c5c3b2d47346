// host micro-benchmark of the strict z-mode's sponge (tools/, not part of the library): g++ -O3 -std=c++17 -I curve25519-dalek_amd/csrc tools/keccak_bench.cpp
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "transcript_host.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    uint64_t st[25]; memset(st, 1, sizeof st);
    const int N = 2000000;
    double t0 = now();
    for (int i = 0; i < N; i++) c25519_tr::keccak_f(st);
    double t1 = now();
    printf("keccak_f (%s): %.1f ns per permutation (%llx)\n", c25519_tr::keccak_impl(), (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
    t0 = now();
    for (int i = 0; i < N; i++) c25519_tr::keccak_f_generic(st);
    t1 = now();
    printf("keccak_f_generic: %.1f ns per permutation (%llx)\n", (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
    const size_t n = 1 << 18;
    std::vector<uint8_t> h(n * 64, 7), s(n * 64, 9), z(n * 16);
    t0 = now(); c25519_transcript_zs(h.data(), s.data(), n, z.data()); t1 = now();
    printf("transcript: %.1f ns per signature = %.2f M/s\n", (t1 - t0) / n * 1e9, n / (t1 - t0) / 1e6);
    return 0;
}
