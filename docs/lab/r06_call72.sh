#!/bin/bash
# round 6, gpurun call 72: the strict z-mode's sponge on the GPU box's host core -- Keccak-f[1600] on AVX-512 (transcript_host.h keccak_f_avx512) against the BMI2 scalar form
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R; mkdir -p gpurun_out
out=gpurun_out/r06_keccak_avx512.txt; : > $out
grep -m1 "model name" /proc/cpuinfo >> $out
cat > /tmp/kb2.cpp <<'EOC'
#include <chrono>
#include <cstdio>
#include <cstring>
#include "transcript_host.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    uint64_t st[25]; const int N = 4000000;
    for (int rep = 0; rep < 3; rep++) {
        memset(st, 1, sizeof st); double t0 = now(); for (int i = 0; i < N; i++) c25519_tr::keccak_f_avx512(st); double t1 = now();
        printf("keccak_f_avx512: %.1f ns (%llx)   ", (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
        memset(st, 1, sizeof st); t0 = now(); for (int i = 0; i < N; i++) c25519_tr::keccak_f_avx512_lat(st); t1 = now();
        printf("keccak_f_avx512_lat: %.1f ns (%llx)   ", (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
        memset(st, 1, sizeof st); t0 = now(); for (int i = 0; i < N; i++) c25519_tr::keccak_f_bmi2(st); t1 = now();
        printf("keccak_f_bmi2: %.1f ns (%llx)\n", (t1 - t0) / N * 1e9, (unsigned long long)st[0]);
    }
    return 0;
}
EOC
for cc in "g++" "/opt/rocm/lib/llvm/bin/clang++"; do
  echo "## $cc -O3" >> $out
  $cc -O3 -std=c++17 -I curve25519-dalek_amd/csrc /tmp/kb2.cpp -o /tmp/kb2 && /tmp/kb2 >> $out 2>&1
  $cc -O3 -std=c++17 -I curve25519-dalek_amd/csrc tools/keccak_bench.cpp -o /tmp/kb && /tmp/kb >> $out 2>&1
done
cat $out
