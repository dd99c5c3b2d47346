// Diagnostics translation unit of libc25519hip.so: nothing here is on a product path.
//   * the device self-tests -- K1, the FIELD layer as compiled for the GPU (selftest.h; this unit holds the chained-carry
//     flavour, finish.hip the ten-column one), and the SCALAR layer (sc28.h: the arithmetic mod l of verify_batch / sign /
//     per-signature verify) -- raw limbs or words in, canonical bytes out, compared with Python big integers by
//     tests/test_gpu_field.py at 2^20 random values plus the extremes of every bound class;
//   * the instruction-rate probes behind c25519_microbench (bench.py's live v_mad_u64_u32 peak, tools/probes.py).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#define C25519_CHAIN 1
#include "../../include/c25519_hip.h"
#include "devio.h"
#include "kernels.h"
#include "selftest.h"
#include "fe26x.h"
#include "sc28.h"
#include "fe9_probe.h"
#include "ctx.h"

#define EXPORT extern "C" __attribute__((visibility("default")))
#define HIPCHK(call)                                                \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return c25519_fail(ctx, _e, #call);   \
    } while (0)

namespace c25519 {
static inline unsigned div_up(u64 a, u64 b) { return (unsigned)((a + b - 1) / b); }

// ================================================================================================
// device self-test of the scalar arithmetic (sc28.h; reference: u64/scalar.rs:66-320, scalar.rs:248-263)
// ================================================================================================
// a, b: 16 u32 per item; out: 8 words per item.
//   op 0: from_wide(a: 16 words, any 512-bit value)                 -> (a mod l) as words
//   op 1: mul_5x10(a[0..5]: 28-bit LIMBS, b[0..10]: 28-bit LIMBS)    -> a b mod l        (contract: a b < 2^393)
//   op 2: mul(a[0..10] limbs, b[0..10] limbs)                        -> a b mod l        (contract: a b < 2^512)
//   op 3: add(a, b: canonical, 8 words each)   op 4: neg(a: canonical)
//   op 5: words_canonical(a: 8 words) -> word 0 = 1 if a < l          op 6: to_words(from_words(a: 8 words, < 2^256))
//   op 7: the signing composite S = r + k a (single.hip): add(mul(from_wide(a: 16 words), from_words(b[0..8])), from_words(b[8..16]: canonical))
__global__ void __launch_bounds__(256) k_selftest_scalar(int op, const u32 *__restrict__ a, const u32 *__restrict__ b, u64 n, u32 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 x[16], y[16], w[8];
    for (int q = 0; q < 16; q++) { x[q] = a[16 * i + q]; y[q] = b ? b[16 * i + q] : 0u; }
    for (int q = 0; q < 8; q++) w[q] = 0;
    switch (op) {
    case 0: sc28_to_words(sc28_from_wide(x), w); break;
    case 1: sc28_to_words(sc28_mul_5x10(x, y), w); break;
    case 2: sc28_to_words(sc28_mul(x, y), w); break;
    case 3: sc28_to_words(sc28_add(sc28_from_words(x), sc28_from_words(y)), w); break;
    case 4: sc28_to_words(sc28_neg(sc28_from_words(x)), w); break;
    case 5: w[0] = sc28_words_canonical(x) ? 1u : 0u; break;
    case 6: sc28_to_words(sc28_from_words(x), w); break;
    default: sc28_to_words(sc28_add(sc28_mul(sc28_from_wide(x).v, sc28_from_words(y).v), sc28_from_words(y + 8)), w); break;
    }
    for (int q = 0; q < 8; q++) out[8 * i + q] = w[q];
}
hipError_t launch_selftest_scalar(int op, const uint32_t *a, const uint32_t *b, uint64_t n, uint32_t *out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_selftest_scalar, dim3(div_up(n, 256)), dim3(256), 0, st, op, a, b, n, out);
    return hipGetLastError();
}

// ================================================================================================
// Integer-multiplier roofline probes (c25519_microbench)
// ================================================================================================
__global__ void __launch_bounds__(256) k_probe_mad(u32 *out, int iters, u32 seed) {
    u32 b = seed | 1u;
    u64 a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    // (four rounds per trip: with an 8-instruction body the rate depended on where the loop fell relative to an instruction-cache line --
    //  24.2 against 33.1 T/s for the same source after unrelated code moved in this file: profiles/r04_instruction_rates.txt)
#pragma unroll 4
    for (int i = 0; i < iters; i++) {
        a0 = (u64)(u32)a1 * b + a0; a1 = (u64)(u32)a2 * b + a1; a2 = (u64)(u32)a3 * b + a2; a3 = (u64)(u32)a4 * b + a3;
        a4 = (u64)(u32)a5 * b + a4; a5 = (u64)(u32)a6 * b + a5; a6 = (u64)(u32)a7 * b + a6; a7 = (u64)(u32)a0 * b + a7;
    }
    u64 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if ((u32)r == 0x12345678u) out[0] = (u32)(r >> 32);
}
__global__ void __launch_bounds__(256) k_probe_add(u32 *out, int iters, u32 seed) {
    u32 b = seed | 1u;
    u32 a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    for (int i = 0; i < iters; i++) {
        a0 += a1 ^ b; a1 += a2 ^ b; a2 += a3 ^ b; a3 += a4 ^ b; a4 += a5 ^ b; a5 += a6 ^ b; a6 += a7 ^ b; a7 += a0 ^ b;
    }
    u32 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345678u) out[0] = r;
}
// plain two-operand adds (VOP2 encoding)
__global__ void __launch_bounds__(256) k_probe_add2(u32 *out, int iters, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    for (int i = 0; i < iters; i++) {
        a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0;
    }
    u32 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345678u) out[0] = r;
}
// co-issue probe: 8 independent v_mad_u64_u32 chains interleaved with 8*R independent v_xad_u32 chains
template <int R>
__global__ void __launch_bounds__(256) k_probe_mix(u32 *out, int iters, u32 seed) {
    u32 b = seed | 1u;
    u64 a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    u32 c[8 * R];
    for (int j = 0; j < 8 * R; j++) c[j] = threadIdx.x * (2 * j + 3) + out[j & 7];
    for (int i = 0; i < iters; i++) {
        a0 = (u64)(u32)a1 * b + a0; a1 = (u64)(u32)a2 * b + a1; a2 = (u64)(u32)a3 * b + a2; a3 = (u64)(u32)a4 * b + a3;
        a4 = (u64)(u32)a5 * b + a4; a5 = (u64)(u32)a6 * b + a5; a6 = (u64)(u32)a7 * b + a6; a7 = (u64)(u32)a0 * b + a7;
#pragma unroll
        for (int j = 0; j < 8 * R; j++) c[j] += c[(j + 1) % (8 * R)] ^ b;
    }
    u64 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    u32 q = 0;
    for (int j = 0; j < 8 * R; j++) q ^= c[j];
    if ((u32)r == 0x12345678u && q == 0x9abcdef0u) out[0] = (u32)(r >> 32);
}
__global__ void __launch_bounds__(256) k_probe_mullo(u32 *out, int iters, u32 seed) {
    u32 b = seed | 1u;
    u32 a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    for (int i = 0; i < iters; i++) {
        a0 = a1 * b; a1 = a2 * b; a2 = a3 * b; a3 = a4 * b; a4 = a5 * b; a5 = a6 * b; a6 = a7 * b; a7 = a0 * b;
    }
    u32 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345678u) out[0] = r;
}
__global__ void __launch_bounds__(256) k_probe_femul(u32 *out, int iters, u32 seed) {
    feT x, y, z;   // operands come from memory so nothing is known at compile time
    for (int i = 0; i < 10; i++) { x.v[i] = (out[i] + seed + threadIdx.x) & M25; y.v[i] = (out[10 + i] + threadIdx.x) & M25; z.v[i] = ((out[20 + i] ^ seed) + threadIdx.x) & M25; }
    for (int i = 0; i < iters; i++) { x = fe_mul(x, z); y = fe_mul(y, z); }
    u32 r = 0;
    for (int i = 0; i < 10; i++) r ^= x.v[i] ^ y.v[i];
    if (r == 0x12345678u) out[0] = r;
}
__global__ void __launch_bounds__(256) k_probe_fesq(u32 *out, int iters, u32 seed) {
    feT x, y;
    for (int i = 0; i < 10; i++) { x.v[i] = (out[i] + seed + threadIdx.x) & M25; y.v[i] = (out[10 + i] + threadIdx.x) & M25; }
    for (int i = 0; i < iters; i++) { x = fe_sq(x); y = fe_sq(y); }
    u32 r = 0;
    for (int i = 0; i < 10; i++) r ^= x.v[i] ^ y.v[i];
    if (r == 0x12345678u) out[0] = r;
}
// ---- instruction-rate probes (which = 10 ..): ITER x 8 independent chains of ONE instruction, written in asm so that
//      the compiler cannot fuse, reorder or drop them.  They price the non-multiplier half of fe_mul.
#define C25519_PROBE32(NAME, ASM)                                                                       \
    __global__ void __launch_bounds__(256) NAME(u32 *out, int iters, u32 seed) {                        \
        u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        u32 b = seed | 1u;                                                                              \
        for (int i = 0; i < iters; i++) {                                                               \
            asm volatile(ASM : "+v"(a0) : "v"(b)); asm volatile(ASM : "+v"(a1) : "v"(b));               \
            asm volatile(ASM : "+v"(a2) : "v"(b)); asm volatile(ASM : "+v"(a3) : "v"(b));               \
            asm volatile(ASM : "+v"(a4) : "v"(b)); asm volatile(ASM : "+v"(a5) : "v"(b));               \
            asm volatile(ASM : "+v"(a6) : "v"(b)); asm volatile(ASM : "+v"(a7) : "v"(b));               \
        }                                                                                               \
        u32 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                  \
        if (r == 0x12345678u) out[0] = r;                                                               \
    }
#define C25519_PROBE64(NAME, ASM)                                                                       \
    __global__ void __launch_bounds__(256) NAME(u32 *out, int iters, u32 seed) {                        \
        u64 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        u64 b = ((u64)seed << 20) | 1u;                                                                 \
        for (int i = 0; i < iters; i++) {                                                               \
            asm volatile(ASM : "+v"(a0) : "v"(b)); asm volatile(ASM : "+v"(a1) : "v"(b));               \
            asm volatile(ASM : "+v"(a2) : "v"(b)); asm volatile(ASM : "+v"(a3) : "v"(b));               \
            asm volatile(ASM : "+v"(a4) : "v"(b)); asm volatile(ASM : "+v"(a5) : "v"(b));               \
            asm volatile(ASM : "+v"(a6) : "v"(b)); asm volatile(ASM : "+v"(a7) : "v"(b));               \
        }                                                                                               \
        u64 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                  \
        if ((u32)r == 0x12345678u) out[0] = (u32)(r >> 32);                                             \
    }
C25519_PROBE64(k_probe_lshr64, "v_lshrrev_b64 %0, 26, %0")
C25519_PROBE64(k_probe_lshladd64, "v_lshl_add_u64 %0, %0, 0, %1")
C25519_PROBE32(k_probe_alignbit, "v_alignbit_b32 %0, %1, %0, 26")
C25519_PROBE32(k_probe_and, "v_and_b32_e32 %0, %1, %0")
C25519_PROBE32(k_probe_lshl, "v_lshlrev_b32_e32 %0, 1, %0")
C25519_PROBE32(k_probe_andor, "v_and_or_b32 %0, %0, %1, %1")
C25519_PROBE32(k_probe_mul24, "v_mul_u32_u24_e32 %0, %1, %0")
C25519_PROBE32(k_probe_mad24, "v_mad_u32_u24 %0, %0, %1, %1")
C25519_PROBE32(k_probe_mulhi, "v_mul_hi_u32 %0, %0, %1")
C25519_PROBE32(k_probe_add3, "v_add3_u32 %0, %0, %1, %1")
C25519_PROBE32(k_probe_bfe, "v_bfe_u32 %0, %0, 3, 26")
C25519_PROBE32(k_probe_lshladd32, "v_lshl_add_u32 %0, %0, 1, %1")
C25519_PROBE32(k_probe_lshr, "v_lshrrev_b32_e32 %0, 3, %0")
C25519_PROBE32(k_probe_sub, "v_sub_u32_e32 %0, %1, %0")
C25519_PROBE32(k_probe_or, "v_or_b32_e32 %0, %1, %0")
C25519_PROBE32(k_probe_xor, "v_xor_b32_e32 %0, %1, %0")
C25519_PROBE32(k_probe_cndmask, "v_cndmask_b32_e32 %0, %1, %0, vcc")
C25519_PROBE32(k_probe_mov, "v_mov_b32_e32 %0, %1")
// the select as k_accumulate / the constant-time scan issue it (VOP3, condition in an SGPR pair), and the arithmetic alternative for a conditional
// swap (xor / and / xor on a lane mask: three full-rate instructions for the two selects of a pair)
__global__ void __launch_bounds__(256) k_probe_cndmask_e64(u32 *out, int iters, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    u32 b = seed | 1u;
    const unsigned long long m = __ballot((threadIdx.x & 3u) != 0u);
    for (int i = 0; i < iters; i++) {
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "s"(m)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "s"(m));
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "s"(m)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "s"(m));
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "s"(m)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "s"(m));
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "s"(m)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "s"(m));
    }
    u32 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345678u) out[0] = r;
}
// the e32 form again, but with VCC written once before the loop by a real compare (the generic probe leaves VCC as the prologue left it)
__global__ void __launch_bounds__(256) k_probe_cndmask_vcc(u32 *out, int iters, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    u32 b = seed | 1u;
    asm volatile("v_cmp_ne_u32_e32 vcc, 0, %0" :: "v"(threadIdx.x & 3u) : "vcc");
    for (int i = 0; i < iters; i++) {
        asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b)); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b));
        asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b)); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b));
        asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a4) : "v"(b)); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a5) : "v"(b));
        asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a6) : "v"(b)); asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a7) : "v"(b));
    }
    u32 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345678u) out[0] = r;
}
C25519_PROBE32(k_probe_perm, "v_perm_b32 %0, %0, %1, %1")
C25519_PROBE32(k_probe_add_sdwa, "v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
C25519_PROBE32(k_probe_lshlor, "v_lshl_or_b32 %0, %0, 6, %1")
// a 64-bit add as a carry pair on 32-bit halves (8 independent pairs per iteration; one pair counts as one operation)
__global__ void __launch_bounds__(256) k_probe_addco(u32 *out, int iters, u32 seed) {
    u32 lo[8], hi[8], b = seed | 1u;
    for (int j = 0; j < 8; j++) { lo[j] = threadIdx.x * (2 * j + 3) + seed; hi[j] = out[j] + threadIdx.x; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            asm volatile("v_add_co_u32_e32 %0, vcc, %2, %0\n\tv_addc_co_u32_e32 %1, vcc, %2, %1, vcc" : "+v"(lo[j]), "+v"(hi[j]) : "v"(b) : "vcc");
    }
    u32 r = 0;
    for (int j = 0; j < 8; j++) r ^= lo[j] ^ hi[j];
    if (r == 0x12345678u) out[0] = r;
}
// R full-rate adds issued beside every v_mad_u64_u32: does the adder run while the multiplier is busy?
template <int R>
__global__ void __launch_bounds__(256) k_probe_mix_add(u32 *out, int iters, u32 seed) {
    u32 b = seed | 1u;
    u64 a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    u32 c[8];
    for (int j = 0; j < 8; j++) c[j] = threadIdx.x * (2 * j + 3) + out[j & 7];
    for (int i = 0; i < iters; i++) {
#define C25519_MIXSTEP(A, C)                                                                                   \
        asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %1, %0" : "+v"(A) : "v"(b) : "s6", "s7");                   \
        for (int r = 0; r < R; r++) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(C) : "v"(b));
        C25519_MIXSTEP(a0, c[0]) C25519_MIXSTEP(a1, c[1]) C25519_MIXSTEP(a2, c[2]) C25519_MIXSTEP(a3, c[3])
        C25519_MIXSTEP(a4, c[4]) C25519_MIXSTEP(a5, c[5]) C25519_MIXSTEP(a6, c[6]) C25519_MIXSTEP(a7, c[7])
#undef C25519_MIXSTEP
    }
    u64 r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    u32 q = 0;
    for (int j = 0; j < 8; j++) q ^= c[j];
    if ((u32)r == 0x12345678u && q == 0x9abcdef0u) out[0] = (u32)(r >> 32);
}
// one DEPENDENT chain of v_mad_u64_u32 per lane (the accumulator feeds the next multiply-add): its latency
__global__ void __launch_bounds__(256) k_probe_mad_dep(u32 *out, int iters, u32 seed) {
    u64 a = threadIdx.x + seed;
    u32 b = seed | 1u, c = threadIdx.x | 3u;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "s6", "s7");
        asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "s6", "s7");
        asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "s6", "s7");
        asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "s6", "s7");
        asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "s6", "s7");
        asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "s6", "s7");
        asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "s6", "s7");
        asm volatile("v_mad_u64_u32 %0, s[6:7], %1, %2, %0" : "+v"(a) : "v"(b), "v"(c) : "s6", "s7");
    }
    if ((u32)a == 0x12345678u) out[0] = (u32)(a >> 32);
}
// three independent products per iteration, both operands changing (19 g and 2 f are recomputed like in the point formulas):
// MODE 0: three calls of this unit's fe_mul (chained); 1: fe_mul_chain_n<3> (the same chains in lockstep); 2: three ten-column products
template <int MODE>
__global__ void __launch_bounds__(256) k_probe_femul3(u32 *out, int iters, u32 seed) {
    feT x[3], y[3];
    _Pragma("unroll") for (int n = 0; n < 3; n++) for (int i = 0; i < 10; i++) { x[n].v[i] = (out[i] + seed + threadIdx.x + n) & M25; y[n].v[i] = (out[10 + i] + threadIdx.x * (n + 2)) & M25; }
    for (int it = 0; it < iters; it++) {
        feT r[3];
#if defined(__HIP_DEVICE_COMPILE__)
        if (MODE == 1) {
            feW f[3]; feL g[3];
            _Pragma("unroll") for (int n = 0; n < 3; n++) { f[n] = x[n]; g[n] = y[n]; }
            fe_mul_chain_n<3>(r, f, g);
        } else if (MODE == 2) {
            _Pragma("unroll") for (int n = 0; n < 3; n++) r[n] = fe_mul_cols_g(x[n], y[n]);
        } else
#endif
        {
            _Pragma("unroll") for (int n = 0; n < 3; n++) r[n] = fe_mul(x[n], y[n]);
        }
        _Pragma("unroll") for (int n = 0; n < 3; n++) { x[n] = y[n]; y[n] = r[n]; }
    }
    u32 q = 0;
    _Pragma("unroll") for (int n = 0; n < 3; n++) for (int i = 0; i < 10; i++) q ^= x[n].v[i] ^ y[n].v[i];
    if (q == 0x12345678u) out[0] = q;
}
// The reference's literal layout: 5 x u64 limbs, u128 products (u64/field.rs:111-214) -- the A/B arm.
struct fe51 { u64 v[5]; };
__device__ __forceinline__ fe51 fe51_mul(const fe51 &x, const fe51 &y) {
    typedef unsigned __int128 u128;
    const u64 *a = x.v, *b = y.v;
    const u64 mask = (1ull << 51) - 1;
    u64 b1 = b[1] * 19, b2 = b[2] * 19, b3 = b[3] * 19, b4 = b[4] * 19;
    u128 c0 = (u128)a[0] * b[0] + (u128)a[4] * b1 + (u128)a[3] * b2 + (u128)a[2] * b3 + (u128)a[1] * b4;
    u128 c1 = (u128)a[1] * b[0] + (u128)a[0] * b[1] + (u128)a[4] * b2 + (u128)a[3] * b3 + (u128)a[2] * b4;
    u128 c2 = (u128)a[2] * b[0] + (u128)a[1] * b[1] + (u128)a[0] * b[2] + (u128)a[4] * b3 + (u128)a[3] * b4;
    u128 c3 = (u128)a[3] * b[0] + (u128)a[2] * b[1] + (u128)a[1] * b[2] + (u128)a[0] * b[3] + (u128)a[4] * b4;
    u128 c4 = (u128)a[4] * b[0] + (u128)a[3] * b[1] + (u128)a[2] * b[2] + (u128)a[1] * b[3] + (u128)a[0] * b[4];
    fe51 o;
    c1 += (u64)(c0 >> 51); o.v[0] = (u64)c0 & mask;
    c2 += (u64)(c1 >> 51); o.v[1] = (u64)c1 & mask;
    c3 += (u64)(c2 >> 51); o.v[2] = (u64)c2 & mask;
    c4 += (u64)(c3 >> 51); o.v[3] = (u64)c3 & mask;
    u64 carry = (u64)(c4 >> 51); o.v[4] = (u64)c4 & mask;
    o.v[0] += carry * 19; o.v[1] += o.v[0] >> 51; o.v[0] &= mask;
    return o;
}
__global__ void __launch_bounds__(256) k_probe_femul51(u32 *out, int iters, u32 seed) {
    fe51 x, y, z;
    for (int i = 0; i < 5; i++) { x.v[i] = ((u64)out[i] << 19 | threadIdx.x) + seed; y.v[i] = ((u64)out[5 + i] << 19) + threadIdx.x; z.v[i] = ((u64)out[10 + i] << 19) ^ seed; }
    for (int i = 0; i < iters; i++) { x = fe51_mul(x, z); y = fe51_mul(y, z); }
    u64 r = 0;
    for (int i = 0; i < 5; i++) r ^= x.v[i] ^ y.v[i];
    if ((u32)r == 0x12345678u) out[0] = (u32)(r >> 32);
}

// ---- ds_bpermute_b32 under different SELECTOR patterns (which = 50 .. 57: eight independent permutes per trip; 60 .. 67: eight DEPENDENT ones, i.e.
//      latency) -- the measurement behind the cross-lane table fetch of the constant-time fixed base (kernels.hip k_mul_base_ctp): does the duration
//      of a permute depend on which lanes the lanes pull from?  Patterns (source lane of lane l, group = 32-lane half):
//        0 identity   1 all lanes pull lane 5   2 pseudo-random inside the own 32-lane half (what k_mul_base_ctp<5> issues)
//        3 pairs 32 lanes apart inside a half: lanes alternate between s and s + 32 (same bank, different lane, if the crossbar had 32 banks)
//        4 pseudo-random over the whole wave   5 two sources only (lane 0 / lane 32)   6 rotate by one   7 pseudo-random, new selectors every trip
//        8 .. 12 (r6, ADVICE r5: which = 80 .. 84 / 85 .. 89) MANY-TO-ONE inside a half: groups of k = 2, 4, 8, 16, 32 lanes pull the same (scattered) source lane --
//        what equal digits in neighbouring lanes make of the selector
template <int PAT, bool DEP>
__global__ void __launch_bounds__(256) k_probe_bpermute(u32 *out, int iters, u32 seed) {
    const u32 lane = threadIdx.x & 63u;
    u32 h = (lane * 2654435761u) ^ (blockIdx.x * 40503u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    u32 src = lane;
    if (PAT == 1) src = 5;
    else if (PAT == 2 || PAT == 7) src = (lane & 32u) | (h & 31u);
    else if (PAT == 3) src = ((lane & 1u) << 5) | ((lane >> 1) & 31u);
    else if (PAT == 4) src = h & 63u;
    else if (PAT == 5) src = (lane & 1u) << 5;
    else if (PAT == 6) src = (lane + 1u) & 63u;
    else if (PAT >= 8) { const u32 k = 2u << (PAT - 8); src = (lane & 32u) | ((((lane & 31u) / k) * 7u + 3u) & 31u); }
    int sel = (int)(src << 2);
    int a0 = (int)(threadIdx.x + seed), a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    for (int i = 0; i < iters; i++) {
        if (DEP) {
            asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(sel));
            asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(sel));
            asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(sel));
            asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(sel));
            asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(sel));
            asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(sel));
            asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(sel));
            asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a0) : "v"(sel));
        } else {
            asm volatile("ds_bpermute_b32 %0, %8, %0\n\tds_bpermute_b32 %1, %8, %1\n\tds_bpermute_b32 %2, %8, %2\n\tds_bpermute_b32 %3, %8, %3\n\t"
                         "ds_bpermute_b32 %4, %8, %4\n\tds_bpermute_b32 %5, %8, %5\n\tds_bpermute_b32 %6, %8, %6\n\tds_bpermute_b32 %7, %8, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sel));
        }
        // (every pattern runs the same selector arithmetic, so that the rows differ by the selectors only; pattern 7 alone USES the new one)
        h = h * 1664525u + 1013904223u;
        const int nsel = (int)(((lane & 32u) | ((h >> 9) & 31u)) << 2);
        asm volatile("" :: "v"(nsel));
        if (PAT == 7) sel = nsel;
    }
    const int r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345678) out[0] = (u32)r;
}
// 9-limb field products (which = 70, 71): fe9_probe.h
template <bool SQ>
__global__ void __launch_bounds__(256) k_probe_fe9(u32 *out, int iters, u32 seed) {
    fe9 x, y, z;
    for (int i = 0; i < 9; i++) { x.v[i] = (out[i] + seed + threadIdx.x) & ((1u << 28) - 1u); y.v[i] = (out[10 + i] + threadIdx.x) & ((1u << 28) - 1u); z.v[i] = ((out[20 + i] ^ seed) + threadIdx.x) & ((1u << 28) - 1u); }
    for (int i = 0; i < iters; i++) {
        if (SQ) { x = fe9_sq(x); y = fe9_sq(y); } else { x = fe9_mul(x, z); y = fe9_mul(y, z); }
    }
    u32 r = 0;
    for (int i = 0; i < 9; i++) r ^= x.v[i] ^ y.v[i];
    if (r == 0x12345678u) out[0] = r;
}
hipError_t launch_selftest_c1(int op, const uint32_t *a, const uint32_t *b, uint64_t n, uint8_t *out, hipStream_t st) {   // chained-carry unit
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_selftest_field<1>, dim3(div_up(n, 256)), dim3(256), 0, st, op, a, b, n, out);
    return hipGetLastError();
}

hipError_t launch_probe(int which, uint32_t *out, int iters, unsigned grid, hipStream_t st) {
    switch (which) {
    case 0: hipLaunchKernelGGL(k_probe_mad, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 1: hipLaunchKernelGGL(k_probe_femul, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 2: hipLaunchKernelGGL(k_probe_fesq, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 3: hipLaunchKernelGGL(k_probe_femul51, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 4: hipLaunchKernelGGL(k_probe_add, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 5: hipLaunchKernelGGL(k_probe_mullo, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 8: hipLaunchKernelGGL(k_probe_add2, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 6: hipLaunchKernelGGL(k_probe_mix<1>, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 7: hipLaunchKernelGGL(k_probe_mix<2>, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 10: hipLaunchKernelGGL(k_probe_lshr64, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 11: hipLaunchKernelGGL(k_probe_lshladd64, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 12: hipLaunchKernelGGL(k_probe_alignbit, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 13: hipLaunchKernelGGL(k_probe_and, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 14: hipLaunchKernelGGL(k_probe_lshl, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 15: hipLaunchKernelGGL(k_probe_andor, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 16: hipLaunchKernelGGL(k_probe_mul24, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 17: hipLaunchKernelGGL(k_probe_mad24, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 18: hipLaunchKernelGGL(k_probe_mulhi, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 19: hipLaunchKernelGGL(k_probe_add3, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 20: hipLaunchKernelGGL(k_probe_bfe, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 21: hipLaunchKernelGGL(k_probe_lshladd32, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 22: hipLaunchKernelGGL(k_probe_mad_dep, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 23: hipLaunchKernelGGL(k_probe_lshr, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 24: hipLaunchKernelGGL(k_probe_sub, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 25: hipLaunchKernelGGL(k_probe_or, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 26: hipLaunchKernelGGL(k_probe_xor, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 27: hipLaunchKernelGGL(k_probe_cndmask, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 28: hipLaunchKernelGGL(k_probe_mov, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 29: hipLaunchKernelGGL(k_probe_perm, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 30: hipLaunchKernelGGL(k_probe_add_sdwa, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 31: hipLaunchKernelGGL(k_probe_lshlor, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 36: hipLaunchKernelGGL(k_probe_cndmask_e64, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 37: hipLaunchKernelGGL(k_probe_cndmask_vcc, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 32: hipLaunchKernelGGL(k_probe_addco, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 40: hipLaunchKernelGGL(k_probe_femul3<0>, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 41: hipLaunchKernelGGL(k_probe_femul3<1>, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 42: hipLaunchKernelGGL(k_probe_femul3<2>, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 33: hipLaunchKernelGGL(k_probe_mix_add<1>, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 34: hipLaunchKernelGGL(k_probe_mix_add<2>, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 35: hipLaunchKernelGGL(k_probe_mix_add<3>, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
#define C25519_BP(PAT) \
    case 50 + PAT: hipLaunchKernelGGL((k_probe_bpermute<PAT, false>), dim3(grid), dim3(256), 0, st, out, iters, 12345u); break; \
    case 60 + PAT: hipLaunchKernelGGL((k_probe_bpermute<PAT, true>), dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    C25519_BP(0) C25519_BP(1) C25519_BP(2) C25519_BP(3) C25519_BP(4) C25519_BP(5) C25519_BP(6) C25519_BP(7)
#undef C25519_BP
#define C25519_BQ(I) \
    case 80 + I: hipLaunchKernelGGL((k_probe_bpermute<8 + I, false>), dim3(grid), dim3(256), 0, st, out, iters, 12345u); break; \
    case 85 + I: hipLaunchKernelGGL((k_probe_bpermute<8 + I, true>), dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    C25519_BQ(0) C25519_BQ(1) C25519_BQ(2) C25519_BQ(3) C25519_BQ(4)
#undef C25519_BQ
    case 70: hipLaunchKernelGGL(k_probe_fe9<false>, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    case 71: hipLaunchKernelGGL(k_probe_fe9<true>, dim3(grid), dim3(256), 0, st, out, iters, 12345u); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
}  // namespace c25519

using namespace c25519;
// ---- diagnostics -------------------------------------------------------------------------------------
// field self-test: raw limbs (n x 10 u32, HOST pointers; b may be NULL for the unary ops) -> n x 32 canonical bytes
EXPORT int32_t c25519_selftest_field(c25519_ctx *ctx, int op, int chain, const uint32_t *a_limbs, const uint32_t *b_limbs, uint64_t n, uint8_t *out) {
    HIPCHK(hipSetDevice(ctx->device));
    if (op < 0 || op > 11 || (chain != 0 && chain != 1)) { ctx->err = "selftest_field: bad op / chain"; return -(int32_t)hipErrorInvalidValue; }
    if (n == 0) return C25519_OK;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, n * 40)) || (r = ctx_reserve(ctx, ctx->tmp_b, n * 40)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 32))) return r;
    HIPCHK(hipMemcpyAsync(ctx->tmp_a.p, a_limbs, n * 40, hipMemcpyHostToDevice, ctx->stream));
    if (b_limbs) HIPCHK(hipMemcpyAsync(ctx->tmp_b.p, b_limbs, n * 40, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t *db = b_limbs ? (const uint32_t *)ctx->tmp_b.p : nullptr;
    if (chain) HIPCHK(launch_selftest_c1(op, (const uint32_t *)ctx->tmp_a.p, db, n, (uint8_t *)ctx->tmp_c.p, ctx->stream));
    else HIPCHK(launch_selftest_c0(op, (const uint32_t *)ctx->tmp_a.p, db, n, (uint8_t *)ctx->tmp_c.p, ctx->stream));
    HIPCHK(hipMemcpyAsync(out, ctx->tmp_c.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return C25519_OK;
}

EXPORT double c25519_microbench(c25519_ctx *ctx, int which, int iters) {
    if (hipSetDevice(ctx->device) != hipSuccess) return -1.0;
    if (ctx_reserve(ctx, ctx->tmp_a, 4096)) return -1.0;
    hipMemsetAsync(ctx->tmp_a.p, 0x5a, 4096, ctx->stream);
    unsigned grid = (unsigned)ctx->num_cus * 8;   // 8 blocks x 4 waves per CU = 8 waves per SIMD
    if (which >= 200) { which -= 200; grid = (unsigned)ctx->num_cus * 3; }   // which + 200: THREE waves per SIMD (the occupancy of k_accumulate)
    else if (which >= 100) { which -= 100; grid = (unsigned)ctx->num_cus; }   // which + 100: ONE wave per SIMD (latency, not throughput)
    if (launch_probe(which, (uint32_t *)ctx->tmp_a.p, 16, grid, ctx->stream) != hipSuccess) return -1.0;  // warm-up
    hipEventRecord(ctx->ev0, ctx->stream);
    if (launch_probe(which, (uint32_t *)ctx->tmp_a.p, iters, grid, ctx->stream) != hipSuccess) return -1.0;
    hipEventRecord(ctx->ev1, ctx->stream);
    float ms = c25519_last_kernel_ms(ctx);
    if (ms <= 0) return -1.0;
    // 6, 7: the mixed probes count their v_mad_u64_u32 only (8 per iteration), so the result reads as
    // "MAC rate with R simple integer ops issued beside every MAC"
    double per_lane = (which >= 40 && which <= 42) ? 3.0 * iters : (which >= 70 && which <= 71) ? 2.0 * iters : (which == 0 || which >= 4) ? 8.0 * iters : 2.0 * iters;
    double total = per_lane * 256.0 * grid;
    return total / (ms * 1e-3) / 1e9;
}

// scalar self-test: n x 16 u32 per operand (HOST pointers; b may be NULL for the unary ops) -> n x 32 bytes (see k_selftest_scalar)
EXPORT int32_t c25519_selftest_scalar(c25519_ctx *ctx, int op, const uint32_t *a_words, const uint32_t *b_words, uint64_t n, uint8_t *out) {
    HIPCHK(hipSetDevice(ctx->device));
    if (op < 0 || op > 7) { ctx->err = "selftest_scalar: bad op"; return -(int32_t)hipErrorInvalidValue; }
    if (n == 0) return C25519_OK;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, n * 64)) || (r = ctx_reserve(ctx, ctx->tmp_b, n * 64)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 32))) return r;
    HIPCHK(hipMemcpyAsync(ctx->tmp_a.p, a_words, n * 64, hipMemcpyHostToDevice, ctx->stream));
    if (b_words) HIPCHK(hipMemcpyAsync(ctx->tmp_b.p, b_words, n * 64, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(launch_selftest_scalar(op, (const uint32_t *)ctx->tmp_a.p, b_words ? (const uint32_t *)ctx->tmp_b.p : nullptr, n, (uint32_t *)ctx->tmp_c.p, ctx->stream));
    HIPCHK(hipMemcpyAsync(out, ctx->tmp_c.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return C25519_OK;
}
