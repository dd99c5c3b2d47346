// Interleaved chained field multiplications (device only): N INDEPENDENT products computed column by column in lockstep.
//
// The chained form of fe_mul (fe26.h: the carry out of column k is the 64-bit addend of the first multiply of column
// k+1) saves nine 64-bit adds per product, but makes a product ONE dependent chain of 100 v_mad_u64_u32 (latency ~12
// cycles each against 4.7 of issue): a kernel with three waves per SIMD cannot cover it, and the form measured neutral
// in k_accumulate.  A mixed addition multiplies three independent pairs and then four: issuing their chains in lockstep
// gives the scheduler N independent accumulators at every step while keeping the chained carries, and only the current
// column of each product is live (N x (2 + 10) registers instead of 20 accumulators).  2 f_i is a full-rate v_add_u32
// (fe_x2).
#pragma once
#include "fe26.h"
#include "ge26.h"

#if defined(__HIP_DEVICE_COMPILE__)
namespace c25519 {

template <int N>
__device__ __forceinline__ void fe_mul_chain_n(feT (&r)[N], const feW (&f)[N], const feL (&g)[N]) {
    u32 g19[N][10], f2[N][10];
#pragma unroll
    for (int n = 0; n < N; n++) {
        C25519_BOUND(f[n].v, W_EVEN, W_ODD, "fe_mul_chain_n f");
        C25519_BOUND(g[n].v, L_EVEN, L_ODD, "fe_mul_chain_n g");
#pragma unroll
        for (int i = 1; i < 10; i++) g19[n][i] = 19u * g[n].v[i];
#pragma unroll
        for (int i = 1; i < 10; i += 2) f2[n][i] = fe_x2(f[n].v[i]);
    }
    u64 h[N], carry[N];
#pragma unroll
    for (int k = 0; k < 10; k++) {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            const bool wrapped = (i + j) >= 10, twice = (i & 1) && (j & 1);
#pragma unroll
            for (int n = 0; n < N; n++) {
                const u32 a = twice ? f2[n][i] : f[n].v[i], b = wrapped ? g19[n][j] : g[n].v[j];
                if (i == 0) h[n] = (k == 0 ? 0ull : carry[n]) + (u64)a * (u64)b;
                else h[n] += (u64)a * (u64)b;
                // (pins: a DEFINITION for three and four products -- as an input (fe26.h C25519_PIN has the reason) k_accumulate spills 21 scratch accesses per addition at its
                //  168 registers, and in lockstep the other products' instructions cover most of the hazard distance anyway: 52 s_nop per 708 products; an INPUT for the pairs of
                //  the mid path (mid_long.h): k_mid_acc_long 168 registers + 48 bytes of scratch -> 148 and none, 94 -> ~50 s_nop per addition)
#ifdef C25519_PAIR_PIN_DEF      // A/B arm (tools/build_variant.sh)
                asm volatile("" : "+v"(h[n]));
#else
                if (N <= 2) asm volatile("" :: "v"(h[n])); else asm volatile("" : "+v"(h[n]));
#endif
            }
        }
#pragma unroll
        for (int n = 0; n < N; n++) {
            r[n].v[k] = (u32)h[n] & ((k & 1) ? M25 : M26);
            carry[n] = h[n] >> ((k & 1) ? 25 : 26);
        }
    }
#pragma unroll
    for (int n = 0; n < N; n++) {
        u64 t = (u64)r[n].v[0] + 19ull * carry[n];
        r[n].v[0] = (u32)t & M26;
        r[n].v[1] += (u32)(t >> 26);
        C25519_BOUND(r[n].v, T_EVEN, T_ODD, "fe_mul_chain_n r");
    }
}

// the ten-column form, written the same way (one product): the A/B arm of the probes
__device__ __forceinline__ feT fe_mul_cols_g(const feW &f, const feL &g) {
    u32 g19[10], f2[10];
#pragma unroll
    for (int i = 1; i < 10; i++) g19[i] = 19u * g.v[i];
#pragma unroll
    for (int i = 1; i < 10; i += 2) f2[i] = 2u * f.v[i];
    u64 h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        h[k] = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            const bool wrapped = (i + j) >= 10, twice = (i & 1) && (j & 1);
            h[k] += (u64)(twice ? f2[i] : f.v[i]) * (u64)(wrapped ? g19[j] : g.v[j]);
        }
    }
    return fe_carry64(h);
}

// ge_madd_signed_p3 (ge26.h) with its seven products issued as one lockstep group of three and one of four
__device__ __forceinline__ ge_p3 ge_madd_signed_p3_lockstep(const ge_p3 &p, const ge_aniels &q, bool neg) {
    feW f3[3]; feL g3[3]; feT r3[3];
    f3[0] = fe_add(p.Y, p.X); f3[1] = fe_sub(p.Y, p.X); f3[2] = p.T;
    const lanemask nm = lane_mask(neg);
#pragma unroll
    for (int i = 0; i < 10; i++) { g3[0].v[i] = sel_u32(q.ypx.v[i], q.ymx.v[i], nm); g3[1].v[i] = sel_u32(q.ymx.v[i], q.ypx.v[i], nm); }
    g3[2] = q.xy2d;
    fe_mul_chain_n<3>(r3, f3, g3);
    const feT &PP = r3[0], &MM = r3[1], &TT = r3[2];
    feL Z2 = fe_twice(p.Z);
    feL X = fe_sub(PP, MM), Y = fe_add(PP, MM);
    feL zp = fe_add_lt(Z2, TT);
    feW zm = fe_sub_w(Z2, TT);
    feW f4[4]; feL g4[4]; feT r4[4];
#pragma unroll
    for (int i = 0; i < 10; i++) { f4[0].v[i] = sel_u32(zm.v[i], zp.v[i], nm); f4[1].v[i] = sel_u32(zp.v[i], zm.v[i], nm); }
    f4[2] = zm; f4[3] = X;
    g4[0] = X; g4[1] = Y; g4[2] = zp; g4[3] = Y;
    fe_mul_chain_n<4>(r4, f4, g4);
    ge_p3 r;
    r.X = r4[0]; r.Y = r4[1]; r.Z = r4[2]; r.T = r4[3];
    return r;
}

// (r6, last) The same addition with the sign handled LAZILY on the accumulator: the caller keeps, per lane, whether the stored point is the true bucket sum or its negative
// (P + sQ = s (sP + Q)), and `flip` (all ones / zero per lane) says that the stored point must change sides before Q -- as it is -- is added.  -P = (-X, Y, Z, -T): two
// conditional negations as x ^ m + (c & m) (two cheap instructions per limb: ~100 issue cycles) where the selects on the record's coordinates and on the two sums cost 40
// v_cndmask_b32 (~185).  The negated coordinates are loose, so Y +- X come out wide: the class fe_mul takes as its first operand anyway.
__device__ __forceinline__ ge_p3 ge_madd_lazy_p3_lockstep(const ge_p3 &p, const ge_aniels &q, u32 flip) {
    const feL Xs = fe_cond_neg(p.X, flip), Ts = fe_cond_neg(p.T, flip);
    feW f3[3]; feL g3[3]; feT r3[3];
    f3[0] = fe_add_w(feL(p.Y), Xs); f3[1] = fe_sub_w(feL(p.Y), Xs); f3[2] = feW(Ts);
    g3[0] = q.ypx; g3[1] = q.ymx; g3[2] = q.xy2d;
    fe_mul_chain_n<3>(r3, f3, g3);
    const feT &PP = r3[0], &MM = r3[1], &TT = r3[2];
    feL Z2 = fe_twice(p.Z);
    feL X = fe_sub(PP, MM), Y = fe_add(PP, MM);
    feL zp = fe_add_lt(Z2, TT);
    feW zm = fe_sub_w(Z2, TT);
    feW f4[4]; feL g4[4]; feT r4[4];
    f4[0] = zm; f4[1] = feW(zp); f4[2] = zm; f4[3] = feW(X);
    g4[0] = X; g4[1] = Y; g4[2] = zp; g4[3] = Y;
    fe_mul_chain_n<4>(r4, f4, g4);
    ge_p3 r;
    r.X = r4[0]; r.Y = r4[1]; r.Z = r4[2]; r.T = r4[3];
    return r;
}

}  // namespace c25519
#endif
