// finish.hip -- the batched finishing kernels: projective results -> canonical bytes with one shared field inversion
// per lane-chunk (Montgomery's trick, field.rs:225-273).  They run at one wave per SIMD (n / 16 lanes), i.e. they are
// bound by the latency of a dependent chain of field operations, not by issue: this translation unit therefore keeps
// the ten independent column sums of fe_mul (C25519_CHAIN 0), while kernels.hip (comb, wide tables, ladder) uses the
// chained-carry form.  Measured: batched compression 0.216 -> 0.194 ms per 2^20 points.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "devio.h"
#include "kernels.h"
#include "selftest.h"

namespace c25519 {

static inline unsigned div_up(u64 a, u64 b) { return (unsigned)((a + b - 1) / b); }

// ================================================================================================
// K3  batched compression (edwards.rs:634-647 compress_batch_alloc): Montgomery's trick
//     (field.rs:225-273) with each lane owning CH projective points: 3 M per point + one field
//     inversion per lane.  Lane t owns points t, t+T, t+2T, ... so a wave's loads stay adjacent.
//     mode 0: Edwards y + sign(x);  mode 2: Montgomery u = (Z+Y)/(Z-Y) (edwards.rs:595-612).
// ================================================================================================
template <int CH>
__global__ void __launch_bounds__(256) k_compress_p32(const u32 *__restrict__ scratch, u32 *__restrict__ prefix, u64 n,
                                                      uint8_t *__restrict__ out) {
    const u64 T = (u64)gridDim.x * blockDim.x, t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    // one wave per SIMD at n = 2^20: nothing else hides the load latency, so every record is fetched one step ahead
    feT acc = fe_one();
    feT Zc = p32_load_z(scratch, t);
    int cnt = 0;
#pragma unroll 1
    for (int j = 0; j < CH; j++) {
        u64 idx = t + (u64)j * T;
        if (idx >= n) break;
        cnt = j + 1;
        const u64 nxt = idx + T;
        feT Zn = (j + 1 < CH && nxt < n) ? p32_load_z(scratch, nxt) : Zc;
        fe48_store(prefix, idx, acc);
        acc = fe_mul(acc, Zc);
        Zc = Zn;
    }
    feT inv = fe_invert(acc);
    u64 idx = t + (u64)(cnt - 1) * T;
    feT Z = p32_load_z(scratch, idx), pre = fe48_load(prefix, idx), X, Y;
    p32_load_xy(scratch, idx, X, Y);
#pragma unroll 1
    for (int j = cnt - 1; j >= 0; j--) {
        const u64 cur = t + (u64)j * T, prv = j > 0 ? cur - T : cur;
        feT Zp = p32_load_z(scratch, prv), prep = fe48_load(prefix, prv), Xp, Yp;
        p32_load_xy(scratch, prv, Xp, Yp);
        feT zi = fe_mul(inv, pre);
        inv = fe_mul(inv, Z);
        u32 w[8];
        ge_affine_compress(fe_mul(X, zi), fe_mul(Y, zi), w);
        store8(out, cur, w);
        Z = Zp; pre = prep; X = Xp; Y = Yp;
    }
}

// ================================================================================================
// batched ratios N_i / D_i -> canonical 32 bytes, with the reference's invert(0) = 0 convention
// (field.rs:225-273: zeros are skipped by Montgomery's trick and stay zero).
//   MODE 0: N = X, D = Z of the P32 record          (X25519 as_affine, montgomery.rs:409: U / W)
//   MODE 1: N = Z + Y, D = Z - Y                    (EdwardsPoint::to_montgomery_batch, edwards.rs:595-612)
// ================================================================================================
template <int CH, int MODE>
__global__ void __launch_bounds__(256) k_ratio_p32(const u32 *__restrict__ scratch, u32 *__restrict__ prefix, u64 n, uint8_t *__restrict__ out) {
    const u64 T = (u64)gridDim.x * blockDim.x, t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    feT acc = fe_one();
#pragma unroll 1
    for (int j = 0; j < CH; j++) {
        u64 idx = t + (u64)j * T;
        if (idx >= n) break;
        feT X, Y, Z = p32_load_z(scratch, idx);
        feT D = Z;
        if (MODE == 1) { p32_load_xy(scratch, idx, X, Y); D = fe_carry(fe_sub(Z, Y)); }
        fe48_store(prefix, idx, acc);
        feT next = fe_mul(acc, D);
        acc = fe_select(next, acc, fe_is_zero(D));
    }
    feT inv = fe_invert(acc);
#pragma unroll 1
    for (int j = CH - 1; j >= 0; j--) {
        u64 idx = t + (u64)j * T;
        if (idx >= n) continue;
        feT X, Y, Z = p32_load_z(scratch, idx);
        p32_load_xy(scratch, idx, X, Y);
        feT D = Z, N = X;
        if (MODE == 1) { D = fe_carry(fe_sub(Z, Y)); N = fe_carry(fe_add(Z, Y)); }
        bool dz = fe_is_zero(D);
        feT dinv = fe_select(fe_mul(inv, fe48_load(prefix, idx)), fe_zero(), dz);
        inv = fe_select(fe_mul(inv, D), inv, dz);
        u32 w[8];
        fe_to_words(fe_mul(N, dinv), w);
        store8(out, idx, w);
    }
}

__global__ void __launch_bounds__(256) k_compress_ristretto(const uint8_t *__restrict__ in_raw, u64 n, uint8_t *__restrict__ out) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    u32 w[8];
    ris_compress(raw160_load(in_raw, idx), w);
    store8(out, idx, w);
}

// compress raw 160-byte points, one inversion per lane (edwards.rs:615); used for small batches
// and as the reference behaviour the batched kernel is tested against.
__global__ void __launch_bounds__(256) k_compress_raw(const uint8_t *__restrict__ in_raw, u64 n, uint8_t *__restrict__ out) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    ge_p3 P = raw160_load(in_raw, idx);
    feT zi = fe_invert(P.Z);
    u32 w[8];
    ge_affine_compress(fe_mul(P.X, zi), fe_mul(P.Y, zi), w);
    store8(out, idx, w);
}
// raw160 -> P32 scratch (so the batched compressor can be reused on caller-supplied points)
__global__ void __launch_bounds__(256) k_raw_to_p32(const uint8_t *__restrict__ in_raw, u64 n, u32 *__restrict__ scratch) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    ge_p3 P = raw160_load(in_raw, idx);
    p32_store(scratch, idx, P.X, P.Y, P.Z);
}


hipError_t launch_selftest_c0(int op, const uint32_t *a, const uint32_t *b, uint64_t n, uint8_t *out, hipStream_t st) {   // ten-column unit
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_selftest_field<0>, dim3(div_up(n, 256)), dim3(256), 0, st, op, a, b, n, out);
    return hipGetLastError();
}

hipError_t launch_compress_p32(const uint32_t *scratch, uint32_t *prefix, u64 n, uint8_t *out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    constexpr int CH = 16;
    u64 threads = (n + CH - 1) / CH;
    unsigned grid = div_up(threads, 256);
    hipLaunchKernelGGL(k_compress_p32<CH>, dim3(grid), dim3(256), 0, st, scratch, prefix, n, out);
    return hipGetLastError();
}

// mode 0: X/Z, mode 1: (Z+Y)/(Z-Y) of the P32 records
hipError_t launch_ratio_p32(int mode, const uint32_t *scratch, uint32_t *prefix, u64 n, uint8_t *out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    constexpr int CH = 16;
    unsigned grid = div_up((n + CH - 1) / CH, 256);
    if (mode == 0) hipLaunchKernelGGL((k_ratio_p32<CH, 0>), dim3(grid), dim3(256), 0, st, scratch, prefix, n, out);
    else hipLaunchKernelGGL((k_ratio_p32<CH, 1>), dim3(grid), dim3(256), 0, st, scratch, prefix, n, out);
    return hipGetLastError();
}

hipError_t launch_compress_ristretto(const uint8_t *in_raw, u64 n, uint8_t *out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_compress_ristretto, dim3(div_up(n, 256)), dim3(256), 0, st, in_raw, n, out);
    return hipGetLastError();
}

hipError_t launch_compress_raw(const uint8_t *in_raw, u64 n, uint8_t *out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_compress_raw, dim3(div_up(n, 256)), dim3(256), 0, st, in_raw, n, out);
    return hipGetLastError();
}

hipError_t launch_raw_to_p32(const uint8_t *in_raw, u64 n, uint32_t *scratch, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_raw_to_p32, dim3(div_up(n, 256)), dim3(256), 0, st, in_raw, n, scratch);
    return hipGetLastError();
}


// (r5; in this translation unit since r6: a lone wave's chain of 252 dependent squarings wants the ten independent column sums of fe_sq here -- a multiply-add every
// ~6 cycles -- not the chained carries of kernels.hip, one dependent chain at ~12 cycles per multiply-add: profiles/r06_small_call_phases.txt)
// verify_batch of at most 128 signatures: the keys A_i and the R_i as ONE block that needs nothing cleared before it and no second launch for B: record 0 = the basepoint, and
// bad[0] / bad[1] (keys / R_i that do not decode) are WRITTEN, not counted into -- what the directly published small path of verify_batch runs while the host
// hashes (verify.hip verify_batch_small_host).  pks / sigs may be page-locked host memory read in place (every byte is read once).
__global__ void __launch_bounds__(256) k_prep_small_verify(const uint8_t *__restrict__ pks, const uint8_t *__restrict__ pk_points, const uint8_t *__restrict__ sigs, u32 n,
                                                           u32 *__restrict__ pts, u32 *__restrict__ bad) {
    const u32 i = threadIdx.x;
    int bad_a = 0, bad_r = 0;
    if (i < 2 * n) {
        const bool is_r = i >= n;
        const u32 j = is_r ? i - n : i;
        if (!is_r && pk_points) {
            // the key's cached point (VerifyingKey, verifying.rs:64-71; any Z): to affine -- one inversion, a chain as long as the decompression beside it
            feT X = raw160_fe(pk_points, j, 0), Y = raw160_fe(pk_points, j, 1);
            if (!raw160_z_is_one(pk_points, j)) {
                const feT zi = fe_invert(raw160_fe(pk_points, j, 2));
                X = fe_mul(X, zi); Y = fe_mul(Y, zi);
            }
            pts_store(pts, (u64)n + 1 + j, X, Y);
        } else {
            u32 w[8];
            load8(is_r ? sigs : pks, is_r ? 2 * (u64)j : (u64)j, w);
            ge_p3 P;
            const bool ok = ge_decompress(P, w);
            pts_store(pts, (u64)(is_r ? 1 : n + 1) + j, P.X, P.Y);
            bad_a = !ok && !is_r; bad_r = !ok && is_r;
        }
    }
    if (i == 255) { const ge_p3 B = ge_basepoint(); pts_store(pts, 0, B.X, B.Y); }       // (the last lane: idle unless n = 128)
    const int ca = __syncthreads_count(bad_a), cr = __syncthreads_count(bad_r);
    if (i == 0) { bad[0] = (u32)ca; bad[1] = (u32)cr; }
}

hipError_t launch_prep_small_verify(const uint8_t *pks, const uint8_t *pk_points, const uint8_t *sigs, uint64_t n, uint32_t *pts, uint32_t *bad, hipStream_t st) {
    if (n == 0 || n > 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_prep_small_verify, dim3(1), dim3(256), 0, st, pks, pk_points, sigs, (u32)n, pts, bad);
    return hipGetLastError();
}


}  // namespace c25519
