// Variable-time multiscalar multiplication (Pippenger bucket method) and ed25519 verify_batch
// for gfx950.
//
// Reference algorithm: backend/serial/scalar_mul/pippenger.rs:67-160 (signed radix-2^w digits,
// buckets, running-sum bucket reduction, Horner fold over the digit columns).  The GPU version keeps
// the algorithm and re-derives its schedule for a machine with 256 CUs and no cheap scatter-add:
//
//   prep      every point -> affine Niels (y+x, y-x, 2dxy) as a 128-byte limb record (one cache line per gather, nothing to
//             unpack), so bucket accumulation is the 7 M mixed addition (curve_models.rs:455) instead of the reference's
//             8 M re-addition; the sign of a digit is an operand swap in ge_madd_signed_p3
//   digits    s' = s + sum_k HALF_k*2^(pos_k)  makes the signed digit of every window independent:
//             d_k = window_k(s') - HALF_k; the 253 bits of a reduced scalar are shared out evenly over the windows
//             (msm_geom), the top content window is unsigned, bits 253..255 get an (empty) window of their own
//   sort      per window, counting sort of the term indices by bucket (the scatter-add "buckets[b] += P" of
//             pippenger.rs:122-136 becomes gather lists): two-pass partition sort through LDS for wide windows,
//             one-pass LDS histogram + sliced scatter for small inputs
//   order     buckets sorted by list length, so that the lanes of a wave walk lists of equal length
//   accumulate one lane per (window, bucket): sequential mixed additions over its gather list, next point and the
//             index after it in flight; lists longer than LONG_CAP go to a wave-cooperative path on the second stream
//   reduce    sum_b (b+1) B_b in two launches: serial running sums over 8 buckets per lane (pippenger.rs:146-151), then
//             wave-wide weighted sums through shuffles (one wave per 512 buckets, then one wave per window)
//   fold      total.mul_by_pow_2(w_k) + column (pippenger.rs:159) over the window sums: on the host, through the
//             same ge26.h formulas (a serial chain of ~250 doublings is a latency-bound tail that a single CPU core
//             finishes faster than a single GPU lane), ONCE per call: passes leave their column sums in device slots
//   passes    inputs beyond 2.6 M terms are cut into passes of <= 1.75 M terms (the multi-GPU decomposition, in time),
//             enqueued back to back on two stream sets by one host thread; no pass waits for the host
//
// Window width c is chosen per call from n (reference: w = 6/7/8, pippenger.rs:81-87).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/c25519_hip.h"
#include "devio.h"
#include "sc_sha.h"
#include "sc28.h"
#include "kernels.h"
#include "ctx.h"
#include "msm_internal.h"
#include "ffi.h"
#include <functional>

using namespace c25519;
#define EXPORT extern "C" __attribute__((visibility("default")))
#define HIPCHK(call)                                                \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return c25519_fail(ctx, _e, #call);   \
    } while (0)

// Wave priorities (s_setprio): the issue arbiter of a SIMD takes the highest priority first and the oldest wave within it.
// k_accumulate and the decompression run for a millisecond with every wave always ready to issue, and the short
// latency-bound kernels of the NEXT pass (digits, sort, hash chain) that share the SIMDs with them are younger: at equal
// priority they only get the issue slots the old waves leave.  They are the critical path, so they go first.
#ifndef C25519_PRIO
#define C25519_PRIO 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && C25519_PRIO
#define C25519_PRIO_CHAIN() __builtin_amdgcn_s_setprio(3)
#define C25519_PRIO_SIDE() __builtin_amdgcn_s_setprio(2)
#ifndef C25519_PRIO_LONG_LEVEL
#define C25519_PRIO_LONG_LEVEL 2
#endif
#define C25519_PRIO_LONG() __builtin_amdgcn_s_setprio(C25519_PRIO_LONG_LEVEL)
#else
#define C25519_PRIO_LONG() do { } while (0)
#define C25519_PRIO_CHAIN() do { } while (0)
#define C25519_PRIO_SIDE() do { } while (0)
#endif

namespace c25519 {

// ================================================================================================
// prep kernels
// ================================================================================================
// (compressed inputs: k_prep_compressed lives in kernels.hip -- a 252-squaring chain per lane at full occupancy wants
// the chained-carry field arithmetic of that translation unit; launch_prep_compressed)
// raw 160-byte points: Montgomery-trick normalisation, CH points per lane (cf. k_compress_p32)
// raw 160-byte points: Montgomery-trick normalisation, CH points per lane (cf. k_compress_p32).  The kernel moves 1.1 GB per
// 2^21 points (Z, then X, Y, Z again, 48-byte prefix products out and back, 128-byte records out): 0.30 ms against a
// memory floor of ~0.27 ms.  Tried in round 2 and dropped: one inversion per BLOCK through an LDS tree (-45 % field
// operations, but one wave inverts while three wait: 0.35 ms; 0.40 ms when LLVM moved the wave-uniform inversion to the
// scalar unit), CH = 32 / 8, fetching every record one step ahead (+12 VGPRs, 0.34 ms), and three launches (lane products,
// ONE batched inversion over the lane totals, unwind: -68 % field operations, 0.083 + 0.107 + 0.192 ms -- the two
// memory passes run at 3.6 - 5.2 TB/s and then contend with the sort on the second stream: no gain end to end), and
// wave-coalesced record I/O transposed through LDS (8x fewer cache-line requests per instruction, but 40 + 30 + 32 LDS
// dword accesses and four barriers per point: 0.47 ms).
// WAVE-COALESCED memory accesses.  Lane t owns points t, t + T, ...: the 64 lanes of a wave own 64 CONSECUTIVE points at
// every step; if each lane fetched its own 40-byte coordinates and stored its own 128-byte record (round 1's k_prep_raw),
// every memory instruction would look up 64 different cache lines -- 1664 look-ups per point and wave, on
// the texture/L1 path that the accumulation of the previous pass (one gather per addition) and the sort also live on.
// Here the wave DMAs the whole 10 KB block of its 64 points into LDS (global_load_lds_dwordx4, 8 lines per instruction),
// the prefix products live in a [step][piece][lane] layout (8 lines per instruction), and the records go out through an
// LDS transpose (piece c of record r at position (c + r) mod 8: conflict-free both ways) as eight fully coalesced
// stores: 272 look-ups per point and wave.  The block of step j+1 (j-1 on the way back) is in flight during step j.
template <int CH, int WPB>                                  // points per lane, waves per block
__global__ void __launch_bounds__(64 * WPB) k_prep_raw2(const uint8_t *__restrict__ in, u64 n, u32 *__restrict__ prefix, u32 *__restrict__ pts, u64 dst0) {
    C25519_PRIO_SIDE();
    __shared__ uint4 stage_in[WPB * 640];                    // per wave: 64 points x 160 bytes
    __shared__ uint4 stage_out[WPB * 512];                   // per wave: 64 records x 128 bytes
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u64 T = (u64)gridDim.x * (64 * WPB), t = (u64)blockIdx.x * (64 * WPB) + threadIdx.x, w0 = t - lane;
    if (w0 >= n) return;                                     // the whole wave is out of range
    uint4 *sin = stage_in + wv * 640, *sout = stage_out + wv * 512;
    const uint4 *in4 = reinterpret_cast<const uint4 *>(in);
    const u64 last4 = n * 10 - 1;
    int nj = 0;
    while (nj < CH && w0 + (u64)nj * T < n) nj++;            // steps with at least one point of this wave in range
#define C25519_PREP_ISSUE(j)                                                                                                   \
    {                                                                                                                          \
        const u64 b4 = (w0 + (u64)(j) * T) * 10;                                                                               \
        _Pragma("unroll") for (int i = 0; i < 10; i++) {                                                                      \
            u64 a = b4 + (u64)(i * 64) + lane;                                                                                 \
            a = a > last4 ? last4 : a;                                                                                         \
            __builtin_amdgcn_global_load_lds((gbl_void *)(in4 + a), (lds_void *)(sin + i * 64), 16, 0, 0);                     \
        }                                                                                                                      \
    }
    const uint4 *my4 = sin + lane * 10;
    const uint2 *my2 = reinterpret_cast<const uint2 *>(sin) + lane * 20;
    uint4 *pre4 = reinterpret_cast<uint4 *>(prefix) + (w0 / 64) * (u64)(CH * 3 * 64) + lane;
    feT acc = fe_one();
    bool affine = true;
    C25519_PREP_ISSUE(0)
#pragma unroll 1
    for (int j = 0; j < nj; j++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 z0 = my4[5], z1 = my4[6];
        const uint2 z2 = my2[14];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j + 1 < nj) C25519_PREP_ISSUE(j + 1)
        if (t + (u64)j * T < n) {
            const u64 l[5] = {z0.x | (u64)z0.y << 32, z0.z | (u64)z0.w << 32, z1.x | (u64)z1.y << 32, z1.z | (u64)z1.w << 32, z2.x | (u64)z2.y << 32};
            affine = affine && (l[0] == 1) && ((l[1] | l[2] | l[3] | l[4]) == 0);
            pre4[(j * 3 + 0) * 64] = make_uint4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]);
            pre4[(j * 3 + 1) * 64] = make_uint4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]);
            pre4[(j * 3 + 2) * 64] = make_uint4(acc.v[8], acc.v[9], 0u, 0u);
            acc = fe_mul(acc, fe_from_limbs51(l));
        }
    }
    feT inv = fe_one();
    if (!affine) inv = fe_invert(acc);
    const u32 sub = lane >> 3, coff = ((lane & 7u) - sub) & 7u;
    uint4 pa = make_uint4(0, 0, 0, 0), pb = pa, pc = pa;     // prefix product of the step about to be unwound
    if (nj > 0) { pa = pre4[((nj - 1) * 3 + 0) * 64]; pb = pre4[((nj - 1) * 3 + 1) * 64]; pc = pre4[((nj - 1) * 3 + 2) * 64]; }
    if (nj > 0) C25519_PREP_ISSUE(nj - 1)
#pragma unroll 1
    for (int j = nj - 1; j >= 0; j--) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 x0 = my4[0], x1 = my4[1], y1 = my4[3], y2 = my4[4], z0 = my4[5], z1 = my4[6];
        const uint2 x2 = my2[4], y0 = my2[5], z2 = my2[14];
        const uint4 a = pa, b = pb, c = pc;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j > 0) {
            pa = pre4[((j - 1) * 3 + 0) * 64]; pb = pre4[((j - 1) * 3 + 1) * 64]; pc = pre4[((j - 1) * 3 + 2) * 64];
            C25519_PREP_ISSUE(j - 1)
        }
        const u64 lx[5] = {x0.x | (u64)x0.y << 32, x0.z | (u64)x0.w << 32, x1.x | (u64)x1.y << 32, x1.z | (u64)x1.w << 32, x2.x | (u64)x2.y << 32};
        const u64 ly[5] = {y0.x | (u64)y0.y << 32, y1.x | (u64)y1.y << 32, y1.z | (u64)y1.w << 32, y2.x | (u64)y2.y << 32, y2.z | (u64)y2.w << 32};
        const u64 lz[5] = {z0.x | (u64)z0.y << 32, z0.z | (u64)z0.w << 32, z1.x | (u64)z1.y << 32, z1.z | (u64)z1.w << 32, z2.x | (u64)z2.y << 32};
        uint4 q[PTS_Q];
        for (int i = 0; i < PTS_Q; i++) q[i] = make_uint4(0, 0, 0, 0);
        if (t + (u64)j * T < n) {
            if (affine) pts_pieces(fe_from_limbs51(lx), fe_from_limbs51(ly), q);
            else {
                feT pre;
                pre.v[0] = a.x; pre.v[1] = a.y; pre.v[2] = a.z; pre.v[3] = a.w; pre.v[4] = b.x; pre.v[5] = b.y; pre.v[6] = b.z; pre.v[7] = b.w;
                pre.v[8] = c.x; pre.v[9] = c.y;
                const feT zi = fe_mul(inv, pre);
                inv = fe_mul(inv, fe_from_limbs51(lz));
                pts_pieces(fe_mul(fe_from_limbs51(lx), zi), fe_mul(fe_from_limbs51(ly), zi), q);
            }
        }
#pragma unroll
        for (int i = 0; i < PTS_Q; i++) sout[lane * 8 + ((i + lane) & 7u)] = q[i];
        uint4 *dst = reinterpret_cast<uint4 *>(pts) + PTS_Q * (dst0 + w0 + (u64)j * T);
#pragma unroll
        for (int i = 0; i < PTS_Q; i++) {
            const u32 r = 8u * i + sub;                       // this lane stores piece coff of record r
            const uint4 v = sout[i * 64 + lane];
            if (w0 + (u64)j * T + r < n) dst[r * 8 + coff] = v;
        }
    }
#undef C25519_PREP_ISSUE
}
__global__ void k_prep_basepoint(u32 *pts, u64 dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { ge_p3 B = ge_basepoint(); pts_store(pts, dst, B.X, B.Y); }
}

// ================================================================================================
// digits + counting sort
// ================================================================================================
// Window layout.  Scalars are reduced mod l (< 2^253, scalar.rs:193-205) in every MSM the reference performs, so the
// 253 bits are shared out EVENLY: equal windows of c bits would leave a 13-bit rump at the top (c = 16) whose few
// buckets collect 16x longer lists than the others.  From the top: an overflow window for bits 253..255 (empty unless
// a caller passes an unreduced scalar, which stays correct), one UNSIGNED window of c-1 bits (its digits 1..2^(c-1)
// fill all `half` buckets and it produces no carry), and below it signed windows of c or c-1 bits.
//   digit k = bits [pos[k], pos[k] + wid[k]) of s' = s + addk, minus 2^(wid[k]-1) for the signed windows,
//   addk = sum over signed windows of 2^(pos[k] + wid[k] - 1).
// (struct msm_geom: msm_internal.h)

// D[k][t] = window k of s' = s + addk  (u16); flags bit 255 of any scalar
__global__ void __launch_bounds__(256) k_digits(const uint8_t *__restrict__ scalars, u64 n, msm_geom g, uint16_t *__restrict__ D, u32 *__restrict__ bad_scalar) {
    C25519_PRIO_CHAIN();
    // (two terms per thread and one 32-bit store per window instead of two 2-byte stores: 0.24 ms against 0.11; the block's
    //  17 x 256 digits through LDS and out as 16-byte stores: 0.29 ms)
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u32 s[9];
    load8(scalars, t, s);
    if (s[7] >> 31) atomicOr(bad_scalar, 1u);
    u64 carry = 0;
    for (int i = 0; i < 8; i++) { u64 v = (u64)s[i] + g.addk[i] + carry; s[i] = (u32)v; carry = v >> 32; }
    s[8] = (u32)carry;
    for (int k = 0; k < g.nwin; k++) {
        int bit = g.pos[k], wi = bit >> 5, sh = bit & 31;
        u64 two = (u64)s[wi] | ((u64)(wi + 1 <= 8 ? s[wi + 1] : 0u) << 32);
        u32 v = (u32)(two >> sh) & ((1u << g.wid[k]) - 1u);
        D[(u64)k * n + t] = (uint16_t)v;
    }
}
// merged layout (precomputed static points): D[k * ns + t] = d + 2^(c-1), d = signed digit k of scalar t in [-2^(c-1), 2^(c-1))
// (windows 0 .. K-2 signed through s' = s + sum_k 2^(c k + c - 1), window K-1 unsigned); t >= n: digit 0
__global__ void __launch_bounds__(256) k_digits_merged(const uint8_t *__restrict__ scalars, u64 n, u64 ns, int c, int K, uint16_t *__restrict__ D, u32 *__restrict__ bad_scalar) {
    C25519_PRIO_CHAIN();
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ns) return;
    const u32 HALF = 1u << (c - 1);
    if (t >= n) { for (int k = 0; k < K; k++) D[(u64)k * ns + t] = (uint16_t)HALF; return; }
    u32 s[9];
    load8(scalars, t, s);
    s[8] = 0;
    u32 carry = 0;
    for (int k = 0; k < K; k++) {
        const int bit = c * k, wi = bit >> 5, sh = bit & 31;
        u32 v = 0;
        if (wi < 8) {
            u64 two = (u64)s[wi] | ((u64)s[wi + 1] << 32);
            v = (u32)(two >> sh) & ((1u << c) - 1u);
        }
        v += carry;
        const bool neg = (k != K - 1) && v >= HALF;              // d = v - 2^c in [-2^(c-1), 0), carry 1; else d = v in [0, 2^(c-1))
        carry = neg ? 1u : 0u;
        // stored value: d + HALF
        const u32 st = neg ? (v + HALF - (1u << c)) : (v + HALF);
        D[(u64)k * ns + t] = (uint16_t)st;
        if (k == K - 1 && v > HALF) atomicOr(bad_scalar, 1u);    // cannot happen for scalars below 2^256 (layout: c (K-1) + c - 1 >= 256)
    }
}
// table of the merged layout: lane i writes 2^(c k) P_i for k = 0 .. K-1 as raw 160-byte points [k][i] (normalised afterwards)
__global__ void __launch_bounds__(256) k_merged_table(const uint8_t *__restrict__ in_raw, u64 ns, int c, int K, uint8_t *__restrict__ out_raw) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    ge_p3 P = raw160_load(in_raw, i);
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        raw160_store(out_raw, (u64)k * ns + i, P);
        if (k + 1 < K) P = ge_mul_by_pow_2(P, c);
    }
}
// signed digit of window k from the stored value
// (a top digit above `half` can only come from a scalar with bit 255 set: k_digits has flagged it and the call fails;
// it is dropped here so that no kernel indexes past its tables)
__device__ __forceinline__ int digit_of(u32 v, int k, const msm_geom &g) {
    return (k >= g.first_unsigned) ? (v <= (u32)g.half ? (int)v : 0) : (int)v - (1 << (g.wid[k] - 1));     // msm_layout: the top two windows are unsigned
}

// histogram of bucket occupancy for (window k = blockIdx.y, chunk j = blockIdx.x)
__global__ void __launch_bounds__(1024) k_hist(const uint16_t *__restrict__ D, u64 n, msm_geom g, u64 chunk, u32 *__restrict__ counts) {
    C25519_PRIO_CHAIN();
    extern __shared__ u32 hist[];
    const int k = blockIdx.x, j = blockIdx.y, nchunk = gridDim.y;
    for (int b = threadIdx.x; b < g.half; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    u64 lo = (u64)j * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (u64 t = lo + threadIdx.x; t < hi; t += blockDim.x) {
        int d = digit_of(D[(u64)k * n + t], k, g);
        if (d != 0) atomicAdd(&hist[(d > 0 ? d : -d) - 1], 1u);
    }
    __syncthreads();
    u32 *out = counts + ((u64)k * nchunk + j) * g.half;
    for (int b = threadIdx.x; b < g.half; b += blockDim.x) out[b] = hist[b];
}
// counting-sort offsets in two steps.
// (1) one lane per (window, bucket): exclusive prefix over the chunks (in place) and the bucket total
__global__ void __launch_bounds__(256) k_scan_chunks(u32 *__restrict__ counts, int nchunk, msm_geom g, u32 *__restrict__ totals) {
    C25519_PRIO_CHAIN();
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (u64)g.nwin * g.half) return;
    int k = (int)(gid / g.half), b = (int)(gid % g.half);
    u32 run = 0;
    for (int j = 0; j < nchunk; j++) {
        u64 at = ((u64)k * nchunk + j) * g.half + b;
        u32 c = counts[at]; counts[at] = run; run += c;
    }
    totals[gid] = run;
}
// (2) one block per window: base[k][b] = exclusive scan of the bucket totals; base[k][half] = #entries
__global__ void __launch_bounds__(1024) k_scan_buckets(const u32 *__restrict__ totals, msm_geom g, u32 *__restrict__ base) {
    C25519_PRIO_CHAIN();
    __shared__ u32 part[1024];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int per = (g.half + 1023) / 1024;
    const int b0 = tid * per, b1 = b0 + per < g.half ? b0 + per : g.half;
    u32 sum = 0;
    for (int b = b0; b < b1; b++) sum += totals[(u64)k * g.half + b];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        u32 v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    u32 run = part[tid] - sum;
    for (int b = b0; b < b1; b++) { base[(u64)k * (g.half + 1) + b] = run; run += totals[(u64)k * g.half + b]; }
    if (tid == 1023) base[(u64)k * (g.half + 1) + g.half] = part[1023];
}
// scatter term indices (sign in bit 31) into bucket order
__global__ void __launch_bounds__(1024) k_scatter(const uint16_t *__restrict__ D, u64 n, msm_geom g, u64 chunk, const u32 *__restrict__ starts,
                                                  const u32 *__restrict__ base, u32 *__restrict__ sorted) {
    C25519_PRIO_CHAIN();
    extern __shared__ u32 cursor[];
    const int k = blockIdx.x, j = blockIdx.y, nchunk = gridDim.y;
    const u32 *st = starts + ((u64)k * nchunk + j) * g.half;
    const u32 *bs = base + (u64)k * (g.half + 1);
    for (int b = threadIdx.x; b < g.half; b += blockDim.x) cursor[b] = st[b] + bs[b];
    __syncthreads();
    u64 lo = (u64)j * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (u64 t = lo + threadIdx.x; t < hi; t += blockDim.x) {
        int d = digit_of(D[(u64)k * n + t], k, g);
        if (d != 0) {
            u32 pos = atomicAdd(&cursor[(d > 0 ? d : -d) - 1], 1u);
            sorted[(u64)k * n + pos] = (u32)t | (d < 0 ? 0x80000000u : 0u);
        }
    }
}

// ================================================================================================
// Two-pass partition sort (wide windows, c >= 13).  A direct scatter writes every 4-byte entry to its own cache line.
// Here pass 1 splits each chunk of a window into SLICES of 256 buckets through an LDS staging buffer, so that what
// goes to HBM are contiguous runs; pass 2 gives each (window, slice) bin -- ~16 K entries, all of it in LDS -- to one
// block that counting-sorts it by the low 8 bucket bits and writes the final list, the bucket totals and the bucket
// offsets, all coalesced.  Intermediate entry: bucket_low8 << 24 | sign << 23 | term index (n <= 2^23).
// ================================================================================================
constexpr int PART_BPS_MAX = 256;        // buckets per slice: 2^g.bps_log2 <= this, chosen so that a bin holds ~16 K entries
constexpr int PART_CAP = 17408;          // bin capacity of the LDS path of pass 2 (mean <= 16384, sigma 128; larger bins take the global path)
// terms per pass-1 block: the staging buffer (4 bytes per term) plus 18 counters per slice must leave room for two blocks
// per CU (2 x 80 KB of the 160 KB LDS)
static inline int part_chunk(int SL) { return SL <= 128 ? 16384 : 15360; }

__device__ __forceinline__ bool part_entry(u32 v, int k, const msm_geom &g, u32 t, u32 &slice, u32 &entry) {
    int d = digit_of(v, k, g);
    if (d == 0) return false;
    u32 b = (u32)((d > 0 ? d : -d) - 1);
    slice = b >> g.bps_log2;
    entry = ((b & ((1u << g.bps_log2) - 1u)) << 24) | (d < 0 ? (1u << 23) : 0u) | t;
    return true;
}
// cc[(k*SL + s)*nchunk + j] = number of non-zero digits of chunk j, window k, that fall into slice s
// (counting while the digits are still in k_digits' registers -- one 1024-thread block per chunk, all windows -- was tried:
//  0.25 ms against 0.11 + 0.09 for the two kernels: the LDS atomics of 17 windows serialise in 128 blocks)
__global__ void __launch_bounds__(256) k_part_hist(const uint16_t *__restrict__ D, u64 n, msm_geom g, int SL, int PART_CHUNK, u32 *__restrict__ cc) {
    C25519_PRIO_CHAIN();
    extern __shared__ u32 sm[];                               // [4][SL]
    const int k = blockIdx.x, j = blockIdx.y, nchunk = gridDim.y, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * SL; i += 256) sm[i] = 0;
    __syncthreads();
    const u64 lo = (u64)j * PART_CHUNK, hi = lo + PART_CHUNK < n ? lo + PART_CHUNK : n;
    const uint16_t *Dk = D + (u64)k * n;
    if ((((u64)k * n) & 7) == 0 && hi - lo == PART_CHUNK) {           // full, 16-byte aligned chunk: eight digits per load
        const uint4 *q = reinterpret_cast<const uint4 *>(Dk + lo);
        constexpr int LD = 8;                                          // PART_CHUNK / 8 / 256 <= 8 loads per thread, all in flight at once
        uint4 v[LD];
#pragma unroll
        for (int r = 0; r < LD; r++) { const int i = threadIdx.x + 256 * r; v[r] = i < PART_CHUNK / 8 ? q[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int r = 0; r < LD; r++) {
            if (threadIdx.x + 256 * r >= PART_CHUNK / 8) break;
            u32 x[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
            for (int h = 0; h < 8; h++) {
                u32 sl, e;
                if (part_entry((x[h >> 1] >> (16 * (h & 1))) & 0xffffu, k, g, 0u, sl, e)) atomicAdd(&sm[w * SL + sl], 1u);
            }
        }
    } else {
        for (u64 t = lo + threadIdx.x; t < hi; t += 256) {
            u32 sl, e;
            if (part_entry(Dk[t], k, g, 0u, sl, e)) atomicAdd(&sm[w * SL + sl], 1u);
        }
    }
    __syncthreads();
    for (int sidx = threadIdx.x; sidx < SL; sidx += 256)
        cc[((u64)k * SL + sidx) * nchunk + j] = sm[sidx] + sm[SL + sidx] + sm[2 * SL + sidx] + sm[3 * SL + sidx];
}
// one block per window: exclusive scan of cc in (slice, chunk) order, in place; bin_base[k][s] (SL+1 entries); base[k][half]
// Every global access is wave-coalesced (tiles of 8192 counters go through LDS, where each thread then owns 8 consecutive ones).
// Round 2's form gave each thread 16 - 32 consecutive counters straight from memory: every load instruction of a wave touched 64
// cache lines, and beside k_accumulate -- whose gathers keep the texture path busy -- the kernel took 520 - 620 us instead of
// its 37 us alone (profiles/r03_msm_2p24_timeline.txt), which made the sort the critical path of a multi-pass MSM.
// (Blocks of 256 threads: the single-block-per-window form below serves round 2's digit-matrix path of the precomputed tables.)
constexpr int SCAN_PER = 16, SCAN_TILE = 256 * SCAN_PER;
__global__ void __launch_bounds__(256) k_part_scan(u32 *__restrict__ cc, int SL, int nchunk, msm_geom g, u32 *__restrict__ bin_base, u32 *__restrict__ base) {
    C25519_PRIO_CHAIN();
    __shared__ u32 tile[SCAN_TILE + SCAN_TILE / 32];         // element a lives at a + a / 32: a thread's consecutive elements and a wave's 64 consecutive ones are both (almost) conflict-free
    __shared__ u32 wsum[4];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, M = SL * nchunk;
    u32 *v = cc + (u64)k * M;
    u32 carry = 0;
#pragma unroll 1
    for (int t0 = 0; t0 < M; t0 += SCAN_TILE) {
#pragma unroll
        for (int r = 0; r < SCAN_PER; r++) { const int a = r * 256 + tid, e = t0 + a; tile[a + (a >> 5)] = e < M ? v[e] : 0u; }
        __syncthreads();
        u32 x[SCAN_PER], sum = 0;
#pragma unroll
        for (int q = 0; q < SCAN_PER; q++) { const int a = tid * SCAN_PER + q; x[q] = tile[a + (a >> 5)]; sum += x[q]; }
        u32 inc = sum;
        for (int off = 1; off < 64; off <<= 1) { const u32 y = __shfl_up(inc, off, 64); if (lane >= off) inc += y; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        u32 wbase = 0, total = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { const u32 ws = wsum[i]; wbase += i < w ? ws : 0u; total += ws; }
        u32 run = carry + wbase + inc - sum;
#pragma unroll
        for (int q = 0; q < SCAN_PER; q++) { const int a = tid * SCAN_PER + q; tile[a + (a >> 5)] = run; run += x[q]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SCAN_PER; r++) {
            const int a = r * 256 + tid, e = t0 + a;
            if (e < M) {
                const u32 val = tile[a + (a >> 5)];
                v[e] = val;
                if (e % nchunk == 0) bin_base[(u64)k * (SL + 1) + e / nchunk] = val;
            }
        }
        carry += total;
        __syncthreads();
    }
    if (tid == 0) { bin_base[(u64)k * (SL + 1) + SL] = carry; base[(u64)k * (g.half + 1) + g.half] = carry; }
}
// pass 1: chunk j of window k -> runs per slice in P1[k][..]
__global__ void __launch_bounds__(1024, 8) k_part1(const uint16_t *__restrict__ D, u64 n, msm_geom g, int SL, int PART_CHUNK, const u32 *__restrict__ gofs, u32 *__restrict__ P1) {
    C25519_PRIO_CHAIN();
    extern __shared__ u32 sm[];
    u32 *cnt = sm;                         // [16][SL]: per-wave counts, then per-wave cursors
    u32 *ls = sm + 16 * SL;                // [SL + 1]: start of each slice in the staging buffer
    u32 *stot = ls + SL + 1;               // [SL]
    u32 *stage = stot + SL;                // [PART_CHUNK]
    const int k = blockIdx.x, j = blockIdx.y, nchunk = gridDim.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * SL; i += 1024) cnt[i] = 0;
    __syncthreads();
    const u64 lo = (u64)j * PART_CHUNK, hi = lo + PART_CHUNK < n ? lo + PART_CHUNK : n;
    // every thread decodes its (at most 16) digits ONCE and keeps slice / entry in registers for the second sweep
    // (r1 re-read and re-decoded the chunk: 0.21 -> 0.17 ms per 2^21 terms together with the wave scan below)
    constexpr int PER = 16;                // PART_CHUNK <= 16384 = 16 x 1024
    u32 ent[PER], slc[PER];
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const u64 t = lo + threadIdx.x + 1024u * r;
        slc[r] = 0xffffffffu;
        if (t < hi) {
            u32 sl, e;
            if (part_entry(D[(u64)k * n + t], k, g, (u32)t, sl, e)) { slc[r] = sl; ent[r] = e; atomicAdd(&cnt[w * SL + sl], 1u); }
        }
    }
    __syncthreads();
    for (int sidx = threadIdx.x; sidx < SL; sidx += 1024) {          // per slice: exclusive prefix over the 16 waves
        u32 run = 0;
        for (int ww = 0; ww < 16; ww++) { u32 c = cnt[ww * SL + sidx]; cnt[ww * SL + sidx] = run; run += c; }
        stot[sidx] = run;
    }
    __syncthreads();
    if (w == 0) {                                                      // exclusive scan of the slice totals by one wave (SL <= 512: 8 per lane)
        const int per = (SL + 63) >> 6;
        u32 c8[8], sum = 0;
        for (int q = 0; q < per; q++) { const int i = per * lane + q; c8[q] = i < SL ? stot[i] : 0u; sum += c8[q]; }
        u32 inc = sum;
        for (int off = 1; off < 64; off <<= 1) { u32 x = __shfl_up(inc, off, 64); if (lane >= off) inc += x; }
        u32 run = inc - sum;
        for (int q = 0; q < per; q++) { const int i = per * lane + q; if (i < SL) ls[i] = run; run += c8[q]; }
        if (lane == 63) ls[SL] = inc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * SL; i += 1024) cnt[i] += ls[i % SL];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; r++)
        if (slc[r] != 0xffffffffu) stage[atomicAdd(&cnt[w * SL + slc[r]], 1u)] = ent[r];
    __syncthreads();
    for (int sidx = w; sidx < SL; sidx += 16) {                       // each wave copies whole runs
        const u32 len = stot[sidx], src = ls[sidx];
        u32 *dst = P1 + (u64)k * n + gofs[((u64)k * SL + sidx) * nchunk + j];
        for (u32 i = lane; i < len; i += 64) dst[i] = stage[src + i];
    }
}
// ---- one sweep over the SCALARS instead of three over a digit matrix ---------------------------------------------------
// Rounds 1-2 wrote every window digit to a u16 matrix D[window][term] (k_digits: 64 MB of scalars in, 71 MB out per 2^21
// terms) and then read it twice, window-major (k_part_hist: slice counts per chunk; k_part1: the partition) -- 277 MB
// and three launches before the first entry reaches its slice; and k_digits indexed its scalar words with a runtime
// window position, i.e. through scratch (0.10 ms for 135 MB).  Here a block owns a CHUNK of terms for ALL windows: a lane
// keeps SWEEP_TPT scalars (s' = s + addk, eight words each) in registers and treats each as a shift register -- the window
// layout is contiguous (msm_layout: pos[k+1] = pos[k] + wid[k]), so window k is always the low wid[k] bits and the next
// window arrives by a funnel shift with a wave-uniform amount: no dynamic register index, no digit matrix.
//
// The partition is CHUNK-LOCAL (third form of round 3).  The first two forms gave every (window, slice, chunk) run its exact place
// in a global (window, slice)-major array, which needs all chunks' counts before any chunk can write: a counting kernel over the
// same scalars (k_sweep_count, 48 - 104 us), a scan of its 557 K counters (k_seg_scan), and a scatter kernel that fetched 128 run
// offsets per window and copied 128 runs of ~256 bytes out (k_sweep_scatter, 136 - 173 us, 13 spilled registers).  Now
//   k_sweep_local   per window: count per wave and slice, block-wide scan, stage the entries by slice in LDS -- and write the staging
//                   buffer out AS IT IS, one contiguous block per (window, chunk), with its 129 slice starts     (97 us per 2^21 terms)
//   k_bin_totals    entries per (window, slice) bin = its run lengths added over the chunks                      (6 us)
//   k_part2g        pass 2 GATHERS a bin's runs from the chunks' blocks (one 256-byte segment per chunk)        (134 us; 105 - 112 with
//                   contiguous bins)
//   k_order_place   1024-thread blocks: a quarter of the per-(block, length class) global atomics               (11 us; 26 with 256)
// 250 us instead of 331 (363 at the start of the round, 560 in round 2), four launches instead of five.
// Deterministic like the kernels they replace (offsets come from exact counts, not from atomics on a global cursor).
constexpr int SWEEP_TPT = 8, SWEEP_THREADS = 1024, SWEEP_WAVES = SWEEP_THREADS / 64, SWEEP_CHUNK = SWEEP_THREADS * SWEEP_TPT;
// (eight words per scalar: s' = s + addk < 2^256 whenever bit 255 of s is clear, and a scalar with bit 255 set fails the call
//  anyway (bad_scalar); a term beyond n is loaded as s = 0, whose digits are all zero: s' = addk puts 2^(wid-1) into every signed
//  window and 0 into the unsigned ones -- it is skipped like any zero digit)
struct sweep_regs { u32 s[SWEEP_TPT][8]; };
__device__ __forceinline__ void sweep_load(const uint8_t *__restrict__ scalars, u64 n, u64 lo, const msm_geom &g, sweep_regs &R, u32 *__restrict__ bad_scalar) {
#pragma unroll
    for (int r = 0; r < SWEEP_TPT; r++) {
        const u64 t = lo + (u64)r * SWEEP_THREADS + threadIdx.x;
        u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t < n) load8(scalars, t, w);
        if (bad_scalar && (w[7] >> 31)) atomicOr(bad_scalar, 1u);
        u32 carry = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { const u64 v = (u64)w[i] + g.addk[i] + carry; R.s[r][i] = (u32)v; carry = (u32)(v >> 32); }
    }
}
// the low `wd` bits of term r, then the next window moves down (wd < 32, wave-uniform)
__device__ __forceinline__ u32 sweep_take(sweep_regs &R, int r, int wd) {
    const u32 v = R.s[r][0] & ((1u << wd) - 1u);
#pragma unroll
    for (int i = 0; i < 7; i++) R.s[r][i] = __funnelshift_r(R.s[r][i], R.s[r][i + 1], (u32)wd);
    R.s[r][7] >>= wd;
    return v;
}
// k_sweep_local: the staging buffer of (window k, chunk j) -- the chunk's entries grouped by slice -- goes to
// P1[k * wstride + j * SWEEP_CHUNK ..] and the slice starts to lsg[(k * (SL + 1) + s) * nchunk + j] (s = SL: the number of entries).
// The (slice, wave) counters live in ONE flat array, slice-major: counter (s, w) at s * NW + (w ^ ((s >> 2) & (NW - 1))) -- the order of
// the waves inside a slice does not matter, and this swizzle spreads a wave's counters of 64 consecutive slices over all 64 banks.
// In that order the counters ARE the layout of the staging buffer, so the cursors are a plain block-wide exclusive scan by all
// sixteen waves (first form: wave 0 walked all waves' counters of every slice -- 32 dependent LDS accesses per slice while fifteen
// waves idled; a build with the scan disabled put it at 45 us of 150).  Four barriers per window; a software-pipelined form with
// three (window k-1 staged while window k is counted) measured the same: the kernel's time is two LDS atomics per entry and
// their latency, not barriers.  Per-wave counters because LDS atomics of many waves on one set of counters serialise (a counting
// kernel with shared counters: 297 us against 48).
// zero_words: the small counters of the kernels further down the chain (bucket-order histogram and cursors, long-bucket counters) --
// zeroed here by block 0 instead of a memset of their own: beside k_accumulate every extra launch of the chain waits 30 - 180 us for
// a dispatch slot.  bad_blk[j] = 1 if a scalar of chunk j has bit 255 set (k_bin_totals ORs them into one word, the bucket
// reduction ORs that into the result slot: the sort itself never touches the slot).
__global__ void __launch_bounds__(SWEEP_THREADS) k_sweep_local(const uint8_t *__restrict__ scalars, u64 n, msm_geom g, int SL, u32 *__restrict__ lsg, u32 *__restrict__ bad_blk,
                                                               u32 *__restrict__ P1, u64 wstride, u32 *__restrict__ zero_words, int nzero) {
    C25519_PRIO_CHAIN();
    extern __shared__ u32 sm[];
    constexpr int NW = SWEEP_WAVES;
    u32 *cnt = sm;                         // [SL * NW], slice-major with the bank swizzle (above)
    u32 *cur = sm + NW * SL;               // [SL * NW]
    u32 *ls = sm + 2 * NW * SL;            // [SL + 1]: start of each slice in the staging buffer, then the number of entries
    u32 *stage = ls + SL + 1;              // [SWEEP_CHUNK]
    __shared__ u32 wtot[NW];
    __shared__ u32 sbad;
    const int j = blockIdx.x, nchunk = gridDim.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < nzero; i += SWEEP_THREADS) zero_words[i] = 0;
    if (threadIdx.x == 0) sbad = 0;
    const int total = NW * SL, ept = total >= SWEEP_THREADS ? total / SWEEP_THREADS : 1;
    const int base = (int)threadIdx.x * ept;
    for (int i = threadIdx.x; i < total; i += SWEEP_THREADS) cnt[i] = 0;
    __syncthreads();
    const u64 lo = (u64)j * SWEEP_CHUNK;
    sweep_regs R;
    sweep_load(scalars, n, lo, g, R, &sbad);
#pragma unroll 1
    for (int k = 0; k < g.nwin; k++) {
        const int wd = g.wid[k];
        u32 ent[SWEEP_TPT], slc[SWEEP_TPT];
#pragma unroll
        for (int r = 0; r < SWEEP_TPT; r++) {
            const u32 v = sweep_take(R, r, wd);
            slc[r] = 0xffffffffu;
            u32 sl, e;
            if (part_entry(v, k, g, (u32)lo + (u32)r * SWEEP_THREADS + threadIdx.x, sl, e)) {
                slc[r] = sl * NW + ((u32)w ^ ((sl >> 2) & (NW - 1)));
                ent[r] = e;
                atomicAdd(&cnt[slc[r]], 1u);
            }
        }
        __syncthreads();                                                   // 1: the counts of this window are complete (and the previous window has left the staging buffer)
        u32 v4[4], tsum = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            v4[e] = 0;
            if (e < ept && base + e < total) { v4[e] = cnt[base + e]; cnt[base + e] = 0; }
            tsum += v4[e];
        }
        u32 inc = tsum;
        for (int off = 1; off < 64; off <<= 1) { const u32 x = __shfl_up(inc, off, 64); if (lane >= off) inc += x; }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();                                                   // 2: the wave totals of the scan
        u32 run = inc - tsum;
#pragma unroll
        for (int q = 0; q < NW; q++) run += q < w ? wtot[q] : 0u;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (e < ept && base + e < total) {
                const int idx = base + e;
                cur[idx] = run;
                if ((idx & (NW - 1)) == 0) ls[idx / NW] = run;
                if (idx == total - 1) ls[SL] = run + v4[e];
            }
            run += v4[e];
        }
        __syncthreads();                                                   // 3: cursors and slice starts
#pragma unroll
        for (int r = 0; r < SWEEP_TPT; r++)
            if (slc[r] != 0xffffffffu) stage[atomicAdd(&cur[slc[r]], 1u)] = ent[r];
        __syncthreads();                                                   // 4: the staging buffer holds the entries slice by slice
        const u32 tot = ls[SL];
        u32 *dst = P1 + (u64)k * wstride + (u64)j * SWEEP_CHUNK;
        for (u32 i = threadIdx.x; i < tot; i += SWEEP_THREADS) dst[i] = stage[i];
        if ((int)threadIdx.x <= SL) lsg[((u64)k * (SL + 1) + threadIdx.x) * nchunk + j] = ls[threadIdx.x];
    }
    if (threadIdx.x == 0) bad_blk[j] = sbad;
}
// entries of every (window, slice) bin: one wave per bin adds the run lengths over the chunks.  Block 0 also folds the chunks'
// bad-scalar flags into one word (bad_sticky: ORed over the passes of a call).
__global__ void __launch_bounds__(256) k_bin_totals(const u32 *__restrict__ lsg, int nchunk, int SL, int nbins, u32 *__restrict__ binm, const u32 *__restrict__ bad_blk,
                                                    u32 *__restrict__ bad_ws, u32 *__restrict__ bad_sticky) {
    C25519_PRIO_CHAIN();
    if (blockIdx.x == 0) {
        u32 any = 0;
        for (int i = threadIdx.x; i < nchunk; i += 256) any |= bad_blk[i];
        any = __syncthreads_or((int)any);
        if (threadIdx.x == 0) { *bad_ws = any ? 1u : 0u; if (any && bad_sticky) atomicOr(bad_sticky, 1u); }
    }
    const int bin = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (bin >= nbins) return;
    const int k = bin / SL, s = bin % SL;
    const u32 *row0 = lsg + ((u64)k * (SL + 1) + s) * nchunk, *row1 = row0 + nchunk;
    u32 sum = 0;
    for (int j = lane; j < nchunk; j += 64) sum += row1[j] - row0[j];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += (u32)__shfl_xor((int)sum, d, 64);
    if (lane == 0) binm[bin] = sum;
}

// a long bucket's list is cut into segments of LONG_SEG entries: one work item each (k_long_segments)
struct long_item { u32 gid, lo, hi, first; };
// what the bucket order needs from one bucket with c entries: its length class (a 256-bin block-local histogram) and, for a
// list beyond the cap, its long-bucket work items
__device__ __forceinline__ void order_note_bucket(u32 c, u64 G, const msm_geom &g, const u32 *__restrict__ base, u32 *h, u32 max_items, long_item *__restrict__ items,
                                                  u32 *__restrict__ counters, u32 *__restrict__ long_gids, u32 *__restrict__ long_first) {
    atomicAdd(&h[255u - (c > 255u ? 255u : c)], 1u);
    if (c > g.long_cap) {
        const int k = (int)(G / g.half), b = (int)(G % g.half);
        const u32 lo = base[(u64)k * (g.half + 1) + b], hi = lo + c;
        const u32 nseg = (c + LONG_SEG - 1) / LONG_SEG;
        const u32 first = atomicAdd(&counters[0], nseg);
        const u32 lb = atomicAdd(&counters[1], 1u);
        long_gids[lb] = (u32)G;
        long_first[lb] = first;
        // number of segments of this bucket is recomputed by the combiner from base[]
        for (u32 sg = 0; sg < nseg && first + sg < max_items; sg++) {
            long_item it; it.gid = (u32)G; it.lo = lo + sg * LONG_SEG; it.hi = (it.lo + LONG_SEG < hi) ? it.lo + LONG_SEG : hi; it.first = first;
            items[first + sg] = it;
        }
    }
}
// pass 2: bin (window k, slice s) -> final order, bucket totals and bucket offsets.  The bin's entries live in
// registers (18 per thread), LDS holds only the sorted copy: 74 KB per block, two blocks per CU.
constexpr int PART_R = PART_CAP / 1024;
__global__ void __launch_bounds__(1024) k_part2(const u32 *__restrict__ P1, u64 n, msm_geom g, int SL, const u32 *__restrict__ bin_base,
                                                u32 *__restrict__ totals, u32 *__restrict__ base, u32 *__restrict__ sorted,
                                                u32 *__restrict__ ord_hist, u32 max_items, long_item *__restrict__ items, u32 *__restrict__ counters,
                                                u32 *__restrict__ long_gids, u32 *__restrict__ long_first) {
    C25519_PRIO_CHAIN();
    extern __shared__ u32 sm[];
    u32 *cnt = sm, *cur = sm + PART_BPS_MAX, *oh = sm + 2 * PART_BPS_MAX, *out = sm + 3 * PART_BPS_MAX;
    const int PART_BPS = 1 << g.bps_log2;
    // (one block per bin: persistent blocks -- two per compute unit, each walking bins b, b + grid, ... -- were tried against the
    //  4.25 rounds of 512 blocks this grid runs as: 177 us instead of 105; the hardware overlaps a retiring block's copy-out with
    //  its successor's loads, a loop with barriers does not)
    const int k = blockIdx.x, sidx = blockIdx.y, tid = threadIdx.x;
    u32 b0, m;
    b0 = bin_base[(u64)k * (SL + 1) + sidx]; m = bin_base[(u64)k * (SL + 1) + sidx + 1] - b0;
    const u32 *src = P1 + (u64)k * n + b0;
    u32 *dst = sorted + (u64)k * n + b0;
    const bool fits = m <= (u32)PART_CAP;
    if (tid < PART_BPS) cnt[tid] = 0;
    if (tid < 256) oh[tid] = 0;
    __syncthreads();
    u32 e[PART_R];
    if (fits) {
#pragma unroll
        for (int r = 0; r < PART_R; r++) { const u32 i = tid + 1024u * r; e[r] = i < m ? src[i] : 0u; }
#pragma unroll
        for (int r = 0; r < PART_R; r++) if (tid + 1024u * r < m) atomicAdd(&cnt[e[r] >> 24], 1u);
    } else {
        for (u32 i = tid; i < m; i += 1024) atomicAdd(&cnt[src[i] >> 24], 1u);
    }
    __syncthreads();
    if (tid < 64) {                                                    // exclusive scan of the bucket counts by one wave (4, 2 or 1 per lane)
        const int per = PART_BPS >> 6;
        u32 c4[4] = {0, 0, 0, 0}, sum = 0;
        for (int q = 0; q < per; q++) { c4[q] = cnt[per * tid + q]; sum += c4[q]; }
        u32 inc = sum;
        for (int off = 1; off < 64; off <<= 1) { u32 x = __shfl_up(inc, off, 64); if (tid >= off) inc += x; }
        u32 run = inc - sum;
        for (int q = 0; q < per; q++) { cur[per * tid + q] = run; run += c4[q]; }
    }
    __syncthreads();
    if (tid < PART_BPS) {
        const u64 b = (u64)sidx * PART_BPS + tid;
        totals[(u64)k * g.half + b] = cnt[tid];
        base[(u64)k * (g.half + 1) + b] = b0 + cur[tid];
    }
    __syncthreads();
    // the bucket order's length histogram and the long-bucket work list, while the counts are here (was k_order_hist,
    // a launch of its own over the totals: 25 us in the gap between two accumulations)
    if (tid < PART_BPS) order_note_bucket(cnt[tid], (u64)k * g.half + (u64)sidx * PART_BPS + tid, g, base, oh, max_items, items, counters, long_gids, long_first);
    __syncthreads();
    if (tid < 256 && oh[tid]) atomicAdd(&ord_hist[tid], oh[tid]);
    if (fits) {
#pragma unroll
        for (int r = 0; r < PART_R; r++)
            if (tid + 1024u * r < m) out[atomicAdd(&cur[e[r] >> 24], 1u)] = (e[r] & 0x7fffffu) | ((e[r] & (1u << 23)) << 8);
        __syncthreads();
        for (u32 i = tid; i < m; i += 1024) dst[i] = out[i];
    } else {
        // oversize bin = heavily skewed digits (e.g. one bucket holding most of the window).  Entries go straight to
        // their final place; lanes of a wave that share the first lane's bucket take their slots with ONE atomic.
        for (u32 i0 = 0; i0 < m; i0 += 1024) {
            const u32 i = i0 + tid;
            const bool have = i < m;
            const u32 ev = have ? src[i] : 0u, bk = ev >> 24;
            const u32 lead_bk = __shfl(bk, __ffsll((long long)__ballot(have)) - 1, 64);
            const unsigned long long same = __ballot(have && bk == lead_bk);
            u32 pos = 0;
            if (have && bk == lead_bk) {
                const int leader = __ffsll((long long)same) - 1, lane = tid & 63;
                u32 first = 0;
                if (lane == leader) first = atomicAdd(&cur[bk], (u32)__popcll(same));
                first = __shfl(first, leader, 64);
                pos = first + (u32)__popcll(same & ((1ull << lane) - 1ull));
            } else if (have) {
                pos = atomicAdd(&cur[bk], 1u);
            }
            if (have) dst[pos] = (ev & 0x7fffffu) | ((ev & (1u << 23)) << 8);
        }
    }
}


// pass 2 of the chunk-local form: the bin's entries are gathered from the chunks' blocks.  Wave w takes chunks w, w + 16, ...; a run
// of a chunk is read in pieces of 64 entries (one 256-byte segment); the pieces of a wave are listed in LDS once and walked twice --
// to count, and (from L2 now) to place: 32 pieces in registers with static indices need more than the 64 VGPRs two 1024-thread
// blocks per compute unit leave a lane (42 spilled).  A bin with more than P2G_ITER pieces per wave or more than PART_CAP entries --
// heavily skewed digits -- walks its chunks without the list and places its entries straight into the sorted array.
constexpr int P2G_ITER = 48;
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_part2g(const u32 *__restrict__ P1, u64 n, u64 wstride, msm_geom g, int SL, int nchunk, const u32 *__restrict__ lsg, const u32 *__restrict__ binm,
         u32 *__restrict__ totals, u32 *__restrict__ base, u32 *__restrict__ sorted,
         u32 *__restrict__ ord_hist, u32 max_items, long_item *__restrict__ items, u32 *__restrict__ counters,
         u32 *__restrict__ long_gids, u32 *__restrict__ long_first) {
    C25519_PRIO_CHAIN();
    extern __shared__ u32 sm[];
    u32 *cnt = sm, *cur = sm + PART_BPS_MAX, *oh = sm + 2 * PART_BPS_MAX, *out = sm + 3 * PART_BPS_MAX, *wl = out + PART_CAP;      // wl[16][P2G_ITER]
    __shared__ u32 red[16];
    const int PART_BPS = 1 << g.bps_log2;
    const int k = blockIdx.x, sidx = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // Everything the block needs from memory before the gather is requested together: the bin totals (the bin's place in the window's
    // sorted list = the entries of the bins before it; SL <= 256 <= the block) and this wave's run starts -- the gather does not wait
    // for the prefix sum.
    const u32 *row0 = lsg + ((u64)k * (SL + 1) + sidx) * nchunk, *row1 = row0 + nchunk;
    const u32 *src = P1 + (u64)k * wstride;
    u32 part = tid < sidx ? binm[(u64)k * SL + tid] : 0u;
    const u32 m = binm[(u64)k * SL + sidx];
    if (tid < PART_BPS) cnt[tid] = 0;
    if (tid < 256) oh[tid] = 0;
    // this wave's pieces: (offset in the window's P1 region) << 7 | entries in the piece
    int nslots = 0;
    for (int j0 = 0; j0 < nchunk; j0 += 16 * 64) {
        const int j = j0 + w + 16 * lane;
        u32 st = 0, len = 0;
        if (j < nchunk) { st = row0[j]; len = row1[j] - st; }
        const u32 np = (len + 63u) >> 6;
        u32 inc = np;
        for (int off = 1; off < 64; off <<= 1) { const u32 x = __shfl_up(inc, off, 64); if (lane >= off) inc += x; }
        const u32 first = (u32)nslots + inc - np;
        for (u32 p = 0; p < np; p++)
            if (first + p < (u32)P2G_ITER) wl[w * P2G_ITER + first + p] = (((u32)j * (u32)SWEEP_CHUNK + st + 64u * p) << 7) | (len - 64u * p < 64u ? len - 64u * p : 64u);
        nslots += (int)__shfl(inc, 63, 64);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += (u32)__shfl_xor((int)part, d, 64);
    if (lane == 0) red[w] = part;
    const bool fits = !__syncthreads_or(nslots > P2G_ITER) && m <= (u32)PART_CAP;      // (the barrier: counters zeroed, wave sums of the prefix written)
    u32 b0 = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) b0 += red[q];
    if (sidx == SL - 1 && tid == 0) base[(u64)k * (g.half + 1) + g.half] = b0 + m;      // number of entries of the window
    u32 *dst = sorted + (u64)k * n + b0;
    if (fits) {
#pragma unroll 1
        for (int t0 = 0; t0 < nslots; t0 += 8) {                         // eight pieces in flight
            u32 ev[8];
            bool ok[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const u32 d = t0 + q < nslots ? wl[w * P2G_ITER + t0 + q] : 0u;
                ok[q] = (u32)lane < (d & 127u);
                ev[q] = ok[q] ? src[(d >> 7) + lane] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) if (ok[q]) atomicAdd(&cnt[ev[q] >> 24], 1u);
        }
    } else {
        for (int j = w; j < nchunk; j += 16) {
            const u32 st = row0[j], en = row1[j];
            for (u32 o = st + lane; o < en; o += 64) atomicAdd(&cnt[src[(u64)j * SWEEP_CHUNK + o] >> 24], 1u);
        }
    }
    __syncthreads();
    if (tid < 64) {                                                    // exclusive scan of the bucket counts by one wave (4, 2 or 1 per lane)
        const int per = PART_BPS >> 6;
        u32 c4[4] = {0, 0, 0, 0}, sum = 0;
        for (int q = 0; q < per; q++) { c4[q] = cnt[per * tid + q]; sum += c4[q]; }
        u32 inc = sum;
        for (int off = 1; off < 64; off <<= 1) { u32 x = __shfl_up(inc, off, 64); if (tid >= off) inc += x; }
        u32 run = inc - sum;
        for (int q = 0; q < per; q++) { cur[per * tid + q] = run; run += c4[q]; }
    }
    __syncthreads();
    if (tid < PART_BPS) {
        const u64 b = (u64)sidx * PART_BPS + tid;
        totals[(u64)k * g.half + b] = cnt[tid];
        base[(u64)k * (g.half + 1) + b] = b0 + cur[tid];
    }
    __syncthreads();
    if (tid < PART_BPS) order_note_bucket(cnt[tid], (u64)k * g.half + (u64)sidx * PART_BPS + tid, g, base, oh, max_items, items, counters, long_gids, long_first);
    __syncthreads();
    if (tid < 256 && oh[tid]) atomicAdd(&ord_hist[tid], oh[tid]);
    if (fits) {
#pragma unroll 1
        for (int t0 = 0; t0 < nslots; t0 += 8) {
            u32 ev[8];
            bool ok[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const u32 d = t0 + q < nslots ? wl[w * P2G_ITER + t0 + q] : 0u;
                ok[q] = (u32)lane < (d & 127u);
                ev[q] = ok[q] ? src[(d >> 7) + lane] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) if (ok[q]) out[atomicAdd(&cur[ev[q] >> 24], 1u)] = (ev[q] & 0x7fffffu) | ((ev[q] & (1u << 23)) << 8);
        }
        __syncthreads();
        for (u32 i = tid; i < m; i += 1024) dst[i] = out[i];
    } else {
        // oversize bin = heavily skewed digits (e.g. one bucket holding most of the window).  Entries go straight to their final place;
        // lanes of a wave that share the first lane's bucket take their slots with ONE atomic.
        for (int j = w; j < nchunk; j += 16) {
            const u32 st = row0[j], en = row1[j];
            for (u32 o0 = st; o0 < en; o0 += 64) {
                const u32 o = o0 + lane;
                const bool have = o < en;
                const u32 ev = have ? src[(u64)j * SWEEP_CHUNK + o] : 0u, bk = ev >> 24;
                const u32 lead_bk = __shfl(bk, __ffsll((long long)__ballot(have)) - 1, 64);
                const unsigned long long same = __ballot(have && bk == lead_bk);
                u32 pos = 0;
                if (have && bk == lead_bk) {
                    const int leader = __ffsll((long long)same) - 1;
                    u32 first = 0;
                    if (lane == leader) first = atomicAdd(&cur[bk], (u32)__popcll(same));
                    first = __shfl(first, leader, 64);
                    pos = first + (u32)__popcll(same & ((1ull << lane) - 1ull));
                } else if (have) {
                    pos = atomicAdd(&cur[bk], 1u);
                }
                if (have) dst[pos] = (ev & 0x7fffffu) | ((ev & (1u << 23)) << 8);
            }
        }
    }
}


// Scatter in bucket-range slices.  A window's sorted list is 4n bytes (8 MB at n = 2^21) and every 128-byte line of it
// collects its 32 entries from 32 different chunk blocks over the whole kernel: written in one sweep, the lines leave
// the 4 MB L2 of the XCD half-filled and every 4-byte store reaches HBM as its own 32-byte sector (measured WRITE_SIZE
// 1.1 GB for 134 MB of payload).  Here a block keeps its chunk's digits in LDS (2 bytes x 65536) and sweeps them
// `parts` times, each time scattering only the buckets of one slice: the 32 chunk blocks of a window run on the same
// XCD at the same time (blockIdx.x = window, linear workgroup id mod 8 = XCD) and move through the slices roughly
// together, so the region being written (4n/parts bytes) stays in that L2 until its lines are complete.
__global__ void __launch_bounds__(1024) k_scatter_sliced(const uint16_t *__restrict__ D, u64 n, msm_geom g, u64 chunk, int parts,
                                                         const u32 *__restrict__ starts, const u32 *__restrict__ base, u32 *__restrict__ sorted) {
    C25519_PRIO_CHAIN();
    extern __shared__ u32 sm[];
    const int k = blockIdx.x, j = blockIdx.y, nchunk = gridDim.y;
    const int per = g.half / parts;
    u32 *cursor = sm;
    uint16_t *dig = reinterpret_cast<uint16_t *>(sm + per);
    const u64 lo = (u64)j * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const u32 cnt = hi > lo ? (u32)(hi - lo) : 0u;
    for (u32 i = threadIdx.x; i < cnt; i += blockDim.x) dig[i] = D[(u64)k * n + lo + i];
    const u32 *st = starts + ((u64)k * nchunk + j) * g.half;
    const u32 *bs = base + (u64)k * (g.half + 1);
#pragma unroll 1
    for (int q = 0; q < parts; q++) {
        const int b0 = q * per;
        __syncthreads();
        for (int i = threadIdx.x; i < per; i += blockDim.x) cursor[i] = st[b0 + i] + bs[b0 + i];
        __syncthreads();
        for (u32 i = threadIdx.x; i < cnt; i += blockDim.x) {
            int d = digit_of(dig[i], k, g);
            int bk = (d > 0 ? d : -d) - 1 - b0;
            if (d != 0 && bk >= 0 && bk < per) {
                u32 pos = atomicAdd(&cursor[bk], 1u);
                sorted[(u64)k * n + pos] = (u32)(lo + i) | (d < 0 ? 0x80000000u : 0u);
            }
        }
    }
}

// ================================================================================================
// bucket accumulation: one lane per (window, bucket)   [pippenger.rs:122-136, as gather lists]
// ================================================================================================
// ---- bucket order: lanes of one wave should own lists of equal length ----------------------------------
// Counting sort of the (window, bucket) ids by list length (clamped to 255), longest first, so that a
// wave's 64 lanes finish together (Poisson-distributed lengths otherwise cost ~25 % idle lanes) and the
// long lists start first.  ord_hist: 256 global bins; perm: bucket ids in processing order.
// The same sweep over the bucket totals also emits the work list of the wave-cooperative long-bucket path (one item per
// segment of LONG_SEG entries of a bucket longer than LONG_CAP), so the list exists before accumulation starts
// (round 1 had a separate k_find_long on the second stream: a 16-VGPR scan that took 0.6 ms starved beside k_accumulate).
__global__ void __launch_bounds__(256) k_order_hist(const u32 *__restrict__ totals, const u32 *__restrict__ base, msm_geom g, u64 gid_off, u64 nb,
                                                    u32 *__restrict__ ord_hist, u32 max_items, long_item *__restrict__ items,
                                                    u32 *__restrict__ counters /* [0]=#items [1]=#long buckets */, u32 *__restrict__ long_gids,
                                                    u32 *__restrict__ long_first) {
    C25519_PRIO_CHAIN();
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < nb) order_note_bucket(totals[gid_off + gid], gid + gid_off, g, base, h, max_items, items, counters, long_gids, long_first);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&ord_hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_order_scan(u32 *__restrict__ ord_hist) {   // one block: exclusive scan of 256 bins
    C25519_PRIO_CHAIN();
    __shared__ u32 p[256];
    u32 v = ord_hist[threadIdx.x];
    p[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        u32 a = (int)threadIdx.x >= off ? p[threadIdx.x - off] : 0;
        __syncthreads();
        p[threadIdx.x] += a;
        __syncthreads();
    }
    ord_hist[threadIdx.x] = p[threadIdx.x] - v;
}
// the same with the scan inside: every block scans the 256-bin histogram itself (read-only) and takes its slots from a
// separate cursor array (zeroed by k_sweep_local) -- one launch less in the chain
template <int BS>                                            // 256 bins, BS >= 256 threads: a block's buckets per bin take their slots with ONE global atomic per bin
__global__ void __launch_bounds__(BS) k_order_place(const u32 *__restrict__ totals, u64 nb, const u32 *__restrict__ ord_hist, u32 *__restrict__ ord_cursor, u32 *__restrict__ perm) {
    C25519_PRIO_CHAIN();
    __shared__ u32 h[256], start[256], basep[256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 256) {
        const u32 mine = ord_hist[threadIdx.x];
        u32 inc = mine;
        for (int off = 1; off < 64; off <<= 1) { const u32 y = __shfl_up(inc, off, 64); if (lane >= off) inc += y; }
        h[threadIdx.x] = 0;
        if (lane == 63) basep[w] = inc;                      // wave totals (basep reused below)
        start[threadIdx.x] = inc - mine;                     // within the wave; the waves before are added after the barrier
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        u32 wb = 0;
        for (int i = 0; i < w; i++) wb += basep[i];
        start[threadIdx.x] += wb;
    }
    const u64 gid = (u64)blockIdx.x * BS + threadIdx.x;
    u32 bin = 0, local = 0;
    if (gid < nb) { const u32 c = totals[gid]; bin = 255u - (c > 255u ? 255u : c); local = atomicAdd(&h[bin], 1u); }
    __syncthreads();
    if (threadIdx.x < 256 && h[threadIdx.x]) basep[threadIdx.x] = start[threadIdx.x] + atomicAdd(&ord_cursor[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (gid < nb) perm[basep[bin] + local] = (u32)gid;
}
__global__ void __launch_bounds__(256) k_order_scatter(const u32 *__restrict__ totals, u64 nb, u32 gid_off, u32 *__restrict__ ord_cursor, u32 *__restrict__ perm) {
    C25519_PRIO_CHAIN();
    __shared__ u32 h[256], basep[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 bin = 0, local = 0;
    if (gid < nb) { u32 c = totals[gid]; bin = 255u - (c > 255u ? 255u : c); local = atomicAdd(&h[bin], 1u); }
    __syncthreads();
    if (h[threadIdx.x]) basep[threadIdx.x] = atomicAdd(&ord_cursor[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (gid < nb) perm[basep[bin] + local] = (u32)gid + gid_off;
}

// Buckets longer than LONG_CAP are left to the wave-cooperative path below, so that no lane ever walks a
// long list alone (skewed inputs: e.g. the +1 carry digit of every unsigned 128-bit z_i in verify_batch lands
// ~n/2 terms in ONE bucket; identical scalars do the same in every window).
// (k_accumulate lives in accum.hip, built once per carry form of fe_mul: launch_accumulate_c0 / _c1)
// ---- long buckets -------------------------------------------------------------------------------------
// work list (made by k_order_hist): one item per (long bucket, segment of LONG_SEG entries); item = {gid, lo, hi, slot}
// sum across the 64 lanes of a wave (complete additions; lane 0 ends with the total)
__device__ __forceinline__ ge_p3 wave_sum(ge_p3 acc) {
#pragma unroll 1
    for (int off = 32; off > 0; off >>= 1) {
        ge_p3 o;
        for (int i = 0; i < 10; i++) {
            o.X.v[i] = __shfl_down(acc.X.v[i], off, 64); o.Y.v[i] = __shfl_down(acc.Y.v[i], off, 64);
            o.Z.v[i] = __shfl_down(acc.Z.v[i], off, 64); o.T.v[i] = __shfl_down(acc.T.v[i], off, 64);
        }
        acc = ge_add(acc, o);
    }
    return acc;
}
// one wave per work item: every lane adds its strided share of the segment, then a shuffle tree
__global__ void __launch_bounds__(64) k_long_segments(const u32 *__restrict__ pts, const u32 *__restrict__ sorted, u64 n, msm_geom g,
                                                      const long_item *__restrict__ items, const u32 *__restrict__ counters, u32 max_items,
                                                      u32 *__restrict__ seg_sums) {
    C25519_PRIO_LONG();
    const u32 nitems = counters[0] < max_items ? counters[0] : max_items;
#pragma unroll 1
    for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
        long_item it = items[item];
        int k = (int)(it.gid / g.half);
        const u32 *list = sorted + (u64)k * n;
        ge_p3 acc = ge_identity();
#pragma unroll 1
        for (u32 i = it.lo + threadIdx.x; i < it.hi; i += 64) {
            u32 e = list[i];
            acc = ge_madd_signed_p3(acc, pts_load(pts, e & 0x7fffffffu), (e >> 31) != 0);
        }
        acc = wave_sum(acc);
        if (threadIdx.x == 0) p40_store(seg_sums, item, acc);
    }
}
// one wave per long bucket: sum its segment sums -> buckets[gid]
__global__ void __launch_bounds__(64) k_long_combine(const u32 *__restrict__ base, msm_geom g, const u32 *__restrict__ counters, u32 max_items,
                                                     const u32 *__restrict__ long_gids, const u32 *__restrict__ long_first,
                                                     const u32 *__restrict__ seg_sums, u32 *__restrict__ buckets, int cont) {
    C25519_PRIO_LONG();
#pragma unroll 1
    for (u32 lb = blockIdx.x; lb < counters[1]; lb += gridDim.x) {
        u32 gid = long_gids[lb], first = long_first[lb];
        int k = (int)(gid / g.half), b = (int)(gid % g.half);
        u32 cnt = base[(u64)k * (g.half + 1) + b + 1] - base[(u64)k * (g.half + 1) + b];
        u32 nseg = (cnt + LONG_SEG - 1) / LONG_SEG;
        ge_p3 acc = ge_identity();
        if (nseg == 1) {
            acc = p40_load(seg_sums, first);
        } else {
#pragma unroll 1
            for (u32 s = threadIdx.x; s < nseg && first + s < max_items; s += 64) acc = ge_add(acc, p40_load(seg_sums, first + s));
            acc = wave_sum(acc);
        }
        if (threadIdx.x == 0) p40_store(buckets, gid, cont ? ge_add(acc, p40_load(buckets, gid)) : acc);      // cont: on top of the earlier passes' sum
    }
}

// ================================================================================================
// bucket reduction: col_k = sum_b (b+1) * B_b   [pippenger.rs:146-151]
//
// Round 1 ran an 8-ary hierarchy of running sums: one serial level and four wave-starved "cooperative" levels with
// 3..12 doublings each -- five launches and a dependent chain of ~62 additions + 30 doublings (0.33 ms at c = 16).
// Now two launches.  Level A: one WAVE per segment of 512 buckets; every lane folds 8 consecutive buckets serially
// (S = sum B_j, W = sum j B_j: 13 additions, the work-efficient part), then the 64 lanes combine through shuffles:
//     T_l = sum_{i >= l} S_i (6-step suffix scan),  V_l = 8 * [l >= 1] T_l + W_l,  W_seg = sum_l V_l (6-step butterfly)
// because sum_l l*S_l = sum_{l >= 1} T_l.  Level B: one wave per window does the same over the <= 64 segment pairs with
// weight 512 and adds S once more (bucket b holds digit magnitude b+1).  Chain: 26 additions + 3 doublings, then
// 14 additions + 9 doublings.
// ================================================================================================
__device__ __forceinline__ ge_p3 p3_shfl_down64(const ge_p3 &a, int d, int lane) {
    ge_p3 o;
    for (int i = 0; i < 10; i++) {
        o.X.v[i] = __shfl_down(a.X.v[i], d, 64); o.Y.v[i] = __shfl_down(a.Y.v[i], d, 64);
        o.Z.v[i] = __shfl_down(a.Z.v[i], d, 64); o.T.v[i] = __shfl_down(a.T.v[i], d, 64);
    }
    const bool in = lane + d < 64;
    const ge_p3 id = ge_identity();
    for (int i = 0; i < 10; i++) {
        o.X.v[i] = in ? o.X.v[i] : id.X.v[i]; o.Y.v[i] = in ? o.Y.v[i] : id.Y.v[i];
        o.Z.v[i] = in ? o.Z.v[i] : id.Z.v[i]; o.T.v[i] = in ? o.T.v[i] : id.T.v[i];
    }
    return o;
}
__device__ __forceinline__ ge_p3 p3_shfl_xor64(const ge_p3 &a, int d) {
    ge_p3 o;
    for (int i = 0; i < 10; i++) {
        o.X.v[i] = __shfl_xor(a.X.v[i], d, 64); o.Y.v[i] = __shfl_xor(a.Y.v[i], d, 64);
        o.Z.v[i] = __shfl_xor(a.Z.v[i], d, 64); o.T.v[i] = __shfl_xor(a.T.v[i], d, 64);
    }
    return o;
}
// in: lane l holds (S_l, W_l).  out (every lane): S = sum_l S_l is in lane 0's S; W = sum_l W_l + 2^shift * sum_l l*S_l in every lane
__device__ __forceinline__ void wave_weighted_sum(ge_p3 &S, ge_p3 &W, int shift, int lane) {
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) S = ge_add(S, p3_shfl_down64(S, d, lane));      // S_l <- sum_{i >= l} S_i
    ge_p3 V = S;
    {
        const ge_p3 id = ge_identity();
        const bool keep = lane >= 1;
        for (int i = 0; i < 10; i++) {
            V.X.v[i] = keep ? V.X.v[i] : id.X.v[i]; V.Y.v[i] = keep ? V.Y.v[i] : id.Y.v[i];
            V.Z.v[i] = keep ? V.Z.v[i] : id.Z.v[i]; V.T.v[i] = keep ? V.T.v[i] : id.T.v[i];
        }
    }
    V = ge_mul_by_pow_2(V, shift);
    V = ge_add(V, W);
#pragma unroll 1
    for (int d = 32; d > 0; d >>= 1) V = ge_add(V, p3_shfl_xor64(V, d));
    W = V;
}
constexpr int RED_LB = 8, RED_SEG = 64 * RED_LB;      // buckets per lane / per wave of level A
// level A: block (one wave) = segment `seg` of window k.  direct: the window has a single segment, write col_k itself.
// bad_ws (may be null): the sort's "a scalar has bit 255 set" word, ORed into the slot's flag 0 (the sort does not touch the slot)
__global__ void __launch_bounds__(64) k_reduce_a(const u32 *__restrict__ buckets, int half, int nseg, u32 *__restrict__ SW, u32 *__restrict__ cols, int direct,
                                                 const u32 *__restrict__ bad_ws) {
    C25519_PRIO_SIDE();
    const int k = blockIdx.x / nseg, seg = blockIdx.x % nseg, lane = threadIdx.x;
    if (bad_ws && blockIdx.x == 0 && lane == 0 && *bad_ws) atomicOr(cols + MSM_MAX_WIN * 40, 1u);
    const int b0 = seg * RED_SEG + lane * RED_LB;
    const u32 *B = buckets + (u64)k * half * 40;
    const ge_p3 id = ge_identity();
    ge_p3 run = (b0 + RED_LB - 1 < half) ? p40_load(B, b0 + RED_LB - 1) : id;
    ge_p3 acc = run;
#pragma unroll 1
    for (int j = RED_LB - 2; j >= 1; j--) {
        run = ge_add(run, (b0 + j < half) ? p40_load(B, b0 + j) : id);
        acc = ge_add(acc, run);
    }
    run = ge_add(run, (b0 < half) ? p40_load(B, b0) : id);
    wave_weighted_sum(run, acc, 3, lane);                // run (lane 0) = S_seg, acc = W_seg = sum (b - seg base) B_b
    if (lane == 0) {
        if (direct) p40_store(cols, k, ge_add(acc, run));
        else { p40_store(SW, 2 * (u64)blockIdx.x, run); p40_store(SW, 2 * (u64)blockIdx.x + 1, acc); }
    }
}
// level B: one wave per window over its nseg <= 64 segment pairs
__global__ void __launch_bounds__(64) k_reduce_b(const u32 *__restrict__ SW, int nseg, u32 *__restrict__ cols) {
    C25519_PRIO_SIDE();
    const int k = blockIdx.x, lane = threadIdx.x;
    const ge_p3 id = ge_identity();
    ge_p3 S = lane < nseg ? p40_load(SW, 2 * ((u64)k * nseg + lane)) : id;
    ge_p3 W = lane < nseg ? p40_load(SW, 2 * ((u64)k * nseg + lane) + 1) : id;
    wave_weighted_sum(S, W, 9, lane);                    // 512 = RED_SEG buckets per segment
    if (lane == 0) p40_store(cols, k, ge_add(W, S));
}

// ================================================================================================
// result slots and partial-result RECORDS
//
// A pass leaves its window column sums and its counters in a slot (msm "result slots" below).  A slot doubles as the
// fixed-size RECORD that travels between ranks / contexts in the multi-GPU decomposition: the flags area
// also carries a header -- the number of terms the window layout was derived from (msm_layout is a function of it
// alone), the number of passes summed into the record and a magic word -- so that whoever holds the records of all ranks
// can add them column by column and do the Horner fold ONCE (c25519_fold_partial_records), without the rank's result
// ever having been on its host.
//   flags [0] a scalar has bit 255 set  [1] points that do not decode  [2] bad A  [3] bad R  [4] non-canonical s
//         [5] bad message offsets       [8] terms (low word)  [9] terms (high word)  [10] passes  [11] magic
// ================================================================================================
constexpr int REC_TERMS_LO = 8, REC_TERMS_HI = 9, REC_PASSES = 10, REC_MAGIC = 11;
constexpr u32 REC_MAGIC_VALUE = 0x52503235u;               // "52PR"
// zero a slot and write its header; pre (may be null): counters a caller computed beforehand (whole-batch hashing in the
// transcript z-mode: [0] non-canonical s, [1] bad message offsets), merged into flags [4] and [5]
__global__ void __launch_bounds__(256) k_slot_init(u32 *__restrict__ slot, u32 terms_lo, u32 terms_hi, u32 passes, const u32 *__restrict__ pre) {
    for (int i = threadIdx.x; i < C25519_SLOT_U32; i += 256) {
        u32 v = 0;
        const int f = i - MSM_MAX_WIN * 40;
        if (f == REC_TERMS_LO) v = terms_lo;
        else if (f == REC_TERMS_HI) v = terms_hi;
        else if (f == REC_PASSES) v = passes;
        else if (f == REC_MAGIC) v = REC_MAGIC_VALUE;
        else if (f == 4 && pre) v = pre[0];
        else if (f == 5 && pre) v = pre[1];
        slot[i] = v;
    }
}
// rec (+)= the column sums and counters of cnt slots that share one window layout (first: rec is overwritten)
__global__ void __launch_bounds__(128) k_record_sum(u32 *__restrict__ rec, const u32 *__restrict__ slots, int cnt, int nwin, int first) {
    C25519_PRIO_SIDE();
    const int t = threadIdx.x;
    if (t < nwin) {
        ge_p3 acc = first ? p40_load(slots, t) : p40_load(rec, t);
#pragma unroll 1
        for (int i = first ? 1 : 0; i < cnt; i++) acc = ge_add(acc, p40_load(slots + (size_t)i * C25519_SLOT_U32, t));
        p40_store(rec, t, acc);
    } else if (t >= 64 && t < 80) {
        const int f = t - 64, at = MSM_MAX_WIN * 40 + f;
        if (f < 8) {                                          // counters add up
            u32 v = first ? 0u : rec[at];
            for (int i = 0; i < cnt; i++) v += slots[(size_t)i * C25519_SLOT_U32 + at];
            rec[at] = v;
        } else if (f == REC_PASSES) {
            u32 v = first ? 0u : rec[at];
            for (int i = 0; i < cnt; i++) v += slots[(size_t)i * C25519_SLOT_U32 + at];
            rec[at] = v;
        } else if (first) rec[at] = slots[at];                // header words: one layout for every pass
    }
}

// ================================================================================================
// verify_batch kernels
// ================================================================================================
// hram_i = SHA-512(R_i || A_i || M_i) (batch.rs:179-191): 64-byte digest out; flags[0] += non-canonical s,
// flags[1] |= 1 if the message offsets are not monotone or run past msgs_len (that message is hashed as empty)
__global__ void __launch_bounds__(256) k_hram(const uint8_t *__restrict__ msgs, const u64 *__restrict__ msg_off, u64 msgs_len, const uint8_t *__restrict__ sigs,
                                              const uint8_t *__restrict__ pks, u64 n, uint8_t *__restrict__ hram, u32 *__restrict__ flags, uint8_t *__restrict__ hred = nullptr) {
    C25519_PRIO_CHAIN();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 r[8], a[8], s[8];
    load8(sigs, 2 * i, r);
    load8(sigs, 2 * i + 1, s);
    load8(pks, i, a);
    if (!sc28_words_canonical(s)) atomicAdd(&flags[0], 1u);     // signature.rs:89-94 check_scalar: s < l, a word-wise comparison
    sha512_stream st;
    st.init();
    for (int j = 0; j < 4; j++) st.w[j] = bswap64((u64)r[2 * j] | ((u64)r[2 * j + 1] << 32));       // R || A fills the
    for (int j = 0; j < 4; j++) st.w[4 + j] = bswap64((u64)a[2 * j] | ((u64)a[2 * j + 1] << 32));   // first 64 bytes
    st.fill = 64; st.total = 64;
    const u64 o0 = msg_off[i], o1 = msg_off[i + 1];
    const bool okoff = o0 <= o1 && o1 <= msgs_len;
    if (!okoff) atomicOr(&flags[1], 1u);
    const uint8_t *m = msgs + o0;
    const u64 len = okoff ? o1 - o0 : 0;
    st.put_bytes(m, len);
    st.finish();
    u32 w[16];
    sha512_digest_words(st.h, w);
    uint4 *q = reinterpret_cast<uint4 *>(hram) + 4 * i;
    for (int j = 0; j < 4; j++) q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
    if (hred) {                                              // h_i mod l, 32 bytes: what the device z-tree commits to (below)
        u32 o[8];
        sc28_to_words(sc28_from_wide(w), o);
        store8(hred, i, o);
    }
}
// the same reduction for hashes that were computed elsewhere
__global__ void __launch_bounds__(256) k_hram_mod_l(const uint8_t *__restrict__ hram, u64 n, uint8_t *__restrict__ hred) {
    C25519_PRIO_CHAIN();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 *hw = reinterpret_cast<const u32 *>(hram) + 16 * i;
    u32 w[16], o[8];
    for (int j = 0; j < 16; j++) w[j] = hw[j];
    sc28_to_words(sc28_from_wide(w), o);
    store8(hred, i, o);
}

// device z-mode (C25519_Z_DEVICE; NOT the reference's derivation -- see include/c25519_hip.h).  The z_i must depend on
// every input bit of the batch (a per-signature or per-subtree derivation allows a 2^64 meet-in-the-middle forgery), so
// they are derived from the root of a hash tree over what the reference's transcript absorbs (batch.rs:191-199) -- hram_i =
// H(R_i || A_i || M_i) and the 32-byte s_i of every signature -- with hram_i taken mod l (v4; 32 bytes instead of 64: the batch
// equation only ever sees h_i mod l, batch.rs:213-217, so that is the value to bind; two blocks per four signatures instead of three).
//   node = first 32 bytes of the SHA-512 chaining value after absorbing  TAG(level, inputs, n) || data , where TAG is one
//   128-byte block (domain separation and shape binding: level, number of inputs of the level, batch size) whose
//   compression is done once on the host (the per-level IVs below), and `data` has a fixed length per level, so no
//   length padding is needed: a Merkle-Damgard chain over fixed-length inputs is collision resistant if the compression
//   function is; 32-byte nodes give the 128-bit level of the z_i.
//   level 0: data = (hram_4j mod l) || s_4j || ... || (hram_4j+3 mod l) || s_4j+3 (absent = zero bytes): 2 blocks per 4 signatures,
//            one lane each (v2 chained 12 blocks over 16 signatures per lane: 1024 waves for 2^20 signatures, one per SIMD, 226
//            VGPRs -- a latency-bound kernel that did not fit beside the decompression; v3: 3 blocks with 64-byte hram_i).
//   level l: data = four children: ONE compression per node.  These levels are pure latency (one dependent SHA-512
//            compression is ~30 us for a single wave), so the last ones (<= 1024 nodes) run inside one block.
constexpr int ZTREE_MAX_LEVELS = 16;
struct ztree_ivs { u64 iv[ZTREE_MAX_LEVELS][8]; };
__global__ void __launch_bounds__(256) k_ztree_first(const uint8_t *__restrict__ hred, const uint8_t *__restrict__ sigs, u64 n, ztree_ivs ivs, uint8_t *__restrict__ out) {
    C25519_PRIO_CHAIN();
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 m_out = (n + 3) / 4;
    if (j >= m_out) return;
    u64 hs[8];
    for (int q = 0; q < 8; q++) hs[q] = ivs.iv[0][q];
    u64 rec[32];                                          // 4 records of 64 bytes = 2 blocks
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u64 c = 4 * j + r;
        const u64 *h = reinterpret_cast<const u64 *>(hred) + 4 * c, *sg = reinterpret_cast<const u64 *>(sigs) + 8 * c + 4;
#pragma unroll
        for (int q = 0; q < 4; q++) rec[8 * r + q] = c < n ? bswap64(h[q]) : 0ull;
#pragma unroll
        for (int q = 0; q < 4; q++) rec[8 * r + 4 + q] = c < n ? bswap64(sg[q]) : 0ull;
    }
#pragma unroll 1
    for (int blk = 0; blk < 2; blk++) {
        u64 w[16];
#pragma unroll
        for (int q = 0; q < 16; q++) w[q] = blk == 0 ? rec[q] : rec[16 + q];
        sha512_compress(hs, w);
    }
    u64 *o = reinterpret_cast<u64 *>(out) + 4 * j;
    for (int q = 0; q < 4; q++) o[q] = hs[q];
}
// one 4-ary level: out[j] = F_level(in[4j] || in[4j+1] || in[4j+2] || in[4j+3])[0..32]
__device__ __forceinline__ void ztree_node4(const u64 *in, u64 m_in, u64 j, const u64 iv[8], u64 *out4) {
    u64 hs[8], w[16];
    for (int q = 0; q < 8; q++) hs[q] = iv[q];
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
        const u64 c = 4 * j + ch;
#pragma unroll
        for (int q = 0; q < 4; q++) w[4 * ch + q] = c < m_in ? in[4 * c + q] : 0ull;
    }
    sha512_compress(hs, w);
    for (int q = 0; q < 4; q++) out4[q] = hs[q];
}
__global__ void __launch_bounds__(256) k_ztree(const uint8_t *__restrict__ in, u64 m_in, u32 level, ztree_ivs ivs, uint8_t *__restrict__ out) {
    C25519_PRIO_CHAIN();
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (m_in + 3) / 4) return;
    u64 r[4];
    ztree_node4(reinterpret_cast<const u64 *>(in), m_in, j, ivs.iv[level], r);
    u64 *o = reinterpret_cast<u64 *>(out) + 4 * j;
    for (int q = 0; q < 4; q++) o[q] = r[q];
}
// the last levels (m_in <= 1024 nodes) in ONE block: no launch gaps between levels that hold a handful of nodes
__global__ void __launch_bounds__(256) k_ztree_tail(const uint8_t *__restrict__ in, u64 m_in, u32 level, ztree_ivs ivs, uint8_t *__restrict__ root) {
    C25519_PRIO_CHAIN();
    __shared__ u64 buf0[1024 * 4], buf1[256 * 4];
    for (u64 i = threadIdx.x; i < m_in * 4; i += 256) buf0[i] = reinterpret_cast<const u64 *>(in)[i];
    __syncthreads();
    u64 *cur = buf0, *nxt = buf1;
    u64 m = m_in;
    while (m > 1) {
        const u64 mo = (m + 3) / 4;                                  // <= 256 = blockDim
        if (threadIdx.x < mo) {
            u64 r[4];
            ztree_node4(cur, m, threadIdx.x, ivs.iv[level], r);
            for (int q = 0; q < 4; q++) nxt[4 * threadIdx.x + q] = r[q];
        }
        __syncthreads();
        u64 *t = cur; cur = nxt; nxt = t;
        m = mo; level++;
    }
    if (threadIdx.x < 4) reinterpret_cast<u64 *>(root)[threadIdx.x] = cur[threadIdx.x];
}
// step 3: (z_4j .. z_4j+3) = the four 16-byte quarters of SHA-512(root || LE64(j)) (standard, padded); n4 = ceil(n/4)
// lanes, z16 has room for 4*n4 entries.  A quarter is read as SIGN-MAGNITUDE: bit 127 = sign, bits 0..126 = |z_i|, i.e.
// z_i is uniform on {-(2^127-1) .. 2^127-1} (2^128 - 1 values; a forged batch passes with probability <= 2^-127.99
// against the reference's 2^-128).  Why signed: the MSM recodes scalars into signed windows, and a magnitude below 2^127
// never carries out of its eighth 16-bit window, so the R_i terms stay out of windows 8..15; an unsigned 128-bit z_i
// (C25519_Z_TRANSCRIPT) puts the carry digit +1 of about half of all R_i into ONE bucket of window 8 (the long-bucket
// path takes it).  The sign is applied to the stored point (k_apply_sign), the MSM scalar is |z_i|.
__global__ void __launch_bounds__(256) k_zderive(const uint8_t *__restrict__ root, u64 n4, uint8_t *__restrict__ z16) {
    C25519_PRIO_CHAIN();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const u64 *h = reinterpret_cast<const u64 *>(root);
    u64 hs[8], w[16];   // 40-byte message: one block
    sha512_init(hs);
    for (int q = 0; q < 4; q++) w[q] = h[q];           // the root is kept as big-endian words of the chaining value
    w[4] = bswap64(i); w[5] = 0x8000000000000000ull;
    for (int q = 6; q < 15; q++) w[q] = 0;
    w[15] = 40 * 8;
    sha512_compress(hs, w);
    u64 *o = reinterpret_cast<u64 *>(z16) + 8 * i;
    for (int q = 0; q < 8; q++) o[q] = bswap64(hs[q]);
}
// R_i <- -R_i where z_i is negative (device z-mode): swap y+x / y-x, negate 2dxy of the stored affine Niels record
__global__ void __launch_bounds__(256) k_apply_sign(u32 *__restrict__ pts, u64 dst0, const uint8_t *__restrict__ z16, u64 n) {
    C25519_PRIO_CHAIN();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!(reinterpret_cast<const u32 *>(z16)[4 * i + 3] >> 31)) return;
    ge_aniels A = pts_load(pts, dst0 + i);
    feT t = fe_carry(fe_neg(A.xy2d));
    u32 w[32];
    for (int q = 0; q < 10; q++) { w[q] = A.ymx.v[q]; w[10 + q] = A.ypx.v[q]; w[20 + q] = t.v[q]; }
    w[30] = 0; w[31] = 0;
    uint4 *q4 = reinterpret_cast<uint4 *>(pts) + PTS_Q * (dst0 + i);
    for (int q = 0; q < PTS_Q; q++) q4[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
// scalars of the batch equation (batch.rs:213-233): msm_scalars[1+i] = |z_i|, [1+n+i] = z_i*h_i;
// per-block partial sums of z_i*s_i (mod l) to `partial` (ten 28-bit limbs each).  signed_z: z16 is sign-magnitude (device z-mode).
// Arithmetic: sc28.h -- radix 2^28, folding with l = 2^252 + c; per signature one 512-bit reduction (95 multiplier instructions)
// and two 5 x 10 limb products with their reductions (95 each), against ~1200 in the 5 x 52 Montgomery form of rounds 1-2.
__global__ void __launch_bounds__(256) k_batch_scalars(const uint8_t *__restrict__ hram, const uint8_t *__restrict__ sigs, const uint8_t *__restrict__ z16,
                                                       u64 n, int signed_z, uint8_t *__restrict__ msm_scalars, u32 *__restrict__ partial) {
    C25519_PRIO_CHAIN();
    __shared__ u32 red[256][10];
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    sc28 zs = sc28_zero();
    if (i < n) {
        const u32 *hw = reinterpret_cast<const u32 *>(hram) + 16 * i;
        u32 h16[16];
        for (int j = 0; j < 16; j++) h16[j] = hw[j];
        const u32 *zw = reinterpret_cast<const u32 *>(z16) + 4 * i;
        const bool neg = signed_z && (zw[3] >> 31);
        u32 zwords[8] = {zw[0], zw[1], zw[2], signed_z ? (zw[3] & 0x7fffffffu) : zw[3], 0, 0, 0, 0};
        u32 s[8], zl[5];
        load8(sigs, 2 * i + 1, s);
        sc28_limbs_from_words<4, 5>(zwords, zl);
        const sc28 h = sc28_from_wide(h16);
        zs = sc28_mul_5x10(zl, sc28_from_words(s).v);        // |z| s   (s < 2^256: a non-canonical s is reduced here and fails the batch through the flag)
        sc28 hz = sc28_mul_5x10(zl, h.v);                    // |z| h
        if (neg) { zs = sc28_neg(zs); hz = sc28_neg(hz); }   // z = -|z|
        u32 out[8];
        sc28_to_words(hz, out);
        store8(msm_scalars, 1 + n + i, out);
        store8(msm_scalars, 1 + i, zwords);
    }
    for (int j = 0; j < 10; j++) red[threadIdx.x][j] = zs.v[j];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sc28 a, b;
            for (int j = 0; j < 10; j++) { a.v[j] = red[threadIdx.x][j]; b.v[j] = red[threadIdx.x + off][j]; }
            a = sc28_add(a, b);
            for (int j = 0; j < 10; j++) red[threadIdx.x][j] = a.v[j];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) for (int j = 0; j < 10; j++) partial[(u64)blockIdx.x * 10 + j] = red[0][j];
}

// msm_scalars[0] = -(sum of the per-block partial sums) mod l: one block, strided sums then a tree
__global__ void __launch_bounds__(256) k_bsum_finish(const u32 *__restrict__ partial, u32 nblk, uint8_t *__restrict__ msm_scalars) {
    C25519_PRIO_CHAIN();
    __shared__ u32 red[256][10];
    sc28 acc = sc28_zero();
    for (u32 b = threadIdx.x; b < nblk; b += 256) {
        sc28 p;
        for (int j = 0; j < 10; j++) p.v[j] = partial[(u64)b * 10 + j];
        acc = sc28_add(acc, p);
    }
    for (int j = 0; j < 10; j++) red[threadIdx.x][j] = acc.v[j];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sc28 a, b;
            for (int j = 0; j < 10; j++) { a.v[j] = red[threadIdx.x][j]; b.v[j] = red[threadIdx.x + off][j]; }
            a = sc28_add(a, b);
            for (int j = 0; j < 10; j++) red[threadIdx.x][j] = a.v[j];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sc28 t;
        for (int j = 0; j < 10; j++) t.v[j] = red[0][j];
        u32 w[8];
        sc28_to_words(sc28_neg(t), w);
        store8(msm_scalars, 0, w);
    }
}

hipError_t launch_hram(const uint8_t *msgs, const uint64_t *msg_off, uint64_t msgs_len, const uint8_t *sigs, const uint8_t *pks, uint64_t n, uint8_t *hram, uint32_t *flags, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hram, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, msgs, msg_off, msgs_len, sigs, pks, n, hram, flags);
    return hipGetLastError();
}

}  // namespace c25519

// ================================================================================================
// host orchestration
// ================================================================================================
static inline unsigned div_up64(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }

static ge_p3 host_p40(const uint32_t *t) {
    ge_p3 p;
    for (int i = 0; i < 10; i++) { p.X.v[i] = t[i]; p.Y.v[i] = t[10 + i]; p.Z.v[i] = t[20 + i]; p.T.v[i] = t[30 + i]; }
    return p;
}
void host_raw160(const ge_p3 &p, uint8_t *out) {
    const feT *f[4] = {&p.X, &p.Y, &p.Z, &p.T};
    for (int c = 0; c < 4; c++) {
        u32 l[10]; fe_canonical_limbs(*f[c], l);
        for (int i = 0; i < 5; i++) { uint64_t v = (uint64_t)l[2 * i] | ((uint64_t)l[2 * i + 1] << 26); memcpy(out + 40 * c + 8 * i, &v, 8); }
    }
}
ge_p3 host_from_raw160(const uint8_t *in) {
    ge_p3 p; feT *f[4] = {&p.X, &p.Y, &p.Z, &p.T};
    for (int c = 0; c < 4; c++) {
        uint64_t l[5];
        memcpy(l, in + 40 * c, 40);
        *f[c] = fe_from_limbs51(l);                        // exact for every u64 limb (devio.h)
    }
    return p;
}
void host_encode(const ge_p3 &R, int out_fmt, uint8_t *out) {
    if (out_fmt == C25519_FMT_RAW160) { host_raw160(R, out); return; }
    u32 w[8];
    if (out_fmt == C25519_FMT_RISTRETTO) ris_compress(R, w);
    else { feT zi = fe_invert(R.Z); ge_affine_compress(fe_mul(R.X, zi), fe_mul(R.Y, zi), w); }
    memcpy(out, w, 32);
}

static int pick_window(uint64_t n) {
    int lg = 0; while ((1ull << (lg + 1)) <= n) lg++;
    int c = lg - 4;
    if (c < 5) c = 5;
    if (c > 16) c = 16;
    return c;
}

// sort / long-bucket parameters that depend on the number of terms per window (n) and buckets per window (g.half)
static void msm_sort_params(uint64_t n, msm_geom &g) {
    // slices of 2^bps_log2 buckets such that a (window, slice) bin holds at most ~16 K entries (PART_CAP with 12 % headroom)
    g.bps_log2 = 8;
    while (g.bps_log2 > 6 && (n << g.bps_log2) / (uint64_t)g.half > 16500) g.bps_log2--;
    if ((1 << g.bps_log2) > g.half) { g.bps_log2 = 0; while ((2 << g.bps_log2) <= g.half) g.bps_log2++; }
    const uint64_t mean = n / (uint64_t)g.half + 1;
    g.long_cap = (u32)std::max<uint64_t>(LONG_CAP_MIN, (mean * 5 + 1) / 2);
}
// window layout for n terms (see msm_geom): signed windows share 253 - (c-1) bits evenly, then the unsigned (c-1)-bit
// window, then bits 253..255
void msm_layout(uint64_t n, msm_geom &g) {
    g.c = pick_window(n);
    g.half = 1 << (g.c - 1);
    const int low_bits = 253 - (g.c - 1), nsig = (low_bits + g.c - 1) / g.c, wbase = low_bits / nsig, wrem = low_bits % nsig;
    uint32_t a[9] = {0};
    int bit = 0;
    for (int k = 0; k < nsig; k++) {
        g.pos[k] = (unsigned char)bit; g.wid[k] = (unsigned char)(wbase + (k < wrem ? 1 : 0));
        bit += g.wid[k];
        a[(bit - 1) >> 5] |= 1u << ((bit - 1) & 31);
    }
    g.pos[nsig] = (unsigned char)bit; g.wid[nsig] = (unsigned char)(g.c - 1);          // bit == 253 - (c-1)
    g.pos[nsig + 1] = 253; g.wid[nsig + 1] = 3;
    g.nwin = nsig + 2;
    g.first_unsigned = nsig;
    msm_sort_params(n, g);
    for (int k = g.nwin; k < MSM_MAX_WIN; k++) { g.pos[k] = 0; g.wid[k] = 1; }
    for (int i = 0; i < 8; i++) g.addk[i] = a[i];
}
// diagnostics (host only, no GPU needed): the layout the MSM would use for n terms
EXPORT int32_t c25519_msm_geometry(uint64_t n, int32_t *c, int32_t *nwin, uint8_t *pos, uint8_t *wid, uint32_t *addk) {
    msm_geom g;
    msm_layout(n, g);
    *c = g.c; *nwin = g.nwin;
    for (int k = 0; k < g.nwin; k++) { pos[k] = g.pos[k]; wid[k] = g.wid[k]; }
    for (int i = 0; i < 8; i++) addk[i] = g.addk[i];
    return C25519_OK;
}

// ---- result slots ------------------------------------------------------------------------------------------
// Nothing in a pass waits for the host any more.  A pass leaves its nwin column sums (col_k = sum_b (b+1) B_kb, 160 bytes
// each) and its flags / counters in a SLOT of a small device buffer; the caller enqueues all passes of a call, reads the
// slots back with ONE copy and one synchronisation, and does the O(windows) Horner fold (pippenger.rs:159) on the host
// through the same ge26.h formulas (a serial chain of ~250 doublings is a latency-bound tail that one CPU core
// finishes faster than one GPU lane).
//   slot flags: [0] a scalar has bit 255 set  [1] points that do not decode (prep)  [2] bad A  [3] bad R  [4] non-canonical s
//               [5] bad message offsets
static_assert(C25519_SLOT_U32 == MSM_MAX_WIN * 40 + 16, "slot layout");
static inline uint32_t *slot_flags(uint32_t *slot) { return slot + MSM_MAX_WIN * 40; }

// An MSM is enqueued in two halves so that a caller can put other work between them:
//   msm_enqueue_sort  digits, counting sort, bucket order and the long-bucket work list -- needs only the SCALARS; runs on
//                     sort_stream (nullptr = the context's main stream).  Memory-bound.
//   msm_enqueue_acc   joins sort_stream into the main stream, then accumulation (+ the long-bucket path on the second
//                     stream) and bucket reduction -- needs the POINTS (packed affine Niels records).  VALU-bound.
// The column sums go to d_slot.  ring (may be null): [0] / [1] bracket k_accumulate, [2] = end of the pass.
// wait_acc (may be null): an event the accumulation waits for -- the previous pass's accumulation on the other stream
// set: two accumulations side by side only share the multipliers, while a sort beside an accumulation is free.
struct msm_plan {
    msm_geom g; uint64_t n, nb; int nseg; uint32_t max_items, max_long;
    uint32_t *base, *sorted, *buckets, *perm, *SW, *counters, *lgids, *lfirst, *segs, *bad_ws, *bad_sticky = nullptr; long_item *items;
    hipStream_t sort_stream;
};
// md (may be null): merged layout -- d_scalars holds n_scalars scalars, the sort runs over md->K * md->ns digit-terms
// n_carve (0 = n): the number of terms the workspace is carved for -- passes that CONTINUE each other's bucket sums (msm_record_enqueue)
// must find the buckets at the same address although the last pass is shorter
int32_t msm_enqueue_sort(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n_scalars, const msm_geom &g, uint32_t *d_slot, hipStream_t sort_stream, msm_plan &pl,
                         const msm_merged *md = nullptr, uint64_t n_carve = 0) {
    const uint64_t n = md ? (uint64_t)md->K * md->ns : n_scalars;
    const uint64_t nc = n_carve > n ? n_carve : n;
    // every region BEFORE the buckets (oK) must be sized from nc, the number of terms the call's passes are carved for, never from
    // this pass's own n: a shorter last pass that CONTINUES its predecessor's bucket sums has to find them at the same offset
    // (round 3 derived nchunk from n: with C25519_MSM_PASS_LOG2 = 21 / 22 a last pass one chunk shorter moved oC .. oK)
    int nchunk = std::max(1, std::min(64, 512 / g.nwin));
    while (nchunk > 1 && nc / nchunk < 4096) nchunk /= 2;
    if ((nc + nchunk - 1) / nchunk > 65536) nchunk = (int)((nc + 65535) / 65536);   // a chunk's digits must fit LDS (k_scatter_sliced)
    uint64_t chunk = (n + nchunk - 1) / nchunk;
    const uint64_t nb = (uint64_t)g.nwin * g.half;
    const int nseg = (g.half + RED_SEG - 1) / RED_SEG;
    // workspace carve-up (tmp_d): D | counts | base | sorted | buckets | segment pairs | flags | perm | long-bucket lists
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t oD = carve((size_t)g.nwin * nc * 2), oC = carve((size_t)g.nwin * nchunk * g.half * 4), oB = carve((size_t)g.nwin * (g.half + 1) * 4);
    size_t oS = carve((size_t)g.nwin * nc * 4), oK = carve(nb * 160), oT = carve(nb * 4);
    size_t oSW = carve((size_t)g.nwin * nseg * 2 * 160), oF = carve(8192), oPerm = carve(nb * 4);
    // long-bucket path: at most (#entries / LONG_SEG + #long buckets) work items; a long bucket has > LONG_CAP entries
    const uint64_t entries = (uint64_t)g.nwin * nc;
    const uint32_t max_long = (uint32_t)std::min<uint64_t>(nb, entries / g.long_cap + 1);
    const uint32_t max_items = (uint32_t)(entries / LONG_SEG + max_long + 1);
    size_t oLI = carve((size_t)max_items * sizeof(long_item)), oLG = carve((size_t)max_long * 4), oLF = carve((size_t)max_long * 4);
    size_t oLS = carve((size_t)max_items * 160);
    // two-pass partition sort (see k_part1): pass-1 output, coarse counts / offsets, bin bases
    const bool use_part = g.c >= 13 && n <= (1ull << 23) && n >= (1ull << 16);
    // (the merged layout of the precomputed tables keeps round 2's digit-matrix kernels: its terms are (window, scalar) pairs)
    const bool sweep = use_part && !md && (g.half >> g.bps_log2) <= 256 && (g.half >> g.bps_log2) >= 8;      // (k_sweep_local: at most four counters per thread)
    const int SL = std::max(1, g.half >> g.bps_log2), PART_CHUNK = sweep ? SWEEP_CHUNK : part_chunk(SL), pchunks = (int)((n + PART_CHUNK - 1) / PART_CHUNK), pchunks_c = (int)((nc + PART_CHUNK - 1) / PART_CHUNK);
    size_t oP1 = 0, oCC = 0, oBB = 0;
    // (the chunk-local form of the sweep path: P1 holds whole chunk blocks, oCC the slice starts [window][SL + 1][chunk], oBB the bin totals and the chunks' flags)
    const size_t p1_words = sweep ? (size_t)g.nwin * pchunks_c * SWEEP_CHUNK : (size_t)g.nwin * nc;
    if (use_part) { oP1 = carve(p1_words * 4); oCC = carve((size_t)g.nwin * (SL + 1) * pchunks_c * 4); oBB = carve((size_t)g.nwin * (SL + 1) * 4 + (size_t)pchunks_c * 4); }
    int32_t r = ctx_reserve(ctx, ctx->tmp_d, off);
    if (r) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_d.p;
    uint16_t *D = (uint16_t *)(ws + oD);
    uint32_t *counts = (uint32_t *)(ws + oC), *base = (uint32_t *)(ws + oB), *sorted = (uint32_t *)(ws + oS), *buckets = (uint32_t *)(ws + oK);
    // small words of the chain (u32 index): [8..] long-bucket counters, [64..319] bucket-order histogram, [320..575] its cursors,
    // [576] "a scalar has bit 255 set" (ORed into the result slot by the bucket reduction: the sort itself never touches the slot)
    uint32_t *flags = (uint32_t *)(ws + oF), *totals = (uint32_t *)(ws + oT), *ord_hist = flags + 64, *ord_cursor = flags + 320, *bad_ws = flags + 576, *perm = (uint32_t *)(ws + oPerm);
    constexpr int ZERO_WORDS = 576;
    pl.g = g; pl.n = n; pl.nb = nb; pl.nseg = nseg; pl.max_items = max_items; pl.max_long = max_long;
    pl.base = base; pl.sorted = sorted; pl.buckets = buckets; pl.perm = perm; pl.SW = (uint32_t *)(ws + oSW); pl.counters = flags + 8; pl.bad_ws = bad_ws;
    pl.items = (long_item *)(ws + oLI); pl.lgids = (uint32_t *)(ws + oLG); pl.lfirst = (uint32_t *)(ws + oLF); pl.segs = (uint32_t *)(ws + oLS);
    pl.sort_stream = sort_stream;
    hipStream_t st = sort_stream ? sort_stream : ctx->stream;
    if (sweep) {
        uint32_t *P1 = (uint32_t *)(ws + oP1), *lsg = (uint32_t *)(ws + oCC), *binm = (uint32_t *)(ws + oBB), *bad_blk = binm + (size_t)g.nwin * (SL + 1);
        const uint64_t wstride = (uint64_t)pchunks_c * SWEEP_CHUNK;
        const size_t lds1 = ((size_t)2 * SWEEP_WAVES * SL + SL + 1 + SWEEP_CHUNK) * 4, lds2 = ((size_t)3 * PART_BPS_MAX + PART_CAP + 16 * P2G_ITER) * 4;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep_local), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part2g), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(k_sweep_local, dim3(pchunks), dim3(SWEEP_THREADS), lds1, st, d_scalars, n, g, SL, lsg, bad_blk, P1, wstride, flags, ZERO_WORDS);
        hipLaunchKernelGGL(k_bin_totals, dim3((g.nwin * SL + 3) / 4), dim3(256), 0, st, lsg, pchunks, SL, g.nwin * SL, binm, bad_blk, bad_ws, pl.bad_sticky);
        hipLaunchKernelGGL(k_part2g, dim3(g.nwin, SL), dim3(1024), lds2, st, P1, n, wstride, g, SL, pchunks, lsg, binm, totals, base, sorted, ord_hist, max_items, pl.items, pl.counters, pl.lgids, pl.lfirst);
        hipLaunchKernelGGL(k_order_place<1024>, dim3(div_up64(nb, 1024)), dim3(1024), 0, st, totals, nb, ord_hist, ord_cursor, perm);
        HIPCHK(hipGetLastError());
        return C25519_OK;
    }
    HIPCHK(hipMemsetAsync(flags, 0, 4096, st));
    if (md) hipLaunchKernelGGL(k_digits_merged, dim3(div_up64(md->ns, 256)), dim3(256), 0, st, d_scalars, n_scalars, md->ns, md->c, md->K, D, bad_ws);
    else hipLaunchKernelGGL(k_digits, dim3(div_up64(n, 256)), dim3(256), 0, st, d_scalars, n, g, D, pl.bad_sticky ? pl.bad_sticky : bad_ws);
    if (use_part) {
        uint32_t *P1 = (uint32_t *)(ws + oP1), *cc = (uint32_t *)(ws + oCC), *bin_base = (uint32_t *)(ws + oBB);
        const size_t lds1 = ((size_t)16 * SL + 2 * SL + 1 + PART_CHUNK) * 4, lds2 = ((size_t)3 * PART_BPS_MAX + PART_CAP) * 4;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(k_part_hist, dim3(g.nwin, pchunks), dim3(256), (size_t)4 * SL * 4, st, D, n, g, SL, PART_CHUNK, cc);
        hipLaunchKernelGGL(k_part_scan, dim3(g.nwin), dim3(256), 0, st, cc, SL, pchunks, g, bin_base, base);
        hipLaunchKernelGGL(k_part1, dim3(g.nwin, pchunks), dim3(1024), lds1, st, D, n, g, SL, PART_CHUNK, cc, P1);
        hipLaunchKernelGGL(k_part2, dim3(g.nwin, SL), dim3(1024), lds2, st, P1, n, g, SL, bin_base, totals, base, sorted, ord_hist, max_items, pl.items, pl.counters, pl.lgids, pl.lfirst);
    } else {
        size_t lds = (size_t)g.half * 4;
        // (window, chunk) grid order: blockIdx.x = window, so that the chunk blocks of one window share an XCD's L2
        if (lds > 48 * 1024) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        hipLaunchKernelGGL(k_hist, dim3(g.nwin, nchunk), dim3(1024), lds, st, D, n, g, chunk, counts);
        hipLaunchKernelGGL(k_scan_chunks, dim3(div_up64(nb, 256)), dim3(256), 0, st, counts, nchunk, g, totals);
        hipLaunchKernelGGL(k_scan_buckets, dim3(g.nwin), dim3(1024), 0, st, totals, g, base);
        constexpr int sparts = 8;                           // bucket-range slices of the scatter (k_scatter_sliced)
        const size_t lds_sliced = (size_t)g.half / sparts * 4 + (size_t)chunk * 2;
        if (g.half >= 1024 * sparts && lds_sliced <= 160 * 1024) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_sliced), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sliced));
            hipLaunchKernelGGL(k_scatter_sliced, dim3(g.nwin, nchunk), dim3(1024), lds_sliced, st, D, n, g, chunk, sparts, counts, base, sorted);
        } else hipLaunchKernelGGL(k_scatter, dim3(g.nwin, nchunk), dim3(1024), lds, st, D, n, g, chunk, counts, base, sorted);
    }
    // bucket order (longest lists first) and the long-bucket work list: still on the sort stream -- they only need the lists
    if (!use_part) hipLaunchKernelGGL(k_order_hist, dim3(div_up64(nb, 256)), dim3(256), 0, st, totals, base, g, (uint64_t)0, nb, ord_hist, max_items, pl.items, pl.counters, pl.lgids, pl.lfirst);
    hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(256), 0, st, ord_hist);
    hipLaunchKernelGGL(k_order_scatter, dim3(div_up64(nb, 256)), dim3(256), 0, st, totals, nb, 0u, ord_hist, perm);
    HIPCHK(hipGetLastError());
    return C25519_OK;
}
// cont: the accumulation starts from the bucket sums the previous pass on this workspace left (no reduction happened in between);
// reduce: the buckets are reduced into d_slot now (the last pass of a stream set; always, for the single-pass callers)
int32_t msm_enqueue_acc(c25519_ctx *ctx, const msm_plan &pl, const uint32_t *d_pts, uint32_t *d_slot, hipEvent_t *ring, hipEvent_t wait_acc, bool cont = false, bool reduce = true,
                        const uint32_t *d_bad_sticky = nullptr) {
    const msm_geom &g = pl.g;
    if (pl.sort_stream && pl.sort_stream != ctx->stream) {                // join: the main stream continues once the lists exist
        HIPCHK(hipEventRecord(ctx->ev_sort, pl.sort_stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_sort, 0));
    }
    hipStream_t st = ctx->stream;
    if (wait_acc) HIPCHK(hipStreamWaitEvent(st, wait_acc, 0));
    // long buckets are independent of k_accumulate (which skips them): fold them on the second stream meanwhile
    HIPCHK(hipEventRecord(ctx->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    hipLaunchKernelGGL(k_long_segments, dim3(std::min<uint32_t>(pl.max_items, 2048u)), dim3(64), 0, ctx->aux, d_pts, pl.sorted, pl.n, g, pl.items, pl.counters, pl.max_items, pl.segs);
    hipLaunchKernelGGL(k_long_combine, dim3(std::min<uint32_t>(pl.max_long, 1024u)), dim3(64), 0, ctx->aux, pl.base, g, pl.counters, pl.max_items, pl.lgids, pl.lfirst, pl.segs, pl.buckets, cont ? 1 : 0);
    if (ring) HIPCHK(hipEventRecord(ring[0], st));
    // (measured in round 2 and dropped: a CU-masked stream for this kernel -- 1/8 or 1/4 of the CUs kept free for the sort of
    //  the next pass -- 18.2 - 20.3 ms per 2^24 terms against 16.5; 512-thread blocks, i.e. two waves per SIMD with 176
    //  registers: 16.9 - 17.0 against 16.6 - 16.9; an LDS reservation to the same effect: 16.2 against 15.9; four waves per SIMD
    //  without a prefetched record: 16.0 / 15.5 against 15.1 - 15.3; un-serialised accumulations of neighbouring passes: +3 - 9 %)
    ctx->kname[0] = launch_accumulate(d_pts, pl.sorted, pl.base, pl.perm, pl.nb, pl.n, g, pl.buckets, cont ? 1 : 0, st);
    HIPCHK(hipEventRecord(ctx->ev_acc, st));
    if (ring) HIPCHK(hipEventRecord(ring[1], st));
    // The bucket reduction runs on the SECOND (high-priority) stream, behind the long-bucket kernels it depends on anyway.  On the
    // main stream its few small blocks had normal priority: once the sort of the next pass stopped being late (round 3) the next
    // accumulation -- on the other stream set -- began before they were dispatched, refilled every hole a retiring block left, and
    // k_reduce_b waited 1.1 ms for its 17 wave slots, holding back this stream set's next pass (profiles/r03_msm_2p24_timeline.txt).
    HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_acc, 0));
    if (reduce) {
        hipLaunchKernelGGL(k_reduce_a, dim3((unsigned)(g.nwin * pl.nseg)), dim3(64), 0, ctx->aux, pl.buckets, g.half, pl.nseg, pl.SW, d_slot, pl.nseg == 1 ? 1 : 0, d_bad_sticky ? d_bad_sticky : pl.bad_ws);
        if (pl.nseg > 1) hipLaunchKernelGGL(k_reduce_b, dim3((unsigned)g.nwin), dim3(64), 0, ctx->aux, pl.SW, pl.nseg, d_slot);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->ev_join, ctx->aux));
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));     // what follows on the main stream (the next pass of this stream set, the read-back) comes after the long buckets / the reduction
    if (ring) HIPCHK(hipEventRecord(ring[2], st));
    return C25519_OK;
}
int32_t msm_enqueue(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, const msm_geom &g, uint32_t *d_slot, hipEvent_t *ring,
                    hipStream_t sort_stream, hipEvent_t wait_acc = nullptr) {
    msm_plan pl;
    int32_t r = msm_enqueue_sort(ctx, d_scalars, n, g, d_slot, sort_stream, pl);
    if (r) return r;
    return msm_enqueue_acc(ctx, pl, d_pts, d_slot, ring, wait_acc);
}

// total = sum_k 2^pos_k col_k by Horner (pippenger.rs:159), host arithmetic over <= 56 points
static ge_p3 msm_horner(const uint32_t *cols, const msm_geom &g) {
    ge_p3 total = ge_identity();
    for (int k = g.nwin - 1; k >= 0; k--) {
        if (k != g.nwin - 1) total = ge_mul_by_pow_2(total, g.pos[k + 1] - g.pos[k]);
        total = ge_add(total, host_p40(&cols[(size_t)k * 40]));
    }
    return total;
}
// read slots [0, count) back (one copy, one synchronisation of the context's main stream)
static int32_t slots_collect(c25519_ctx *ctx, int count) {
    HIPCHK(hipMemcpyAsync(ctx->h_msm, ctx->d_slots, (size_t)count * C25519_SLOT_U32 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return C25519_OK;
}
static inline uint32_t *dslot(c25519_ctx *ctx, int i) { return ctx->d_slots + (size_t)i * C25519_SLOT_U32; }
static inline const uint32_t *hslot(c25519_ctx *ctx, int i) { return (const uint32_t *)ctx->h_msm + (size_t)i * C25519_SLOT_U32; }
// the context's own record (slot C25519_MAX_SLOTS of d_slots / h_msm): where a call that answers on the host sums its passes
static inline uint32_t *drec(c25519_ctx *ctx) { return dslot(ctx, C25519_MAX_SLOTS); }
static int32_t rec_collect(c25519_ctx *ctx) {
    HIPCHK(hipMemcpyAsync((uint32_t *)ctx->h_msm + (size_t)C25519_MAX_SLOTS * C25519_SLOT_U32, drec(ctx), (size_t)C25519_SLOT_U32 * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return C25519_OK;
}
static inline void slot_init(uint32_t *d_slot, uint64_t terms, const uint32_t *d_pre, hipStream_t st) {
    hipLaunchKernelGGL(k_slot_init, dim3(1), dim3(256), 0, st, d_slot, (uint32_t)terms, (uint32_t)(terms >> 32), terms ? 1u : 0u, d_pre);
}
static_assert(C25519_PARTIAL_RECORD_BYTES == C25519_SLOT_U32 * 4, "record = slot");

// Fold `count` records (HOST memory) into one point and one set of counters.  Records made with the same number of
// terms share their window layout: their columns are added window by window and the Horner fold (pippenger.rs:159) runs
// once; otherwise every record is folded on its own.  Pure host arithmetic over O(count x windows) points.
static int32_t records_fold(const uint8_t *records, uint64_t count, ge_p3 &R, uint32_t flags[8], std::string *err) {
    R = ge_identity();
    for (int j = 0; j < 8; j++) flags[j] = 0;
    bool same = true;
    uint64_t terms0 = 0;
    for (uint64_t i = 0; i < count; i++) {
        uint32_t f[16];
        memcpy(f, records + i * C25519_PARTIAL_RECORD_BYTES + (size_t)MSM_MAX_WIN * 160, sizeof f);
        if (f[REC_MAGIC] != REC_MAGIC_VALUE) { if (err) *err = "fold: not a partial-result record (bad magic)"; return -(int32_t)hipErrorInvalidValue; }
        for (int j = 0; j < 8; j++) flags[j] += f[j];
        const uint64_t terms = (uint64_t)f[REC_TERMS_LO] | ((uint64_t)f[REC_TERMS_HI] << 32);
        if (i == 0) terms0 = terms; else if (terms != terms0) same = false;
    }
    if (count == 0) return C25519_OK;
    std::vector<uint32_t> cols((size_t)MSM_MAX_WIN * 40);
    if (same) {
        if (terms0 == 0) return C25519_OK;                  // empty shards only
        msm_geom g;
        msm_layout(terms0, g);
        memcpy(cols.data(), records, (size_t)g.nwin * 160);
        for (uint64_t i = 1; i < count; i++) {
            const uint32_t *c = (const uint32_t *)(records + i * C25519_PARTIAL_RECORD_BYTES);
            for (int k = 0; k < g.nwin; k++) {
                const ge_p3 sum = ge_add(host_p40(&cols[(size_t)k * 40]), host_p40(c + (size_t)k * 40));
                for (int q = 0; q < 10; q++) { cols[(size_t)k * 40 + q] = sum.X.v[q]; cols[(size_t)k * 40 + 10 + q] = sum.Y.v[q]; cols[(size_t)k * 40 + 20 + q] = sum.Z.v[q]; cols[(size_t)k * 40 + 30 + q] = sum.T.v[q]; }
            }
        }
        R = msm_horner(cols.data(), g);
        return C25519_OK;
    }
    for (uint64_t i = 0; i < count; i++) {
        const uint32_t *c = (const uint32_t *)(records + i * C25519_PARTIAL_RECORD_BYTES);
        const uint32_t *f = c + (size_t)MSM_MAX_WIN * 40;
        const uint64_t terms = (uint64_t)f[REC_TERMS_LO] | ((uint64_t)f[REC_TERMS_HI] << 32);
        if (terms == 0) continue;
        msm_geom g;
        msm_layout(terms, g);
        memcpy(cols.data(), c, (size_t)g.nwin * 160);        // (records are only 4-byte aligned in general)
        R = ge_add(R, msm_horner(cols.data(), g));
    }
    return C25519_OK;
}

// one-shot MSM over prepared points (extra.hip: precomputed tables): enqueue, collect, fold
int32_t msm_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, ge_p3 &R) {
    msm_geom g;
    msm_layout(n, g);
    HIPCHK(hipMemsetAsync(dslot(ctx, 0), 0, C25519_SLOT_U32 * 4, ctx->stream));
    int32_t r = msm_enqueue(ctx, d_scalars, n, d_pts, g, dslot(ctx, 0), nullptr, nullptr);
    if (r) return r;
    if ((r = slots_collect(ctx, 1))) return r;
    if (slot_flags((uint32_t *)hslot(ctx, 0))[0]) { ctx->err = "msm: a scalar has bit 255 set (Scalar invariant #1 violated)"; return -(int32_t)hipErrorInvalidValue; }
    R = msm_horner(hslot(ctx, 0), g);
    return C25519_OK;
}

// ---- precomputed static points (VartimePrecomputedStraus, precomputed_straus.rs:57-127) --------------------------------
// The reference keeps, per static point, a table of odd multiples for width-8 NAF and still shares ONE doubling chain
// across all points.  The bucket method's analogue of "precompute so that the doublings disappear": keep 2^(c k) P_i for
// every window k (msm_merged).  Then every digit of every scalar is a term of ONE bucket problem -- a single
// accumulation over K * ns gather lists, a single bucket reduction, no Horner fold, no per-call point preparation.
void msm_merged_layout(uint64_t ns, msm_merged &m) {
    // window width from the number of digit-terms (as pick_window does for plain terms): c = clamp(log2(17 ns) - 4, 5, 16)
    m.ns = ns;
    m.c = pick_window(std::max<uint64_t>(1, ns) * 17);
    m.K = (257 + m.c - 1) / m.c;                          // c (K - 1) + (c - 1) >= 256: the unsigned top window cannot overflow its buckets
    while (m.c * (m.K - 1) + m.c - 1 < 256) m.K++;
}
int32_t msm_merged_build(c25519_ctx *ctx, const uint8_t *d_points, uint64_t ns, int in_fmt, const msm_merged &m, uint32_t *d_table, uint32_t *d_badcount) {
    // points -> raw 160-byte (decompress if needed), multiples by repeated doubling, then the MSM's own normaliser
    hipStream_t st = ctx->stream;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_c2, ns * 160 + ns + 64)) || (r = ctx_reserve(ctx, ctx->tmp_f, (size_t)m.K * ns * 160 + 64))) return r;
    uint8_t *raw = (uint8_t *)ctx->tmp_c2.p, *ok = raw + ns * 160;
    if (in_fmt == C25519_FMT_RAW160) HIPCHK(hipMemcpyAsync(raw, d_points, ns * 160, hipMemcpyDeviceToDevice, st));
    else if (in_fmt == C25519_FMT_EDWARDS_Y) HIPCHK(launch_decompress_edwards(d_points, ns, raw, ok, d_badcount, st));
    else if (in_fmt == C25519_FMT_RISTRETTO) HIPCHK(launch_decompress_ristretto(d_points, ns, raw, ok, d_badcount, st));
    else { ctx->err = "precomp: bad in_fmt"; return -(int32_t)hipErrorInvalidValue; }
    hipLaunchKernelGGL(k_merged_table, dim3(div_up64(ns, 256)), dim3(256), 0, st, raw, ns, m.c, m.K, (uint8_t *)ctx->tmp_f.p);
    HIPCHK(hipGetLastError());
    return prep_points(ctx, (const uint8_t *)ctx->tmp_f.p, (uint64_t)m.K * ns, C25519_FMT_RAW160, d_table, 0, d_badcount + 1);
}
int32_t msm_merged_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const msm_merged &m, const uint32_t *d_table, ge_p3 &R) {
    msm_geom g;
    memset(&g, 0, sizeof g);
    g.c = m.c; g.nwin = 1; g.half = 1 << (m.c - 1); g.first_unsigned = 1;
    msm_sort_params((uint64_t)m.K * m.ns, g);
    g.pos[0] = 0; g.wid[0] = (unsigned char)m.c;
    for (int k = 1; k < MSM_MAX_WIN; k++) g.wid[k] = 1;
    HIPCHK(hipMemsetAsync(dslot(ctx, 0), 0, C25519_SLOT_U32 * 4, ctx->stream));
    msm_plan pl;
    int32_t r = msm_enqueue_sort(ctx, d_scalars, n, g, dslot(ctx, 0), nullptr, pl, &m);
    if (r) return r;
    if ((r = msm_enqueue_acc(ctx, pl, d_table, dslot(ctx, 0), nullptr, nullptr))) return r;
    if ((r = slots_collect(ctx, 1))) return r;
    if (slot_flags((uint32_t *)hslot(ctx, 0))[0]) { ctx->err = "precomp_msm: internal error (top digit out of range)"; return -(int32_t)hipErrorInvalidValue; }
    R = host_p40(hslot(ctx, 0));                          // one window at position 0: the column sum IS the result
    return C25519_OK;
}

// points in any format -> packed affine Niels at d_pts[dst0..]; *d_badcount counts the encodings that do not decode
int32_t prep_points(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_pts, uint64_t dst0, uint32_t *d_badcount) {
    hipStream_t st = ctx->stream;
    if (n == 0) return C25519_OK;
    if (in_fmt == C25519_FMT_EDWARDS_Y) HIPCHK(launch_prep_compressed(0, d_points, 1, n, d_pts, dst0, d_badcount, false, st));
    else if (in_fmt == C25519_FMT_RISTRETTO) HIPCHK(launch_prep_compressed(1, d_points, 1, n, d_pts, dst0, d_badcount, false, st));
    else if (in_fmt == C25519_FMT_RAW160) {
        int32_t r;
        // points per lane and inversion: 64 when the launch still has >= 2048 waves (the records of the later passes of a
        // multi-pass call), 16 for one pass of 2^21 points (2^24 terms: 15.8 ms with 16 everywhere, 15.3 with 64)
        const int CH = n >= (1ull << 23) ? 64 : n >= (1ull << 22) ? 32 : 16;
        constexpr int wpb = 4;
        const unsigned blocks = (unsigned)div_up64((n + CH - 1) / CH, 64 * wpb);
        // the prefix buffer is addressed per wave (CH x 3 x 64 pieces): blocks x wpb waves of them
        r = ctx_reserve(ctx, ctx->prefix, (size_t)blocks * wpb * CH * 3 * 64 * 16);
        if (r) return r;
        if (CH == 64) hipLaunchKernelGGL((k_prep_raw2<64, wpb>), dim3(blocks), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)ctx->prefix.p, d_pts, dst0);
        else if (CH == 32) hipLaunchKernelGGL((k_prep_raw2<32, wpb>), dim3(blocks), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)ctx->prefix.p, d_pts, dst0);
        else hipLaunchKernelGGL((k_prep_raw2<16, wpb>), dim3(blocks), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)ctx->prefix.p, d_pts, dst0);
    } else { ctx->err = "msm: bad in_fmt"; return -(int32_t)hipErrorInvalidValue; }
    HIPCHK(hipGetLastError());
    return C25519_OK;
}

// ---- passes -------------------------------------------------------------------------------------------------------
// The window width stops at c = 16 (the two-pass sort keeps a (window, slice) bin in LDS), so beyond ~2^22 terms the
// lists per bucket only get longer: larger inputs are cut into passes of at most 1.75 M terms -- the same decomposition the
// multi-GPU path uses across ranks.  This also bounds the workspace (~0.5 GB per stream set) for any n.
// Passes are independent and nothing in them waits for the host, so ONE host thread deals them alternately to the
// caller's context and a peer context (own streams and workspaces on the same GPU): the low-VALU two thirds of a pass
// (normalise, sort, reduce) overlap the accumulation of its neighbour.  (Round 1 used a host thread per stream set and
// a stream synchronisation + host fold per pass.)
// Pass size: the 128-byte gather records of a pass should stay resident in the 256 MiB MALL while k_accumulate gathers each of
// them 16 times -- 1.75 M terms = 224 MB (2^21 terms = 268 MB spill: k_accumulate 0.61 - 0.62 ns per term against 0.59 - 0.60, the
// 2^24-term call 14.32 - 14.54 ms in 8 passes against 14.10 - 14.18 in 10, profiles/r03_ab_pass_size.txt; with the bucket
// continuation a pass more costs a sort's fixed part, not a reduction).  C25519_MSM_PASS_LOG2 (tests: many small passes) overrides.
static const uint64_t MSM_PASS = []() -> uint64_t { const char *e = getenv("C25519_MSM_PASS_LOG2"); if (!e) return (uint64_t)1750000; int v = atoi(e); return 1ull << (v < 16 ? 16 : (v > 22 ? 22 : v)); }();
static const uint64_t MSM_PASS_MAX = MSM_PASS + MSM_PASS / 2;
static int pass_lanes() { static const int v = [] { int x = env_int("C25519_PASS_LANES", 2); return x < 1 ? 1 : (x > 4 ? 4 : x); }(); return v; }   // A/B knob: stream sets (2, 3, 4 measure the same within 3 %: the GPU is saturated)

struct pass_set { c25519_ctx *c[4]; int lanes; };
// the peers' streams start after everything already enqueued on the caller's stream (the inputs are complete)
static int32_t passes_begin(c25519_ctx *ctx, uint64_t passes, pass_set &ps) {
    ps.c[0] = ctx; ps.lanes = 1;
    ctx->last_passes.clear();
    const int want = (int)std::min<uint64_t>(passes, (uint64_t)pass_lanes());
    while (ps.lanes < want) { c25519_ctx *p = ctx_peer(ps.c[ps.lanes - 1]); if (!p) break; ps.c[ps.lanes++] = p; }   // each the peer of the previous one
    if (ps.lanes > 1) {
        HIPCHK(hipEventRecord(ctx->ev_in, ctx->stream));
        for (int l = 1; l < ps.lanes; l++) HIPCHK(hipStreamWaitEvent(ps.c[l]->stream, ctx->ev_in, 0));
    }
    return C25519_OK;
}
// the caller's stream continues after the peers' passes
static int32_t passes_join(c25519_ctx *ctx, pass_set &ps) {
    for (int l = 1; l < ps.lanes; l++) {
        HIPCHK(hipEventRecord(ps.c[l]->ev_in, ps.c[l]->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ps.c[l]->ev_in, 0));
    }
    return C25519_OK;
}
static hipEvent_t *pass_ring(c25519_ctx *owner, c25519_ctx *c, uint8_t kind) {
    const int idx = (int)(c->ncalls++ % c25519_ctx::RING);
    c->ring_kind[idx] = kind;
    owner->last_passes.push_back({c, idx});
    return c->ring[idx];
}

// One bucket-method pass over at most MSM_PASS_MAX terms, enqueued on context c (ctx or its peer); results to d_slot.
// Normalisation on the main stream, sort on the second one.
// Where the gather records of the pass come from:
//   ahead == nullptr                the pass prepares its own n points into the context's buffer
//   ahead->launch                   this pass ALSO prepares the points of all later passes (ahead->n points from d_points
//                                   on) into ahead->pts in the same launch, and records ahead->done: 3584 waves instead
//                                   of 512 hide the latency of the normaliser's 64-step chains, and no later pass has
//                                   a normalisation between the reduction before it and its accumulation
//   otherwise                       the records are at ahead->pts + ahead->offset once ahead->done has fired
struct pts_ahead { uint32_t *pts; uint64_t n, offset; hipEvent_t done; bool launch; };
static int32_t msm_pass_enqueue(c25519_ctx *owner, c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, const msm_geom &g, uint64_t terms, uint32_t *d_slot,
                                hipEvent_t wait_acc, const pts_ahead *ahead = nullptr, hipEvent_t wait_in = nullptr,
                                bool cont = false, bool reduce = true, uint64_t n_carve = 0, uint32_t *d_bad_sticky = nullptr) {
    int32_t r;
    uint32_t *d_pts;
    if (wait_in) HIPCHK(hipStreamWaitEvent(ctx->stream, wait_in, 0));      // host-pointer calls: this pass's inputs are still on their way up
    if (!ahead) {
        if ((r = ctx_reserve(ctx, ctx->tmp_e, std::max(n, n_carve) * PTS_BYTES + 256))) return r;
        d_pts = (uint32_t *)ctx->tmp_e.p;
    } else d_pts = ahead->pts + ahead->offset * (PTS_BYTES / 4);
    hipEvent_t *ring = pass_ring(owner, ctx, 1);
    HIPCHK(hipEventRecord(ring[3], ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));     // the sort does not touch the slot: it need not wait for k_slot_init's dispatch
    HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    if (!cont) slot_init(d_slot, terms, nullptr, ctx->stream);            // (a continuing pass adds its counters to the slot of its stream set)
    msm_plan pl;
    pl.bad_sticky = d_bad_sticky;
    // (normalisation first, then the sort on the second stream: 2.26 against 2.34 ms at 2^21 terms the other way round)
    static const int serial_sort = env_int("C25519_PROFILE_SERIAL_SORT", 0);     // profiling: the sort only starts after the normaliser, so that its kernels can be timed alone
    if (!ahead) { if ((r = prep_points(ctx, d_points, n, in_fmt, d_pts, 0, slot_flags(d_slot) + 1))) return r; }
    else if (ahead->launch) {
        if ((r = prep_points(ctx, d_points, ahead->n, in_fmt, ahead->pts, 0, slot_flags(d_slot) + 1))) return r;
        HIPCHK(hipEventRecord(ahead->done, ctx->stream));
    } else HIPCHK(hipStreamWaitEvent(ctx->stream, ahead->done, 0));
    if (serial_sort) { HIPCHK(hipEventRecord(ctx->ev_z, ctx->stream)); HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_z, 0)); }
    if ((r = msm_enqueue_sort(ctx, d_scalars, n, g, d_slot, ctx->aux, pl, nullptr, n_carve))) return r;
    // a continuing pass adds onto the bucket sums its predecessor on this stream set left: they must be where it left them
    if (cont && pl.buckets != ctx->cont_buckets) { ctx->err = "msm: internal error (the workspace of a continuing pass moved its buckets)"; return -(int32_t)hipErrorInvalidValue; }
    ctx->cont_buckets = pl.buckets;
    return msm_enqueue_acc(ctx, pl, d_pts, d_slot, ring, wait_acc, cont, reduce, d_bad_sticky);
}
// The whole MSM, enqueued: every pass on its stream set, the passes' column sums added on the device, the RECORD (column
// sums + counters + header) left at d_record.  Nothing here waits for the host.
// fetch (host-pointer calls, may be null): called right before pass [lo, lo + m) is enqueued; it starts the upload of that pass's
// scalars and points on the copy stream and returns the event the pass has to wait for -- so pass i computes while the inputs
// of pass i+1 travel.  pass_terms (0 = default): terms per pass, smaller for host-pointer calls so that the link and the
// kernels overlap at a finer grain.
typedef std::function<int32_t(uint64_t lo, uint64_t m, hipEvent_t *ready)> msm_fetch;
static int32_t msm_record_enqueue(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_record,
                                  const msm_fetch *fetch = nullptr, uint64_t pass_terms = 0) {
    HIPCHK(hipSetDevice(ctx->device));
    if (in_fmt < 0 || in_fmt > 2) { ctx->err = "msm: bad in_fmt"; return -(int32_t)hipErrorInvalidValue; }
    if (n >= (1ull << 40)) { ctx->err = "msm: n must be < 2^40"; return -(int32_t)hipErrorInvalidValue; }
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    if (n == 0) {                                           // the identity: an empty record (records_fold skips it)
        ctx->last_passes.clear();
        slot_init(d_record, 0, nullptr, ctx->stream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        return C25519_OK;
    }
    const uint64_t PT = pass_terms ? pass_terms : MSM_PASS, PTMAX = pass_terms ? pass_terms + pass_terms / 2 : MSM_PASS_MAX;
    const uint64_t passes = n <= PTMAX ? 1 : (n + PT - 1) / PT, per = (n + passes - 1) / passes;
    const size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    msm_geom g;
    msm_layout(per, g);                                   // one layout for every pass: their column sums add up window by window
    pass_set ps;
    int32_t r;
    uint32_t *sticky = (uint32_t *)ctx->d_flag + 40;      // "a scalar has bit 255 set", ORed over the passes of the call
    HIPCHK(hipMemsetAsync(sticky, 0, 4, ctx->stream));
    if ((r = passes_begin(ctx, passes, ps))) return r;
    hipEvent_t prev_acc = nullptr;                         // the accumulation of the previous pass (on the other stream set)
    // raw points, several passes on two stream sets: pass 1 (the first one on the peer) prepares the records of ALL later
    // passes in one launch beside the sort and the accumulation of pass 0 (pts_ahead; 128 bytes per point stay allocated)
    // (not with a fetch: the later passes' points are not on the device yet)
    const bool ahead = passes > 1 && ps.lanes > 1 && in_fmt == C25519_FMT_RAW160 && !fetch;
    if (ahead && (r = ctx_reserve(ctx, ctx->pts_all, (n - per) * PTS_BYTES + 256))) return r;
    // ONE bucket reduction per stream set, not one per pass: the passes dealt to a stream set run one after the other anyway, so
    // each continues from the bucket sums its predecessor left (k_accumulate `cont`) and only the last one reduces them into the
    // set's slot -- for 2^24 terms 2 reductions instead of 8 (0.24 ms each, and each a handful of small blocks that starved
    // beside the other set's accumulation), and the sort of pass i+2 no longer queues behind the reduction of pass i.
    const int L = ps.lanes;
    for (uint64_t p = 0; p < passes; p++) {
        const uint64_t lo = p * per, m = std::min(per, n - lo);
        const int l = (int)(p % L);
        c25519_ctx *c = ps.c[l];
        const bool first = p < (uint64_t)L, last = p + L >= passes;
        pts_ahead ah = {(uint32_t *)ctx->pts_all.p, n - per, lo - per, ctx->ev_pts, p == 1};
        uint32_t *slot = passes == 1 ? d_record : dslot(ctx, l);         // a single pass writes the record itself
        hipEvent_t in_ev = nullptr;
        if (fetch && (r = (*fetch)(lo, m, &in_ev))) return r;
        if ((r = msm_pass_enqueue(ctx, c, d_scalars + lo * 32, d_points + lo * psz, m, in_fmt, g, per, slot, prev_acc, (ahead && p >= 1) ? &ah : nullptr, in_ev,
                                  !first, last, per, sticky))) {
            if (ctx->err.empty()) ctx->err = c->err;
            return r;
        }
        prev_acc = L > 1 ? c->ev_acc : nullptr;
    }
    if ((r = passes_join(ctx, ps))) return r;
    if (passes > 1) hipLaunchKernelGGL(k_record_sum, dim3(1), dim3(128), 0, ctx->stream, d_record, ctx->d_slots, (int)std::min<uint64_t>((uint64_t)L, passes), g.nwin, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    return C25519_OK;
}
// flags of a folded MSM record -> status
static int32_t msm_record_status(c25519_ctx *ctx, const uint32_t flags[8]) {
    if (flags[0]) { if (ctx) ctx->err = "msm: a scalar has bit 255 set (Scalar invariant #1 violated)"; return -(int32_t)hipErrorInvalidValue; }
    return flags[1] ? C25519_NONE : C25519_OK;            // the status does not depend on the split
}
static int32_t msm_partial_impl(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, ge_p3 &R) {
    int32_t r = msm_record_enqueue(ctx, d_scalars, d_points, n, in_fmt, drec(ctx));
    if (r) return r;
    if ((r = rec_collect(ctx))) return r;
    uint32_t flags[8];
    if ((r = records_fold((const uint8_t *)hslot(ctx, C25519_MAX_SLOTS), 1, R, flags, &ctx->err))) return r;
    return msm_record_status(ctx, flags);
}

// This rank's (context's) share of a sharded MSM as a RECORD in device memory: enqueue only.  The records
// of all ranks are exchanged (one all_gather over RCCL) and folded once by c25519_fold_partial_records.
EXPORT int32_t c25519_msm_partial_record_dev(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, uint8_t *d_record) {
    return msm_record_enqueue(ctx, d_scalars, d_points, n, in_fmt, (uint32_t *)d_record);
}
EXPORT int32_t c25519_fold_partial_records(c25519_ctx *ctx, const uint8_t *records, uint64_t count, int out_fmt, uint8_t *out) {
    if (out_fmt < 0 || out_fmt > 2) { if (ctx) ctx->err = "fold: bad out_fmt"; return -(int32_t)hipErrorInvalidValue; }
    ge_p3 R;
    uint32_t flags[8];
    int32_t r = records_fold(records, count, R, flags, ctx ? &ctx->err : nullptr);
    if (r) return r;
    if ((r = msm_record_status(ctx, flags))) return r;
    host_encode(R, out_fmt, out);
    return C25519_OK;
}
// A record holding a given point (host arithmetic; ctx-less): lets a participant that computed its partial sum elsewhere --
// the host-pointer entry points, a CPU -- join the same fold.  status: C25519_OK or C25519_NONE.
EXPORT int32_t c25519_partial_record_pack(const uint8_t *point160, int32_t status, const uint32_t *counters8, uint8_t *record) {
    if (status != C25519_OK && status != C25519_NONE) return -(int32_t)hipErrorInvalidValue;
    std::vector<uint32_t> rec(C25519_SLOT_U32, 0u);
    msm_geom g;
    msm_layout(1, g);                                     // terms = 1: window 0 sits at bit 0, so column 0 IS the point
    const ge_p3 id = ge_identity(), P = host_from_raw160(point160);
    for (int k = 0; k < g.nwin; k++) {
        const ge_p3 &q = k == 0 ? P : id;
        for (int i = 0; i < 10; i++) { rec[(size_t)k * 40 + i] = q.X.v[i]; rec[(size_t)k * 40 + 10 + i] = q.Y.v[i]; rec[(size_t)k * 40 + 20 + i] = q.Z.v[i]; rec[(size_t)k * 40 + 30 + i] = q.T.v[i]; }
    }
    uint32_t *f = rec.data() + (size_t)MSM_MAX_WIN * 40;
    if (counters8) for (int j = 0; j < 8; j++) f[j] = counters8[j];
    if (status == C25519_NONE) f[1] += 1;
    f[REC_TERMS_LO] = 1; f[REC_PASSES] = 1; f[REC_MAGIC] = REC_MAGIC_VALUE;
    memcpy(record, rec.data(), C25519_PARTIAL_RECORD_BYTES);
    return C25519_OK;
}

EXPORT int32_t c25519_msm_partial_dev(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, uint8_t *out160) {
    ge_p3 R;
    int32_t r = msm_partial_impl(ctx, d_scalars, d_points, n, in_fmt, R);
    if (r != C25519_OK) return r;
    host_raw160(R, out160);
    return C25519_OK;
}
EXPORT int32_t c25519_msm_vartime_dev(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out) {
    if (out_fmt < 0 || out_fmt > 2) { ctx->err = "msm: bad out_fmt"; return -(int32_t)hipErrorInvalidValue; }
    ge_p3 R;
    int32_t r = msm_partial_impl(ctx, d_scalars, d_points, n, in_fmt, R);
    if (r != C25519_OK) return r;
    host_encode(R, out_fmt, out);
    return C25519_OK;
}
EXPORT int32_t c25519_msm_vartime(c25519_ctx *ctx, const uint8_t *scalars, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out) {
    HIPCHK(hipSetDevice(ctx->device));
    if (out_fmt < 0 || out_fmt > 2 || in_fmt < 0 || in_fmt > 2) { ctx->err = "msm: bad format"; return -(int32_t)hipErrorInvalidValue; }
    const size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, n * 32 + 16)) || (r = ctx_reserve(ctx, ctx->tmp_b, n * psz + 16))) return r;
    uint8_t *d_s = (uint8_t *)ctx->tmp_a.p, *d_p = (uint8_t *)ctx->tmp_b.p;
    if ((r = ffi_begin(ctx))) return r;
    ffi_guard guard(ctx);
    // The inputs go up pass by pass on the copy stream while the previous pass computes.  Raw points are 192 bytes per term:
    // the link (56 GB/s = 0.29 G terms/s) is slower than the kernels (1.1 G terms/s), so passes are small (2^19 terms)
    // and what remains after the last byte has arrived is one small pass; compressed points (64 bytes per term) are bound
    // by their decompression instead.
    uint64_t up = 0;
    int slot = 0;
    const msm_fetch fetch = [&](uint64_t lo, uint64_t m, hipEvent_t *ready) -> int32_t {
        HIPCHK(hipMemcpyAsync(d_s + lo * 32, scalars + lo * 32, m * 32, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(hipMemcpyAsync(d_p + lo * psz, points + lo * psz, m * psz, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(hipEventRecord(ctx->ev_up[slot], ctx->s_h2d));
        *ready = ctx->ev_up[slot];
        slot = (slot + 1) % c25519_ctx::FFI_MAXCH;
        up += m * (32 + psz);
        return C25519_OK;
    };
    ge_p3 R;
    uint32_t flags[8];
    const uint64_t pass_terms = n >= (1ull << 20) ? (in_fmt == C25519_FMT_RAW160 ? (1ull << 19) : (1ull << 20)) : 0;
    r = msm_record_enqueue(ctx, d_s, d_p, n, in_fmt, drec(ctx), &fetch, pass_terms);
    if (!r) r = rec_collect(ctx);
    guard.dismiss();
    const int32_t r2 = ffi_end(ctx, up, 0);
    if (r || (r = r2)) return r;
    if ((r = records_fold((const uint8_t *)hslot(ctx, C25519_MAX_SLOTS), 1, R, flags, &ctx->err))) return r;
    if ((r = msm_record_status(ctx, flags))) return r;
    host_encode(R, out_fmt, out);
    return C25519_OK;
}
// fold of per-rank partial sums: plain complete additions, identical on every rank
EXPORT int32_t c25519_fold_partials(c25519_ctx *ctx, const uint8_t *partials160, uint64_t count, int out_fmt, uint8_t *out) {
    // pure host arithmetic over <= world_size points: ctx may be NULL (no GPU is touched)
    if (out_fmt < 0 || out_fmt > 2) { if (ctx) ctx->err = "fold: bad out_fmt"; return -(int32_t)hipErrorInvalidValue; }
    ge_p3 acc = ge_identity();
    for (uint64_t i = 0; i < count; i++) acc = ge_add(acc, host_from_raw160(partials160 + 160 * i));
    host_encode(acc, out_fmt, out);
    return C25519_OK;
}

// ---- verify_batch ---------------------------------------------------------------------------------------
#include "transcript_host.h"

// IVs of the z tree: iv[l] = SHA-512 chaining value after the one-block tag of level l (see k_ztree_first)
static void ztree_make_ivs(uint64_t n, ztree_ivs &ivs) {
    uint64_t count = n;                                  // inputs of level 0: signatures
    for (int l = 0; l < ZTREE_MAX_LEVELS; l++) {
        u64 w[16] = {0};
        const char tag[] = "c25519-hip/verify_batch/z-tree/v4";
        static_assert(sizeof(tag) - 1 <= 64, "tag fits the first half of the block");
        uint8_t blk[128] = {0};
        memcpy(blk, tag, sizeof(tag) - 1);
        for (int q = 0; q < 16; q++) { u64 v = 0; for (int b = 0; b < 8; b++) v = (v << 8) | blk[8 * q + b]; w[q] = v; }
        w[13] = (u64)l; w[14] = count; w[15] = n;        // level, number of inputs of this level, batch size
        sha512_init(ivs.iv[l]);
        sha512_compress(ivs.iv[l], w);
        count = (count + 3) / 4;
    }
}
// the z_i of one pass (n signatures) by the device derivation; z16: room for 4 * ceil(n/4) entries.  t0 / t1: tree scratch
// ((n/4 + 1) * 32 bytes each).  Enqueued on `sa`.
static int32_t zchain_enqueue(c25519_ctx *ctx, hipStream_t sa, const uint8_t *hred, const uint8_t *d_sigs, uint64_t n, uint8_t *t0, uint8_t *t1, uint8_t *z16) {
    ztree_ivs ivs;
    ztree_make_ivs(n, ivs);
    uint64_t mm = (n + 3) / 4; uint8_t *a = t0, *b = t1;
    uint32_t level = 1;
    hipLaunchKernelGGL(k_ztree_first, dim3(div_up64(mm, 256)), dim3(256), 0, sa, hred, d_sigs, n, ivs, a);
    while (mm > 1024) {
        uint64_t mo = (mm + 3) / 4;
        hipLaunchKernelGGL(k_ztree, dim3(div_up64(mo, 256)), dim3(256), 0, sa, a, mm, level, ivs, b);
        mm = mo; level++; std::swap(a, b);
    }
    hipLaunchKernelGGL(k_ztree_tail, dim3(1), dim3(256), 0, sa, a, mm, level, ivs, b);      // leaves the 32-byte root at b
    hipLaunchKernelGGL(k_zderive, dim3(div_up64((n + 3) / 4, 256)), dim3(256), 0, sa, b, (n + 3) / 4, z16);
    HIPCHK(hipGetLastError());
    return C25519_OK;
}

// One random-linear-combination check over at most VERIFY_PASS_MAX signatures (an MSM of 2n+1 terms), enqueued on
// context ctx (the caller's or its peer); column sums and counters go to d_slot, nothing waits for the host.
// d_pk_points (may be NULL): the keys' decompressed points, n x 160 raw -- what VerifyingKey carries beside its bytes
// (verifying.rs:64-71), so that, like the reference (batch.rs:236), the batch does not decompress A_i again.
// d_hram_pre / d_z_pre (transcript z-mode): H(R||A||M) and the z_i of these signatures, computed over the whole batch.
// stage (host-pointer calls, may be null): called right before the first kernels that need an input array are enqueued, in the
// order 0 = signatures, 1 = key bytes, 2 = messages + offsets, 3 = the keys' points (only if given); it starts the upload of
// that array's slice for THIS pass on the copy stream and returns the event to wait for.  So R_i is being decompressed
// while the keys and messages travel, and the hash chain runs while the (five times larger) key points travel.
typedef std::function<int32_t(int what, hipEvent_t *ready)> verify_stage;
static int32_t verify_pass_enqueue(c25519_ctx *owner, c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                   const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, uint64_t n, uint32_t z_mode,
                                   const uint8_t *d_hram_pre, const uint8_t *d_z_pre, const uint32_t *d_pre_flags, const msm_geom &g, uint64_t terms, uint32_t *d_slot, hipEvent_t wait_acc,
                                   const verify_stage *stage = nullptr) {
    hipStream_t st = ctx->stream;
    const uint64_t m = 2 * n + 1;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_e, m * PTS_BYTES + 256))) return r;
    // tmp_f: hram (64n) | z16 (16n) | msm scalars (32m) | tree scratch | partial sums
    const unsigned nblk = div_up64(n, 256);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t oH = carve(n * 64), oZ = carve((n + 4) * 16), oSc = carve(m * 32), oT0 = carve((n / 4 + 2) * 32), oT1 = carve((n / 4 + 2) * 32), oP = carve((size_t)nblk * 40);
    const size_t oHr = carve(d_z_pre ? 0 : n * 32);       // h_i mod l, for the device z-tree
    if ((r = ctx_reserve(ctx, ctx->tmp_f, off))) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_f.p;
    uint8_t *hram = ws + oH, *z16 = ws + oZ, *msc = ws + oSc, *t0 = ws + oT0, *t1 = ws + oT1, *hred = d_z_pre ? nullptr : ws + oHr;
    uint32_t *partial = (uint32_t *)(ws + oP);
    uint32_t *d_pts = (uint32_t *)ctx->tmp_e.p;
    uint32_t *d_cnt = slot_flags(d_slot);             // [2] bad A, [3] bad R, [4] bad s, [5] bad offsets
    hipEvent_t *ring = pass_ring(owner, ctx, 2);
    HIPCHK(hipEventRecord(ring[3], st));
    slot_init(d_slot, terms, d_pre_flags, st);
    // Two independent chains: (S) decompress R_i and A_i -- VALU-bound; (A) hash, derive z_i, batch scalars, sort --
    // partly latency-bound (the tree levels).  They run on two streams and join before the accumulation.
    hipStream_t sa = ctx->aux;
    HIPCHK(hipEventRecord(ctx->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(sa, ctx->ev_fork, 0));
    hipEvent_t ev_sig = nullptr, ev_pk = nullptr, ev_msg = nullptr, ev_pts = nullptr;
    // (S) points: [0] = B, [1..n] = R_i, [n+1..2n] = A_i     (batch.rs:235-244)
    hipLaunchKernelGGL(k_prep_basepoint, dim3(1), dim3(64), 0, st, d_pts, (uint64_t)0);
    auto prep_A = [&]() -> int32_t {
        if (d_pk_points) {
            if (stage) { int32_t q = (*stage)(3, &ev_pts); if (q) return q; HIPCHK(hipStreamWaitEvent(st, ev_pts, 0)); }
            return prep_points(ctx, d_pk_points, n, C25519_FMT_RAW160, d_pts, n + 1, d_cnt + 2);
        }
        HIPCHK(launch_prep_compressed(0, d_pks, 1, n, d_pts, n + 1, d_cnt + 2, true, st));
        return C25519_OK;
    };
    auto prep_R = [&]() -> int32_t {   // R_i = the first half of every 64-byte signature (stride 2)
        HIPCHK(hipEventRecord(ring[4], st));
        ctx->kname[1] = "c25519::k_prep_compressed<0> (decompression of R_i)";
        HIPCHK(launch_prep_compressed(0, d_sigs, 2, n, d_pts, 1, d_cnt + 3, true, st));
        HIPCHK(hipEventRecord(ring[5], st));
        return C25519_OK;
    };
    if (stage) {
        // host-pointer call: in the order the inputs arrive -- signatures, then R_i decompresses while keys and messages travel
        if ((r = (*stage)(0, &ev_sig))) return r;
        HIPCHK(hipStreamWaitEvent(st, ev_sig, 0));
        if ((r = prep_R())) return r;
        if ((r = (*stage)(1, &ev_pk)) || (r = (*stage)(2, &ev_msg))) return r;
        HIPCHK(hipStreamWaitEvent(sa, ev_sig, 0)); HIPCHK(hipStreamWaitEvent(sa, ev_pk, 0)); HIPCHK(hipStreamWaitEvent(sa, ev_msg, 0));
        if (!d_pk_points) { HIPCHK(hipStreamWaitEvent(st, ev_pk, 0)); if ((r = prep_A())) return r; }
    } else {
        if ((r = prep_A()) || (r = prep_R())) return r;
    }
    // (A)
    const uint8_t *hr = d_hram_pre;
    if (!hr) { hipLaunchKernelGGL(k_hram, dim3(nblk), dim3(256), 0, sa, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, hram, d_cnt + 4, hred); hr = hram; }
    else if (hred) hipLaunchKernelGGL(k_hram_mod_l, dim3(nblk), dim3(256), 0, sa, hr, n, hred);
    HIPCHK(hipGetLastError());
    const uint8_t *zz = d_z_pre;
    if (!zz) {
        if ((r = zchain_enqueue(ctx, sa, hred, d_sigs, n, t0, t1, z16))) return r;
        zz = z16;
        // the sign of z_i goes onto the stored R_i (main stream, beside the sort on the second one)
        HIPCHK(hipEventRecord(ctx->ev_z, sa));
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_z, 0));
        hipLaunchKernelGGL(k_apply_sign, dim3(nblk), dim3(256), 0, st, d_pts, (uint64_t)1, zz, n);
    }
    hipLaunchKernelGGL(k_batch_scalars, dim3(nblk), dim3(256), 0, sa, hr, d_sigs, zz, n, z_mode == C25519_Z_DEVICE ? 1 : 0, msc, partial);
    // the basepoint coefficient -sum z_i s_i (batch.rs:240): the per-block partial sums are folded by one more block
    hipLaunchKernelGGL(k_bsum_finish, dim3(1), dim3(256), 0, sa, partial, nblk, msc);
    HIPCHK(hipGetLastError());
    if (stage && d_pk_points && (r = prep_A())) return r;   // the keys' points come last: only the accumulation needs them
    // the MSM's digit/sort phase continues on the second stream while (S) is still decompressing
    return msm_enqueue(ctx, msc, m, d_pts, g, d_slot, ring, sa, wait_acc);
}
// Batches beyond ~1.5 * 2^20 signatures are checked as several independent random linear combinations of about
// 2^20 signatures each (same reason as MSM_PASS_MAX; in the device z-mode every pass derives its own z_i from its own
// tree; in the transcript z-mode the z_i come from ONE transcript over the whole batch, exactly the reference's).  Every
// pass keeps its own identity check.  All passes run even after a failure so that the reference's precedence -- key
// decoding, then ScalarFormat for ANY non-canonical s (batch.rs:208-211), then Verify -- does not depend on where the
// batch was cut.
static const int VERIFY_PASS_LOG2 = [] { int v = env_int("C25519_VERIFY_PASS_LOG2", 20); return v < 15 ? 15 : (v > 21 ? 21 : v); }();   // A/B knob
static const uint64_t VERIFY_PASS = 1ull << VERIFY_PASS_LOG2, VERIFY_PASS_MAX = 3ull << (VERIFY_PASS_LOG2 - 1);
// flags of a folded verify_batch record + its point -> the reference's verdict (precedence: key decoding, then ScalarFormat
// for ANY non-canonical s, batch.rs:208-211, then Verify, :244-250)
static int32_t verify_record_verdict(c25519_ctx *ctx, const ge_p3 &R, const uint32_t flags[8]) {
    if (flags[5]) { if (ctx) ctx->err = "verify_batch: msg_off is not monotone or runs past msgs_len"; return -(int32_t)hipErrorInvalidValue; }
    if (flags[0]) { if (ctx) ctx->err = "verify_batch: internal error (batch scalar with bit 255 set)"; return -(int32_t)hipErrorInvalidValue; }
    if (flags[2]) return C25519_NONE;                       // a key that VerifyingKey::from_bytes rejects
    if (flags[4]) return C25519_SCALAR_FORMAT;
    if (flags[3]) return C25519_VERIFY;                     // batch.rs:244 (an R that fails to decompress)
    return ge_is_identity(R) ? C25519_OK : C25519_VERIFY;   // batch.rs:246-250
}
// Every pass of a batch whose z_i are GIVEN (transcript z-mode: d_hram = H(R||A||M) of these n signatures followed by a
// 64-byte trailer of counters -- [0] non-canonical s, [1] bad offsets -- as ed25519_batch_hram_dev leaves them; d_z16 = their
// z_i), summed into ONE record at d_record: the reference's single equation (batch.rs:235-250) whatever the pass split.
static int32_t verify_record_enqueue(c25519_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, const uint8_t *d_hram, const uint8_t *d_z16,
                                     uint64_t n, uint32_t *d_record) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n >= (1ull << 40)) { ctx->err = "verify_batch: n too large"; return -(int32_t)hipErrorInvalidValue; }
    const uint32_t *d_pre = (const uint32_t *)(d_hram + n * 64);
    if (n == 0) { ctx->last_passes.clear(); slot_init(d_record, 0, d_pre, ctx->stream); HIPCHK(hipGetLastError()); return C25519_OK; }
    const uint64_t passes = n <= VERIFY_PASS_MAX ? 1 : (n + VERIFY_PASS - 1) / VERIFY_PASS, per = (n + passes - 1) / passes;
    msm_geom g;
    msm_layout(2 * per + 1, g);
    int32_t r;
    pass_set ps;
    if ((r = passes_begin(ctx, passes, ps))) return r;
    hipEvent_t prev_acc = nullptr;
    for (uint64_t p0 = 0; p0 < passes; p0 += C25519_MAX_SLOTS) {
        const int cnt = (int)std::min<uint64_t>(C25519_MAX_SLOTS, passes - p0);
        if (p0 && ps.lanes > 1) {
            HIPCHK(hipEventRecord(ctx->ev_in, ctx->stream));
            for (int l = 1; l < ps.lanes; l++) HIPCHK(hipStreamWaitEvent(ps.c[l]->stream, ctx->ev_in, 0));
        }
        for (int i = 0; i < cnt; i++) {
            const uint64_t lo = (p0 + i) * per, m = std::min(per, n - lo);
            c25519_ctx *c = ps.c[(p0 + i) % ps.lanes];
            uint32_t *slot = passes == 1 ? d_record : dslot(ctx, i);
            r = verify_pass_enqueue(ctx, c, nullptr, nullptr, 0, d_sigs + lo * 64, d_pks + lo * 32, d_pk_points ? d_pk_points + lo * 160 : nullptr, m, C25519_Z_TRANSCRIPT,
                                    d_hram + lo * 64, d_z16 + lo * 16, (p0 + i == 0) ? d_pre : nullptr, g, 2 * per + 1, slot, prev_acc);
            if (r) { if (ctx->err.empty()) ctx->err = c->err; return r; }
            prev_acc = ps.lanes > 1 ? c->ev_acc : nullptr;
        }
        if ((r = passes_join(ctx, ps))) return r;
        if (passes > 1) hipLaunchKernelGGL(k_record_sum, dim3(1), dim3(128), 0, ctx->stream, d_record, ctx->d_slots, cnt, g.nwin, p0 == 0 ? 1 : 0);
    }
    HIPCHK(hipGetLastError());
    return C25519_OK;
}
// H(R_i || A_i || M_i) of n signatures to d_hram (n x 64 bytes) followed by a 64-byte trailer of counters ([0] signatures
// with a non-canonical s, [1] bad message offsets): the per-signature half of the transcript z-mode, enqueue only.
static int32_t batch_hram_enqueue(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len, const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, uint8_t *d_hram) {
    HIPCHK(hipSetDevice(ctx->device));
    uint32_t *fl = (uint32_t *)(d_hram + n * 64);
    HIPCHK(hipMemsetAsync(fl, 0, 64, ctx->stream));
    HIPCHK(launch_hram(d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, d_hram, fl, ctx->stream));
    return C25519_OK;
}
EXPORT int32_t ed25519_batch_hram_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len, const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n,
                                      uint8_t *d_hram) {
    return batch_hram_enqueue(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, d_hram);
}
// the reference's z_i from the bytes its transcript absorbs (batch.rs:168-222): host arithmetic, no context, sequential
EXPORT int32_t ed25519_batch_transcript_zs(const uint8_t *hram, const uint8_t *sigs, uint64_t n, uint8_t *z16) {
    c25519_transcript_zs(hram, sigs, n, z16);
    return C25519_OK;
}
EXPORT int32_t ed25519_verify_batch_record_dev(c25519_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, const uint8_t *d_hram, const uint8_t *d_z16,
                                               uint64_t n, uint8_t *d_record) {
    return verify_record_enqueue(ctx, d_sigs, d_pks, d_pk_points, d_hram, d_z16, n, (uint32_t *)d_record);
}
EXPORT int32_t ed25519_fold_verify_records(c25519_ctx *ctx, const uint8_t *records, uint64_t count) {
    ge_p3 R;
    uint32_t flags[8];
    int32_t r = records_fold(records, count, R, flags, ctx ? &ctx->err : nullptr);
    if (r) return r;
    return verify_record_verdict(ctx, R, flags);
}

// pass_stage (host-pointer calls, device z-mode; may be null): (first signature of the pass, its length, which array, event out)
typedef std::function<int32_t(uint64_t lo, uint64_t m, int what, hipEvent_t *ready)> verify_fetch;
static int32_t verify_batch_impl(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                 const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, uint64_t n, uint32_t z_mode, const verify_fetch *fetch) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;                      // batch.rs: 1-term MSM 0*B = identity
    if (n >= (1ull << 40)) { ctx->err = "verify_batch: n too large"; return -(int32_t)hipErrorInvalidValue; }
    if (z_mode > 1) { ctx->err = "verify_batch: bad z_mode"; return -(int32_t)hipErrorInvalidValue; }
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    int32_t r;
    if (z_mode == C25519_Z_TRANSCRIPT) {
        // the reference's sequential Merlin transcript (batch.rs:168-222) over the WHOLE batch, on one host core; then ONE
        // equation over the whole batch (the passes' column sums are added on the device), exactly batch.rs:235-250
        try {
            if ((r = ctx_reserve(ctx, ctx->tmp_c2, n * 80 + 128))) return r;
            uint8_t *d_hram_all = (uint8_t *)ctx->tmp_c2.p, *d_z_all = d_hram_all + n * 64 + 64;
            if ((r = batch_hram_enqueue(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, d_hram_all))) return r;
            std::vector<uint8_t> hh(n * 64), hs(n * 64), hz(n * 16);
            HIPCHK(hipMemcpyAsync(hh.data(), d_hram_all, n * 64, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(hs.data(), d_sigs, n * 64, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            c25519_transcript_zs(hh.data(), hs.data(), n, hz.data());
            HIPCHK(hipMemcpyAsync(d_z_all, hz.data(), n * 16, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));      // hz is a local buffer
            if ((r = verify_record_enqueue(ctx, d_sigs, d_pks, d_pk_points, d_hram_all, d_z_all, n, drec(ctx)))) return r;
        } catch (const std::exception &e) { ctx->err = std::string("verify_batch: ") + e.what(); return -(int32_t)hipErrorOutOfMemory; }
        if ((r = rec_collect(ctx))) return r;
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        ge_p3 R;
        uint32_t flags[8];
        if ((r = records_fold((const uint8_t *)hslot(ctx, C25519_MAX_SLOTS), 1, R, flags, &ctx->err))) return r;
        return verify_record_verdict(ctx, R, flags);
    }
    // device z-mode: every pass derives its own z_i from its own tree and is its own random linear combination (summing
    // passes with independent z_i would open a 2^126 birthday attack across passes); all passes run even after a failure so
    // that the precedence does not depend on where the batch was cut
    const uint64_t passes = n <= VERIFY_PASS_MAX ? 1 : (n + VERIFY_PASS - 1) / VERIFY_PASS, per = (n + passes - 1) / passes;
    msm_geom g;
    msm_layout(2 * per + 1, g);
    pass_set ps;
    if ((r = passes_begin(ctx, passes, ps))) return r;
    bool seen[5] = {false, false, false, false, false}, bad_off = false, bad_scalar = false;
    hipEvent_t prev_acc = nullptr;
    for (uint64_t p0 = 0; p0 < passes; p0 += C25519_MAX_SLOTS) {
        const int cnt = (int)std::min<uint64_t>(C25519_MAX_SLOTS, passes - p0);
        for (int i = 0; i < cnt; i++) {
            const uint64_t lo = (p0 + i) * per, m = std::min(per, n - lo);
            c25519_ctx *c = ps.c[(p0 + i) % ps.lanes];
            const verify_stage stage = [&](int what, hipEvent_t *ready) -> int32_t { return (*fetch)(lo, m, what, ready); };
            r = verify_pass_enqueue(ctx, c, d_msgs, d_msg_off + lo, msgs_len, d_sigs + lo * 64, d_pks + lo * 32, d_pk_points ? d_pk_points + lo * 160 : nullptr, m, z_mode,
                                    nullptr, nullptr, nullptr, g, 2 * per + 1, dslot(ctx, i), prev_acc, fetch ? &stage : nullptr);
            if (r) { if (ctx->err.empty()) ctx->err = c->err; return r; }
            prev_acc = ps.lanes > 1 ? c->ev_acc : nullptr;
        }
        if ((r = passes_join(ctx, ps)) || (r = slots_collect(ctx, cnt))) return r;
        for (int i = 0; i < cnt; i++) {
            const uint32_t *s = hslot(ctx, i), *f = s + MSM_MAX_WIN * 40;
            if (f[0]) bad_scalar = true;
            if (f[5]) bad_off = true;
            uint32_t fl[8] = {0, 0, f[2], f[3], f[4], 0, 0, 0};
            const bool clean = !(f[2] | f[3] | f[4]);
            seen[verify_record_verdict(nullptr, clean ? msm_horner(s, g) : ge_identity(), fl)] = true;
        }
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (bad_off) { ctx->err = "verify_batch: msg_off is not monotone or runs past msgs_len"; return -(int32_t)hipErrorInvalidValue; }
    if (bad_scalar) { ctx->err = "verify_batch: internal error (batch scalar with bit 255 set)"; return -(int32_t)hipErrorInvalidValue; }
    return seen[C25519_NONE] ? C25519_NONE : seen[C25519_SCALAR_FORMAT] ? C25519_SCALAR_FORMAT : seen[C25519_VERIFY] ? C25519_VERIFY : C25519_OK;
}

EXPORT int32_t ed25519_verify_batch_keys_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                             const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, uint64_t n, uint32_t z_mode) {
    return verify_batch_impl(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, d_pk_points, n, z_mode, nullptr);
}
EXPORT int32_t ed25519_verify_batch_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                        const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, uint32_t z_mode) {
    return ed25519_verify_batch_keys_dev(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, nullptr, n, z_mode);
}
// host-side check of the offsets array (the _dev entry points check on the device, inside k_hram)
static bool offsets_ok(const uint64_t *msg_off, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) if (msg_off[i] > msg_off[i + 1]) return false;
    return true;
}
EXPORT int32_t ed25519_verify_batch_keys(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks,
                                         const uint8_t *pk_points, uint64_t n, uint32_t z_mode) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;
    if (z_mode > 1) { ctx->err = "verify_batch: bad z_mode"; return -(int32_t)hipErrorInvalidValue; }
    if (!offsets_ok(msg_off, n)) { ctx->err = "verify_batch: msg_off is not monotone"; return -(int32_t)hipErrorInvalidValue; }
    const uint64_t mlen = msg_off[n];
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, mlen + 64)) || (r = ctx_reserve(ctx, ctx->tmp_b, (n + 1) * 8)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 64)) ||
        (r = ctx_reserve(ctx, ctx->scratch, n * 32 + (pk_points ? n * 160 : 0) + 16)))
        return r;
    uint8_t *d_msg = (uint8_t *)ctx->tmp_a.p, *d_sig = (uint8_t *)ctx->tmp_c.p, *d_pk = (uint8_t *)ctx->scratch.p, *d_pp = pk_points ? d_pk + n * 32 : nullptr;
    uint64_t *d_off = (uint64_t *)ctx->tmp_b.p;
    if ((r = ffi_begin(ctx))) return r;
    ffi_guard guard(ctx);                                 // the early exits of the uploads below drain the copy stream as well
    uint64_t up = 0;
    if (z_mode == C25519_Z_TRANSCRIPT) {
        // the whole batch is hashed before anything else can start and the sequential host transcript dominates: upload everything
        if (mlen) HIPCHK(hipMemcpyAsync(d_msg, msgs, mlen, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(hipMemcpyAsync(d_off, msg_off, (n + 1) * 8, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(hipMemcpyAsync(d_sig, sigs, n * 64, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(hipMemcpyAsync(d_pk, pks, n * 32, hipMemcpyHostToDevice, ctx->s_h2d));
        if (pk_points) HIPCHK(hipMemcpyAsync(d_pp, pk_points, n * 160, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(hipEventRecord(ctx->ev_up[0], ctx->s_h2d));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_up[0], 0));
        up = mlen + (n + 1) * 8 + n * 96 + (pk_points ? n * 160 : 0);
        r = verify_batch_impl(ctx, d_msg, d_off, mlen, d_sig, d_pk, d_pp, n, z_mode, nullptr);
    } else {
        // device z-mode: every array goes up right before the first kernels that need it (verify_pass_enqueue), pass by pass
        int slot = 0;
        const verify_fetch fetch = [&](uint64_t lo, uint64_t m, int what, hipEvent_t *ready) -> int32_t {
            if (what == 0) { HIPCHK(hipMemcpyAsync(d_sig + lo * 64, sigs + lo * 64, m * 64, hipMemcpyHostToDevice, ctx->s_h2d)); up += m * 64; }
            else if (what == 1) { HIPCHK(hipMemcpyAsync(d_pk + lo * 32, pks + lo * 32, m * 32, hipMemcpyHostToDevice, ctx->s_h2d)); up += m * 32; }
            else if (what == 2) {
                // the kernels index the blob through ABSOLUTE offsets: this pass's offsets and the bytes they span, in place
                const uint64_t b0 = msg_off[lo], b1 = msg_off[lo + m];
                if (b1 > b0) HIPCHK(hipMemcpyAsync(d_msg + b0, msgs + b0, b1 - b0, hipMemcpyHostToDevice, ctx->s_h2d));
                HIPCHK(hipMemcpyAsync(d_off + lo, msg_off + lo, (m + 1) * 8, hipMemcpyHostToDevice, ctx->s_h2d));
                up += (b1 - b0) + (m + 1) * 8;
            } else { HIPCHK(hipMemcpyAsync(d_pp + lo * 160, pk_points + lo * 160, m * 160, hipMemcpyHostToDevice, ctx->s_h2d)); up += m * 160; }
            HIPCHK(hipEventRecord(ctx->ev_up[slot], ctx->s_h2d));
            *ready = ctx->ev_up[slot];
            slot = (slot + 1) % c25519_ctx::FFI_MAXCH;
            return C25519_OK;
        };
        r = verify_batch_impl(ctx, d_msg, d_off, mlen, d_sig, d_pk, d_pp, n, z_mode, &fetch);
    }
    guard.dismiss();
    const int32_t r2 = ffi_end(ctx, up, 0);
    return (r < 0 || !r2) ? r : r2;
}
EXPORT int32_t ed25519_verify_batch(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks,
                                    uint64_t n, uint32_t z_mode) {
    return ed25519_verify_batch_keys(ctx, msgs, msg_off, sigs, pks, nullptr, n, z_mode);
}

// diagnostics: the z_i a batch of n <= VERIFY_PASS_MAX signatures gets (16 bytes each to the HOST buffer out_z16;
// device z-mode: sign-magnitude, see k_zderive).  For the tests that pin the derivation's dependence on every input.
EXPORT int32_t c25519_debug_batch_zs(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks, uint64_t n,
                                     uint32_t z_mode, uint8_t *out_z16) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;
    if (n > VERIFY_PASS_MAX || z_mode > 1 || !offsets_ok(msg_off, n)) { ctx->err = "debug_batch_zs: bad arguments"; return -(int32_t)hipErrorInvalidValue; }
    const uint64_t mlen = msg_off[n];
    int32_t r;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t oM = carve(mlen + 64), oO = carve((n + 1) * 8), oS = carve(n * 64), oK = carve(n * 32), oH = carve(n * 64), oZ = carve((n + 4) * 16), oT0 = carve((n / 4 + 2) * 32), oT1 = carve((n / 4 + 2) * 32);
    const size_t oHr = carve(n * 32);
    if ((r = ctx_reserve(ctx, ctx->tmp_f, off))) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_f.p;
    hipStream_t st = ctx->stream;
    if (mlen) HIPCHK(hipMemcpyAsync(ws + oM, msgs, mlen, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ws + oO, msg_off, (n + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ws + oS, sigs, n * 64, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ws + oK, pks, n * 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(ctx->d_flag, 0, 16, st));
    HIPCHK(launch_hram(ws + oM, (const uint64_t *)(ws + oO), mlen, ws + oS, ws + oK, n, ws + oH, (uint32_t *)ctx->d_flag, st));
    if (z_mode == C25519_Z_DEVICE) {
        hipLaunchKernelGGL(k_hram_mod_l, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws + oH, n, ws + oHr);
        if ((r = zchain_enqueue(ctx, st, ws + oHr, ws + oS, n, ws + oT0, ws + oT1, ws + oZ))) return r;
        HIPCHK(hipMemcpyAsync(out_z16, ws + oZ, n * 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    } else {
        std::vector<uint8_t> hh(n * 64);
        HIPCHK(hipMemcpyAsync(hh.data(), ws + oH, n * 64, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        c25519_transcript_zs(hh.data(), sigs, n, out_z16);
    }
    return C25519_OK;
}
