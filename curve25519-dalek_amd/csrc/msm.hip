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
//   reduce    sum_b (b+1) B_b by an 8-ary hierarchy of running sums (pippenger.rs:146-151 per segment); the narrow
//             upper levels use eight lanes per segment
//   fold      total.mul_by_pow_2(w_k) + column (pippenger.rs:159) over the window sums: on the host, through the
//             same ge26.h formulas (a serial chain of ~250 doublings is a latency-bound tail that a single CPU core
//             finishes faster than a single GPU lane)
//   passes    inputs beyond 3 * 2^20 terms are cut into passes of ~2^21 terms (the multi-GPU decomposition, in time)
//
// Window width c is chosen per call from n (reference: w = 6/7/8, pippenger.rs:81-87).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "../../include/c25519_hip.h"
#include "devio.h"
#include "sc_sha.h"
#include "kernels.h"
#include "ctx.h"
#include "msm_internal.h"

using namespace c25519;
#define EXPORT extern "C" __attribute__((visibility("default")))
#define HIPCHK(call)                                                \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return c25519_fail(ctx, _e, #call);   \
    } while (0)

namespace c25519 {

// ================================================================================================
// prep kernels
// ================================================================================================
// (compressed inputs: k_prep_compressed lives in kernels.hip -- a 252-squaring chain per lane at full occupancy wants
// the chained-carry field arithmetic of that translation unit; launch_prep_compressed)
// raw 160-byte points: Montgomery-trick normalisation, CH points per lane (cf. k_compress_p32)
template <int CH>
__global__ void __launch_bounds__(256) k_prep_raw(const uint8_t *__restrict__ in, u64 n, u32 *__restrict__ prefix, u32 *__restrict__ pts, u64 dst0) {
    const u64 T = (u64)gridDim.x * blockDim.x, t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    feT acc = fe_one();
    bool affine = true;                 // every Z of this lane is literally 1 (points straight from a decompression,
                                        // e.g. VerifyingKey.point): the shared inversion is skipped
#pragma unroll 1
    for (int j = 0; j < CH; j++) {
        u64 idx = t + (u64)j * T;
        if (idx >= n) break;
        affine = affine && raw160_z_is_one(in, idx);
        uint4 *q = reinterpret_cast<uint4 *>(prefix) + 3 * idx;
        q[0] = make_uint4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]); q[1] = make_uint4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]);
        q[2] = make_uint4(acc.v[8], acc.v[9], 0u, 0u);
        acc = fe_mul(acc, raw160_fe(in, idx, 2));
    }
    feT inv = fe_one();
    if (!affine) inv = fe_invert(acc);
#pragma unroll 1
    for (int j = CH - 1; j >= 0; j--) {
        u64 idx = t + (u64)j * T;
        if (idx >= n) continue;
        const uint4 *q = reinterpret_cast<const uint4 *>(prefix) + 3 * idx;
        uint4 a = q[0], b = q[1], c = q[2];
        feT pre;
        pre.v[0] = a.x; pre.v[1] = a.y; pre.v[2] = a.z; pre.v[3] = a.w; pre.v[4] = b.x; pre.v[5] = b.y; pre.v[6] = b.z; pre.v[7] = b.w;
        pre.v[8] = c.x; pre.v[9] = c.y;
        feT Z = raw160_fe(in, idx, 2);
        feT zi = fe_mul(inv, pre);
        inv = fe_mul(inv, Z);
        pts_store(pts, dst0 + idx, fe_mul(raw160_fe(in, idx, 0), zi), fe_mul(raw160_fe(in, idx, 1), zi));
    }
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
constexpr int MSM_MAX_WIN = 56;
struct msm_geom { int c, nwin, half; u32 addk[8]; unsigned char pos[MSM_MAX_WIN], wid[MSM_MAX_WIN]; };

// D[k][t] = window k of s' = s + addk  (u16); flags bit 255 of any scalar
__global__ void __launch_bounds__(256) k_digits(const uint8_t *__restrict__ scalars, u64 n, msm_geom g, uint16_t *__restrict__ D, u32 *__restrict__ bad_scalar) {
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
// signed digit of window k from the stored value
// (a top digit above `half` can only come from a scalar with bit 255 set: k_digits has flagged it and the call fails;
// it is dropped here so that no kernel indexes past its tables)
__device__ __forceinline__ int digit_of(u32 v, int k, const msm_geom &g) {
    return (k >= g.nwin - 2) ? (v <= (u32)g.half ? (int)v : 0) : (int)v - (1 << (g.wid[k] - 1));     // the top two windows are unsigned
}

// histogram of bucket occupancy for (window k = blockIdx.y, chunk j = blockIdx.x)
template <bool XCD_SWAP>
__global__ void __launch_bounds__(1024) k_hist(const uint16_t *__restrict__ D, u64 n, msm_geom g, u64 chunk, u32 *__restrict__ counts) {
    extern __shared__ u32 hist[];
    const int k = XCD_SWAP ? blockIdx.x : blockIdx.y, j = XCD_SWAP ? blockIdx.y : blockIdx.x, nchunk = XCD_SWAP ? gridDim.y : gridDim.x;
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
template <bool XCD_SWAP>
__global__ void __launch_bounds__(1024) k_scatter(const uint16_t *__restrict__ D, u64 n, msm_geom g, u64 chunk, const u32 *__restrict__ starts,
                                                  const u32 *__restrict__ base, u32 *__restrict__ sorted) {
    extern __shared__ u32 cursor[];
    const int k = XCD_SWAP ? blockIdx.x : blockIdx.y, j = XCD_SWAP ? blockIdx.y : blockIdx.x, nchunk = XCD_SWAP ? gridDim.y : gridDim.x;
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
constexpr int PART_BPS = 256;            // buckets per slice
constexpr int PART_CHUNK = 16384;        // terms per pass-1 block (64 KB of staging: two blocks per CU)
constexpr int PART_CAP = 18432;          // bin capacity of the LDS path of pass 2 (mean 16384 at n = 2^21; larger bins take the global path)

__device__ __forceinline__ bool part_entry(u32 v, int k, const msm_geom &g, u32 t, u32 &slice, u32 &entry) {
    int d = digit_of(v, k, g);
    if (d == 0) return false;
    u32 b = (u32)((d > 0 ? d : -d) - 1);
    slice = b / PART_BPS;
    entry = ((b % PART_BPS) << 24) | (d < 0 ? (1u << 23) : 0u) | t;
    return true;
}
// cc[(k*SL + s)*nchunk + j] = number of non-zero digits of chunk j, window k, that fall into slice s
__global__ void __launch_bounds__(256) k_part_hist(const uint16_t *__restrict__ D, u64 n, msm_geom g, int SL, u32 *__restrict__ cc) {
    extern __shared__ u32 sm[];                               // [4][SL]
    const int k = blockIdx.x, j = blockIdx.y, nchunk = gridDim.y, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * SL; i += 256) sm[i] = 0;
    __syncthreads();
    const u64 lo = (u64)j * PART_CHUNK, hi = lo + PART_CHUNK < n ? lo + PART_CHUNK : n;
    const uint16_t *Dk = D + (u64)k * n;
    if ((((u64)k * n) & 7) == 0 && hi - lo == PART_CHUNK) {           // full, 16-byte aligned chunk: eight digits per load
        const uint4 *q = reinterpret_cast<const uint4 *>(Dk + lo);
        for (int i = threadIdx.x; i < PART_CHUNK / 8; i += 256) {
            uint4 v = q[i];
            u32 x[4] = {v.x, v.y, v.z, v.w};
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
__global__ void __launch_bounds__(1024) k_part_scan(u32 *__restrict__ cc, int SL, int nchunk, msm_geom g, u32 *__restrict__ bin_base, u32 *__restrict__ base) {
    __shared__ u32 part[1024];
    const int k = blockIdx.x, tid = threadIdx.x, M = SL * nchunk;
    u32 *v = cc + (u64)k * M;
    const int per = (M + 1023) / 1024, i0 = tid * per, i1 = i0 + per < M ? i0 + per : M;
    u32 sum = 0;
    for (int i = i0; i < i1; i++) sum += v[i];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        u32 x = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    u32 run = part[tid] - sum;
    for (int i = i0; i < i1; i++) {
        u32 c = v[i]; v[i] = run;
        if (i % nchunk == 0) bin_base[(u64)k * (SL + 1) + i / nchunk] = run;
        run += c;
    }
    if (tid == 1023) { bin_base[(u64)k * (SL + 1) + SL] = part[1023]; base[(u64)k * (g.half + 1) + g.half] = part[1023]; }
}
// pass 1: chunk j of window k -> runs per slice in P1[k][..]
__global__ void __launch_bounds__(1024, 8) k_part1(const uint16_t *__restrict__ D, u64 n, msm_geom g, int SL, const u32 *__restrict__ gofs, u32 *__restrict__ P1) {
    extern __shared__ u32 sm[];
    u32 *cnt = sm;                         // [16][SL]: per-wave counts, then per-wave cursors
    u32 *ls = sm + 16 * SL;                // [SL + 1]: start of each slice in the staging buffer
    u32 *stot = ls + SL + 1;               // [SL]
    u32 *stage = stot + SL;                // [PART_CHUNK]
    const int k = blockIdx.x, j = blockIdx.y, nchunk = gridDim.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * SL; i += 1024) cnt[i] = 0;
    __syncthreads();
    const u64 lo = (u64)j * PART_CHUNK, hi = lo + PART_CHUNK < n ? lo + PART_CHUNK : n;
    for (u64 t = lo + threadIdx.x; t < hi; t += 1024) {
        u32 sl, e;
        if (part_entry(D[(u64)k * n + t], k, g, (u32)t, sl, e)) atomicAdd(&cnt[w * SL + sl], 1u);
    }
    __syncthreads();
    for (int sidx = threadIdx.x; sidx < SL; sidx += 1024) {          // per slice: exclusive prefix over the 16 waves
        u32 run = 0;
        for (int ww = 0; ww < 16; ww++) { u32 c = cnt[ww * SL + sidx]; cnt[ww * SL + sidx] = run; run += c; }
        stot[sidx] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) { u32 run = 0; for (int sidx = 0; sidx < SL; sidx++) { ls[sidx] = run; run += stot[sidx]; } ls[SL] = run; }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * SL; i += 1024) cnt[i] += ls[i % SL];
    __syncthreads();
    for (u64 t = lo + threadIdx.x; t < hi; t += 1024) {
        u32 sl, e;
        if (part_entry(D[(u64)k * n + t], k, g, (u32)t, sl, e)) stage[atomicAdd(&cnt[w * SL + sl], 1u)] = e;
    }
    __syncthreads();
    for (int sidx = w; sidx < SL; sidx += 16) {                       // each wave copies whole runs
        const u32 len = stot[sidx], src = ls[sidx];
        u32 *dst = P1 + (u64)k * n + gofs[((u64)k * SL + sidx) * nchunk + j];
        for (u32 i = lane; i < len; i += 64) dst[i] = stage[src + i];
    }
}
// pass 2: bin (window k, slice s) -> final order, bucket totals and bucket offsets.  The bin's entries live in
// registers (18 per thread), LDS holds only the sorted copy: 74 KB per block, two blocks per CU.
constexpr int PART_R = PART_CAP / 1024;
__global__ void __launch_bounds__(1024, 8) k_part2(const u32 *__restrict__ P1, u64 n, msm_geom g, int SL, const u32 *__restrict__ bin_base,
                                                u32 *__restrict__ totals, u32 *__restrict__ base, u32 *__restrict__ sorted) {
    extern __shared__ u32 sm[];
    u32 *cnt = sm, *cur = sm + PART_BPS, *out = sm + 2 * PART_BPS;
    const int k = blockIdx.x, sidx = blockIdx.y, tid = threadIdx.x;
    const u32 b0 = bin_base[(u64)k * (SL + 1) + sidx], m = bin_base[(u64)k * (SL + 1) + sidx + 1] - b0;
    const u32 *src = P1 + (u64)k * n + b0;
    u32 *dst = sorted + (u64)k * n + b0;
    const bool fits = m <= (u32)PART_CAP;
    if (tid < PART_BPS) cnt[tid] = 0;
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
    if (tid < 64) {                                                    // exclusive scan of the 256 bucket counts by one wave
        u32 c0 = cnt[4 * tid], c1 = cnt[4 * tid + 1], c2 = cnt[4 * tid + 2], c3 = cnt[4 * tid + 3];
        u32 sum = c0 + c1 + c2 + c3, inc = sum;
        for (int off = 1; off < 64; off <<= 1) { u32 x = __shfl_up(inc, off, 64); if (tid >= off) inc += x; }
        u32 run = inc - sum;
        cur[4 * tid] = run; cur[4 * tid + 1] = run + c0; cur[4 * tid + 2] = run + c0 + c1; cur[4 * tid + 3] = run + c0 + c1 + c2;
    }
    __syncthreads();
    if (tid < PART_BPS) {
        const u64 b = (u64)sidx * PART_BPS + tid;
        totals[(u64)k * g.half + b] = cnt[tid];
        base[(u64)k * (g.half + 1) + b] = b0 + cur[tid];
    }
    __syncthreads();
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

// Scatter in bucket-range slices.  A window's sorted list is 4n bytes (8 MB at n = 2^21) and every 128-byte line of it
// collects its 32 entries from 32 different chunk blocks over the whole kernel: written in one sweep, the lines leave
// the 4 MB L2 of the XCD half-filled and every 4-byte store reaches HBM as its own 32-byte sector (measured WRITE_SIZE
// 1.1 GB for 134 MB of payload).  Here a block keeps its chunk's digits in LDS (2 bytes x 65536) and sweeps them
// `parts` times, each time scattering only the buckets of one slice: the 32 chunk blocks of a window run on the same
// XCD at the same time (blockIdx.x = window, linear workgroup id mod 8 = XCD) and move through the slices roughly
// together, so the region being written (4n/parts bytes) stays in that L2 until its lines are complete.
__global__ void __launch_bounds__(1024) k_scatter_sliced(const uint16_t *__restrict__ D, u64 n, msm_geom g, u64 chunk, int parts,
                                                         const u32 *__restrict__ starts, const u32 *__restrict__ base, u32 *__restrict__ sorted) {
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
__global__ void __launch_bounds__(256) k_order_hist(const u32 *__restrict__ totals, u64 nb, u32 *__restrict__ ord_hist) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < nb) { u32 c = totals[gid]; atomicAdd(&h[255u - (c > 255u ? 255u : c)], 1u); }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&ord_hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_order_scan(u32 *__restrict__ ord_hist) {   // one block: exclusive scan of 256 bins
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
__global__ void __launch_bounds__(256) k_order_scatter(const u32 *__restrict__ totals, u64 nb, u32 gid_off, u32 *__restrict__ ord_cursor, u32 *__restrict__ perm) {
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
// long list alone (skewed inputs: e.g. the +1 carry digit of every 128-bit z_i in verify_batch lands
// ~n/2 terms in ONE bucket; identical scalars do the same in every window).
constexpr u32 LONG_CAP = 192;      // > mean + 8 sigma of a balanced bucket (mean <= 96)
constexpr u32 LONG_SEG = 1024;     // entries per wave in the long path (16 per lane)
template <int PIPE>   // 0: plain loop; 1: next index prefetched; 2: next index and next point prefetched
__global__ void __launch_bounds__(256) k_accumulate(const u32 *__restrict__ pts, const u32 *__restrict__ sorted, const u32 *__restrict__ base,
                                                    const u32 *__restrict__ perm, u64 count, u64 n, msm_geom g, u32 *__restrict__ buckets) {
    u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= count) return;
    const u64 gid = perm[tid];
    int k = (int)(gid / g.half), b = (int)(gid % g.half);
    u32 lo = base[(u64)k * (g.half + 1) + b], hi = base[(u64)k * (g.half + 1) + b + 1];
    if (hi - lo > LONG_CAP) return;
    const u32 *list = sorted + (u64)k * n;
    ge_p3 acc = ge_identity();
    if (PIPE == 0) {
#pragma unroll 1
        for (u32 i = lo; i < hi; i++) {
            u32 e = list[i];
            acc = ge_madd_signed_p3(acc, pts_load(pts, e & 0x7fffffffu), (e >> 31) != 0);
        }
    } else if (PIPE == 1) {
        u32 e_next = lo < hi ? list[lo] : 0u;
#pragma unroll 1
        for (u32 i = lo; i < hi; i++) {
            u32 e = e_next;
            if (i + 1 < hi) e_next = list[i + 1];
            acc = ge_madd_signed_p3(acc, pts_load(pts, e & 0x7fffffffu), (e >> 31) != 0);
        }
    } else if (PIPE == 3) {
        // point i+1 AND index i+2 in flight during addition i: the gather of the next point never waits for its index
        uint4 q[PTS_Q];
        u32 e = 0, e1 = 0;
        if (lo < hi) { e = list[lo]; const uint4 *src = reinterpret_cast<const uint4 *>(pts) + PTS_Q * (u64)(e & 0x7fffffffu); for (int j = 0; j < PTS_Q; j++) q[j] = src[j]; }
        if (lo + 1 < hi) e1 = list[lo + 1];
#pragma unroll 1
        for (u32 i = lo; i < hi; i++) {
            const ge_aniels A = pts_from_q(q);
            const bool neg = (e >> 31) != 0;
            e = e1;
            if (i + 1 < hi) { const uint4 *src = reinterpret_cast<const uint4 *>(pts) + PTS_Q * (u64)(e & 0x7fffffffu); for (int j = 0; j < PTS_Q; j++) q[j] = src[j]; }
            if (i + 2 < hi) e1 = list[i + 2];
            acc = ge_madd_signed_p3(acc, A, neg);
        }
    } else {
        uint4 q[PTS_Q];
        u32 e = 0;
        if (lo < hi) { e = list[lo]; const uint4 *src = reinterpret_cast<const uint4 *>(pts) + PTS_Q * (u64)(e & 0x7fffffffu); for (int j = 0; j < PTS_Q; j++) q[j] = src[j]; }
#pragma unroll 1
        for (u32 i = lo; i < hi; i++) {
            const ge_aniels A = pts_from_q(q);
            const bool neg = (e >> 31) != 0;
            if (i + 1 < hi) { e = list[i + 1]; const uint4 *src = reinterpret_cast<const uint4 *>(pts) + PTS_Q * (u64)(e & 0x7fffffffu); for (int j = 0; j < PTS_Q; j++) q[j] = src[j]; }
            acc = ge_madd_signed_p3(acc, A, neg);
        }
    }
    p40_store(buckets, gid, acc);
}

// ---- long buckets -------------------------------------------------------------------------------------
// work list: one item per (long bucket, segment of LONG_SEG entries); item = {gid, lo, hi, slot}
struct long_item { u32 gid, lo, hi, first; };
__global__ void __launch_bounds__(256) k_find_long(const u32 *__restrict__ base, msm_geom g, u64 gid_off, u64 count, u32 max_items, long_item *__restrict__ items,
                                                   u32 *__restrict__ counters /* [0]=#items [1]=#long buckets */, u32 *__restrict__ long_gids,
                                                   u32 *__restrict__ long_first) {
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= count) return;
    gid += gid_off;
    int k = (int)(gid / g.half), b = (int)(gid % g.half);
    u32 lo = base[(u64)k * (g.half + 1) + b], hi = base[(u64)k * (g.half + 1) + b + 1];
    u32 cnt = hi - lo;
    if (cnt <= LONG_CAP) return;
    u32 nseg = (cnt + LONG_SEG - 1) / LONG_SEG;
    u32 first = atomicAdd(&counters[0], nseg);
    u32 lb = atomicAdd(&counters[1], 1u);
    long_gids[lb] = (u32)gid;
    long_first[lb] = first;
    // number of segments of this bucket is recomputed by the combiner from base[]
    for (u32 s = 0; s < nseg && first + s < max_items; s++) {
        long_item it; it.gid = (u32)gid; it.lo = lo + s * LONG_SEG; it.hi = (it.lo + LONG_SEG < hi) ? it.lo + LONG_SEG : hi; it.first = first;
        items[first + s] = it;
    }
}
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
                                                     const u32 *__restrict__ seg_sums, u32 *__restrict__ buckets) {
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
        if (threadIdx.x == 0) p40_store(buckets, gid, acc);
    }
}

// ================================================================================================
// bucket reduction: one level of the 8-ary running-sum hierarchy   [pippenger.rs:146-151 per segment]
//   S_out[seg] = sum_j S_in[seg*L + j]
//   P_out[seg] = 2^shift * sum_j j * S_in[seg*L + j]  +  sum_j P_in[seg*L + j]
// ================================================================================================
__global__ void __launch_bounds__(128) k_reduce_level(const u32 *__restrict__ S_in, const u32 *__restrict__ P_in, int m_in, int L, int shift,
                                                      int nwin, u32 *__restrict__ S_out, u32 *__restrict__ P_out) {
    int m_out = m_in / L;
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (u64)nwin * m_out) return;
    int k = (int)(gid / m_out), seg = (int)(gid % m_out);
    u64 in0 = (u64)k * m_in + (u64)seg * L;
    ge_p3 run = p40_load(S_in, in0 + L - 1);
    ge_p3 acc = run;                      // weight of entry L-1 so far: 1
#pragma unroll 1
    for (int j = L - 2; j >= 1; j--) {
        run = ge_add(run, p40_load(S_in, in0 + j));
        acc = ge_add(acc, run);
    }
    if (L > 1) {
        run = ge_add(run, p40_load(S_in, in0));
    } else {
        acc = ge_identity();              // L == 1: weight 0
    }
    if (shift > 0) acc = ge_mul_by_pow_2(acc, shift);
    if (P_in) {
#pragma unroll 1
        for (int j = 0; j < L; j++) acc = ge_add(acc, p40_load(P_in, in0 + j));
    }
    p40_store(S_out, gid, run);
    p40_store(P_out, gid, acc);
}

// The same level computed by EIGHT lanes per segment (lane j holds entry j): a 3-step suffix scan gives the running
// sums, two 3-step butterflies give sum_j j*S_j and sum_j P_j -- 10 dependent additions instead of 22.  The upper
// levels have so few segments that they are pure latency (one wave per SIMD or less), so the 8x lane count is free
// there; the first level keeps k_reduce_level (its 8x lanes would be real work).
__device__ __forceinline__ ge_p3 p3_shfl_down8(const ge_p3 &a, int d, int j) {
    ge_p3 o;
    for (int i = 0; i < 10; i++) {
        o.X.v[i] = __shfl_down(a.X.v[i], d, 8); o.Y.v[i] = __shfl_down(a.Y.v[i], d, 8);
        o.Z.v[i] = __shfl_down(a.Z.v[i], d, 8); o.T.v[i] = __shfl_down(a.T.v[i], d, 8);
    }
    const bool in = j + d < 8;
    const ge_p3 id = ge_identity();
    for (int i = 0; i < 10; i++) {
        o.X.v[i] = in ? o.X.v[i] : id.X.v[i]; o.Y.v[i] = in ? o.Y.v[i] : id.Y.v[i];
        o.Z.v[i] = in ? o.Z.v[i] : id.Z.v[i]; o.T.v[i] = in ? o.T.v[i] : id.T.v[i];
    }
    return o;
}
__device__ __forceinline__ ge_p3 p3_sum8(ge_p3 a) {          // every lane of the group ends with the group total
#pragma unroll 1
    for (int d = 4; d > 0; d >>= 1) {
        ge_p3 o;
        for (int i = 0; i < 10; i++) {
            o.X.v[i] = __shfl_xor(a.X.v[i], d, 8); o.Y.v[i] = __shfl_xor(a.Y.v[i], d, 8);
            o.Z.v[i] = __shfl_xor(a.Z.v[i], d, 8); o.T.v[i] = __shfl_xor(a.T.v[i], d, 8);
        }
        a = ge_add(a, o);
    }
    return a;
}
__global__ void __launch_bounds__(128) k_reduce_level_coop(const u32 *__restrict__ S_in, const u32 *__restrict__ P_in, int m_in, int L, int shift,
                                                           int nwin, u32 *__restrict__ S_out, u32 *__restrict__ P_out) {
    const int m_out = m_in / L;
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x, gid = t >> 3;
    const int j = (int)(t & 7);
    const bool active = gid < (u64)nwin * m_out;              // whole 8-lane groups are active or not; nobody returns early
    const int k = active ? (int)(gid / m_out) : 0, seg = active ? (int)(gid % m_out) : 0;
    const u64 in0 = (u64)k * m_in + (u64)seg * L;
    const bool have = active && j < L;
    ge_p3 run = have ? p40_load(S_in, in0 + j) : ge_identity();
    ge_p3 psum = (have && P_in) ? p40_load(P_in, in0 + j) : ge_identity();
#pragma unroll 1
    for (int d = 1; d < 8; d <<= 1) run = ge_add(run, p3_shfl_down8(run, d, j));      // run_j = sum_{i >= j} S_i
    ge_p3 acc = p3_sum8(j >= 1 ? run : ge_identity());                                  // sum_{j >= 1} run_j = sum_i i * S_i
    if (shift > 0) acc = ge_mul_by_pow_2(acc, shift);
    if (P_in) acc = ge_add(acc, p3_sum8(psum));
    if (active && j == 0) { p40_store(S_out, gid, run); p40_store(P_out, gid, acc); }
}

// ================================================================================================
// verify_batch kernels
// ================================================================================================
// hram_i = SHA-512(R_i || A_i || M_i) (batch.rs:179-191): 64-byte digest out + canonical-s flag
__global__ void __launch_bounds__(256) k_hram(const uint8_t *__restrict__ msgs, const u64 *__restrict__ msg_off, const uint8_t *__restrict__ sigs,
                                              const uint8_t *__restrict__ pks, u64 n, uint8_t *__restrict__ hram, u32 *__restrict__ bad_s) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 r[8], a[8], s[8];
    load8(sigs, 2 * i, r);
    load8(sigs, 2 * i + 1, s);
    load8(pks, i, a);
    if (!sc_is_canonical(s)) atomicAdd(bad_s, 1u);     // signature.rs:89-94 check_scalar
    sha512_stream st;
    st.init();
    for (int j = 0; j < 4; j++) st.w[j] = bswap64((u64)r[2 * j] | ((u64)r[2 * j + 1] << 32));       // R || A fills the
    for (int j = 0; j < 4; j++) st.w[4 + j] = bswap64((u64)a[2 * j] | ((u64)a[2 * j + 1] << 32));   // first 64 bytes
    st.fill = 64; st.total = 64;
    const uint8_t *m = msgs + msg_off[i];
    u64 len = msg_off[i + 1] - msg_off[i];
    for (u64 j = 0; j < len; j++) st.put_byte(m[j]);
    st.finish();
    u32 w[16];
    sha512_digest_words(st.h, w);
    uint4 *q = reinterpret_cast<uint4 *>(hram) + 4 * i;
    for (int j = 0; j < 4; j++) q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}
// device z-mode.  The z_i must depend on every input bit of the batch (a per-signature or per-subtree derivation
// allows a 2^64 meet-in-the-middle forgery), so they are derived from the root of a hash tree over the batch.  The tree
// is built from the SHA-512 COMPRESSION FUNCTION on fixed-size inputs -- no length padding, hence no extra padding
// block per node -- with the node's level and the level's node count folded into the chaining value (domain separation
// and shape binding); nodes are the first 32 bytes of the output (128-bit collision resistance, the level of the z_i).
// A tree of fixed shape over a collision-resistant compression function is binding, which is all that is needed.
//   level 0: node_j = F_0(hram_16j[0..32] || s_16j || ... || hram_16j+15[0..32] || s_16j+15): 8 chained blocks per 16
//            signatures; hram_i = H(R_i || A_i || M_i) already commits to (R_i, A_i, M_i).  Throughput-bound.
//   level l: node_j = F_l(four children): ONE compression per node.  These levels are pure latency (one dependent
//            SHA-512 compression is ~30 us for a single wave), so the last ones (<= 1024 nodes) run inside one block.
__device__ __forceinline__ void ztree_iv(u64 hs[8], u32 level, u64 count) {
    sha512_init(hs);
    hs[0] ^= 0x7a5f747265650000ull | level;           // "z_tree" || level
    hs[1] ^= count;                                    // number of nodes of the level being consumed
}
__global__ void __launch_bounds__(256) k_ztree_first(const uint8_t *__restrict__ hram, const uint8_t *__restrict__ sigs, u64 n, uint8_t *__restrict__ out) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 m_out = (n + 15) / 16;
    if (j >= m_out) return;
    u64 hs[8], w[16];
    ztree_iv(hs, 0u, n);
#pragma unroll 1
    for (int blk = 0; blk < 8; blk++) {                   // 2 signatures x 64 B per block (absent = zero bytes)
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u64 c = 16 * j + 2 * blk + half;
            const u64 *h = reinterpret_cast<const u64 *>(hram) + 8 * c, *sg = reinterpret_cast<const u64 *>(sigs) + 8 * c + 4;
#pragma unroll
            for (int q = 0; q < 4; q++) { w[8 * half + q] = c < n ? bswap64(h[q]) : 0ull; w[8 * half + 4 + q] = c < n ? bswap64(sg[q]) : 0ull; }
        }
        sha512_compress(hs, w);
    }
    u64 *o = reinterpret_cast<u64 *>(out) + 4 * j;
    for (int q = 0; q < 4; q++) o[q] = hs[q];
}
// one 4-ary level: out[j] = F_level(in[4j] || in[4j+1] || in[4j+2] || in[4j+3])[0..32]
__device__ __forceinline__ void ztree_node4(const u64 *in, u64 m_in, u64 j, u32 level, u64 *out4) {
    u64 hs[8], w[16];
    ztree_iv(hs, level, m_in);
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
        const u64 c = 4 * j + ch;
#pragma unroll
        for (int q = 0; q < 4; q++) w[4 * ch + q] = c < m_in ? in[4 * c + q] : 0ull;
    }
    sha512_compress(hs, w);
    for (int q = 0; q < 4; q++) out4[q] = hs[q];
}
__global__ void __launch_bounds__(256) k_ztree(const uint8_t *__restrict__ in, u64 m_in, u32 level, uint8_t *__restrict__ out) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (m_in + 3) / 4) return;
    u64 r[4];
    ztree_node4(reinterpret_cast<const u64 *>(in), m_in, j, level, r);
    u64 *o = reinterpret_cast<u64 *>(out) + 4 * j;
    for (int q = 0; q < 4; q++) o[q] = r[q];
}
// the last levels (m_in <= 1024 nodes) in ONE block: no launch gaps between levels that hold a handful of nodes
__global__ void __launch_bounds__(256) k_ztree_tail(const uint8_t *__restrict__ in, u64 m_in, u32 level, uint8_t *__restrict__ root) {
    __shared__ u64 buf0[1024 * 4], buf1[256 * 4];
    for (u64 i = threadIdx.x; i < m_in * 4; i += 256) buf0[i] = reinterpret_cast<const u64 *>(in)[i];
    __syncthreads();
    u64 *cur = buf0, *nxt = buf1;
    u64 m = m_in;
    while (m > 1) {
        const u64 mo = (m + 3) / 4;                                  // <= 256 = blockDim
        if (threadIdx.x < mo) {
            u64 r[4];
            ztree_node4(cur, m, threadIdx.x, level, r);
            for (int q = 0; q < 4; q++) nxt[4 * threadIdx.x + q] = r[q];
        }
        __syncthreads();
        u64 *t = cur; cur = nxt; nxt = t;
        m = mo; level++;
    }
    if (threadIdx.x < 4) reinterpret_cast<u64 *>(root)[threadIdx.x] = cur[threadIdx.x];
}
// step 3: (z_4j .. z_4j+3) = the four 16-byte quarters of SHA-512(root || LE64(j)); n4 = ceil(n/4) lanes,
// z16 has room for 4*n4 entries
__global__ void __launch_bounds__(256) k_zderive(const uint8_t *__restrict__ root, u64 n4, uint8_t *__restrict__ z16) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const u64 *h = reinterpret_cast<const u64 *>(root);
    u64 hs[8], w[16];   // 40-byte message: one block
    sha512_init(hs);
    for (int q = 0; q < 4; q++) w[q] = bswap64(h[q]);
    w[4] = bswap64(i); w[5] = 0x8000000000000000ull;
    for (int q = 6; q < 15; q++) w[q] = 0;
    w[15] = 40 * 8;
    sha512_compress(hs, w);
    // z_i < 2^127: with the signed recoding s' = s + sum HALF 2^(ck) a 127-bit value never carries out of its eighth
    // 16-bit window, so the R_i terms leave windows 8..15 empty; a full 128-bit z_i would put about half of all R_i
    // into ONE bucket of window 8 (carry digit +1).  A forged batch then passes with probability 2^-127 instead of 2^-128.
    u64 *o = reinterpret_cast<u64 *>(z16) + 8 * i;
    for (int q = 0; q < 8; q++) o[q] = bswap64(hs[q]) & ((q & 1) ? 0x7fffffffffffffffull : ~0ull);
}
// scalars of the batch equation (batch.rs:213-233): msm_scalars[1+i] = z_i, [1+n+i] = z_i*h_i;
// per-block partial sums of z_i*s_i (mod l) to `partial`
__global__ void __launch_bounds__(256) k_batch_scalars(const uint8_t *__restrict__ hram, const uint8_t *__restrict__ sigs, const uint8_t *__restrict__ z16,
                                                       u64 n, uint8_t *__restrict__ msm_scalars, u64 *__restrict__ partial) {
    __shared__ u64 red[256][5];
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    sc52 zs = sc_zero();
    if (i < n) {
        const u32 *hw = reinterpret_cast<const u32 *>(hram) + 16 * i;
        u32 h16[16];
        for (int j = 0; j < 16; j++) h16[j] = hw[j];
        const u32 *zw = reinterpret_cast<const u32 *>(z16) + 4 * i;
        u32 zwords[8] = {zw[0], zw[1], zw[2], zw[3], 0, 0, 0, 0};
        u32 s[8];
        load8(sigs, 2 * i + 1, s);
        sc52 z = sc_from_words(zwords), h = sc_from_wide(h16), sv = sc_from_words(s);
        zs = sc_mul(z, sv);
        u32 out[8];
        sc_to_words(sc_mul(h, z), out);
        store8(msm_scalars, 1 + n + i, out);
        store8(msm_scalars, 1 + i, zwords);
    }
    for (int j = 0; j < 5; j++) red[threadIdx.x][j] = zs.v[j];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sc52 a, b;
            for (int j = 0; j < 5; j++) { a.v[j] = red[threadIdx.x][j]; b.v[j] = red[threadIdx.x + off][j]; }
            a = sc_add(a, b);
            for (int j = 0; j < 5; j++) red[threadIdx.x][j] = a.v[j];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) for (int j = 0; j < 5; j++) partial[(u64)blockIdx.x * 5 + j] = red[0][j];
}

// msm_scalars[0] = -(sum of the per-block partial sums) mod l: one block, strided sums then a tree
__global__ void __launch_bounds__(256) k_bsum_finish(const u64 *__restrict__ partial, u32 nblk, uint8_t *__restrict__ msm_scalars) {
    __shared__ u64 red[256][5];
    sc52 acc = sc_zero();
    for (u32 b = threadIdx.x; b < nblk; b += 256) {
        sc52 p;
        for (int j = 0; j < 5; j++) p.v[j] = partial[(u64)b * 5 + j];
        acc = sc_add(acc, p);
    }
    for (int j = 0; j < 5; j++) red[threadIdx.x][j] = acc.v[j];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sc52 a, b;
            for (int j = 0; j < 5; j++) { a.v[j] = red[threadIdx.x][j]; b.v[j] = red[threadIdx.x + off][j]; }
            a = sc_add(a, b);
            for (int j = 0; j < 5; j++) red[threadIdx.x][j] = a.v[j];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sc52 t;
        for (int j = 0; j < 5; j++) t.v[j] = red[0][j];
        u32 w[8];
        sc_to_words(sc_neg(t), w);
        store8(msm_scalars, 0, w);
    }
}

hipError_t launch_hram(const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks, uint64_t n, uint8_t *hram, uint32_t *bad_s, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hram, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, msgs, msg_off, sigs, pks, n, hram, bad_s);
    return hipGetLastError();
}

}  // namespace c25519

// ================================================================================================
// host orchestration
// ================================================================================================
static inline unsigned div_up64(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

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
        feW t;
        for (int i = 0; i < 5; i++) { uint64_t v; memcpy(&v, in + 40 * c + 8 * i, 8); t.v[2 * i] = (u32)v & M26; t.v[2 * i + 1] = (u32)(v >> 26); }
        *f[c] = fe_carry(t);
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

// window layout for n terms (see msm_geom): signed windows share 253 - (c-1) bits evenly, then the unsigned (c-1)-bit
// window, then bits 253..255
static void msm_layout(uint64_t n, msm_geom &g) {
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
    for (int k = g.nwin; k < MSM_MAX_WIN; k++) { g.pos[k] = 0; g.wid[k] = 1; }
    for (int i = 0; i < 8; i++) g.addk[i] = a[i];
}
// diagnostics (host only, no GPU needed): the layout msm_core would use for n terms
EXPORT int32_t c25519_msm_geometry(uint64_t n, int32_t *c, int32_t *nwin, uint8_t *pos, uint8_t *wid, uint32_t *addk) {
    msm_geom g;
    msm_layout(n, g);
    *c = g.c; *nwin = g.nwin;
    for (int k = 0; k < g.nwin; k++) { pos[k] = g.pos[k]; wid[k] = g.wid[k]; }
    for (int i = 0; i < 8; i++) addk[i] = g.addk[i];
    return C25519_OK;
}

// Sum over `nterms` (scalars at d_scalars, packed affine Niels points at d_pts) -> R.
// sort_stream: stream on which the scalars become ready and on which the digit/sort kernels are enqueued
// (nullptr = the context's main stream).  Sorting depends only on the scalars, so a caller can run it on the
// second stream while the main stream still prepares the points; msm_core joins the two itself.
int32_t msm_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, ge_p3 &R, hipEvent_t *ring, hipStream_t sort_stream,
                 void *extra_dst, const void *extra_src, size_t extra_bytes) {
    msm_geom g;
    msm_layout(n, g);
    int nchunk = std::max(1, std::min(64, 512 / g.nwin));
    while (nchunk > 1 && n / nchunk < 4096) nchunk /= 2;
    if ((n + nchunk - 1) / nchunk > 65536) nchunk = (int)((n + 65535) / 65536);   // a chunk's digits must fit LDS (k_scatter_sliced)
    uint64_t chunk = (n + nchunk - 1) / nchunk;
    const uint64_t nb = (uint64_t)g.nwin * g.half;
    // level plan for the reduction
    struct lvl { int m_in, L, shift; };
    std::vector<lvl> plan;
    { int m = g.half, sh = 0; while (m > 1) { int L = m >= 8 ? 8 : m; plan.push_back({m, L, sh}); int l2 = 0; while ((1 << l2) < L) l2++; sh += l2; m /= L; } }
    // workspace carve-up (tmp_d): D | counts | base | sorted | buckets | S/P ping-pong | flags
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t oD = carve((size_t)g.nwin * n * 2), oC = carve((size_t)g.nwin * nchunk * g.half * 4), oB = carve((size_t)g.nwin * (g.half + 1) * 4);
    size_t oS = carve((size_t)g.nwin * n * 4), oK = carve(nb * 160), oT = carve(nb * 4);
    size_t lvl_pts = plan.empty() ? (size_t)g.nwin : (size_t)g.nwin * (plan[0].m_in / plan[0].L);
    size_t oR0 = carve(lvl_pts * 160 * 2), oR1 = carve(lvl_pts * 160 * 2), oF = carve(256 + 2048), oPerm = carve(nb * 4);
    // long-bucket path: at most (#entries / LONG_SEG + #long buckets) work items; a long bucket has > LONG_CAP entries
    const uint64_t entries = (uint64_t)g.nwin * n;
    const uint32_t max_long = (uint32_t)std::min<uint64_t>(nb, entries / LONG_CAP + 1);
    const uint32_t max_items = (uint32_t)(entries / LONG_SEG + max_long + 1);
    size_t oLI = carve((size_t)max_items * sizeof(long_item)), oLG = carve((size_t)max_long * 4), oLF = carve((size_t)max_long * 4);
    size_t oLS = carve((size_t)max_items * 160);
    // two-pass partition sort (see k_part1): pass-1 output, coarse counts / offsets, bin bases
    static const int sort2 = [] { const char *e = getenv("C25519_SORT2"); return e ? atoi(e) : 1; }();
    const bool use_part = sort2 && g.c >= 13 && n <= (1ull << 23) && n >= (1ull << 16);
    const int SL = g.half / PART_BPS, pchunks = (int)((n + PART_CHUNK - 1) / PART_CHUNK);
    size_t oP1 = 0, oCC = 0, oBB = 0;
    if (use_part) { oP1 = carve((size_t)g.nwin * n * 4); oCC = carve((size_t)g.nwin * SL * pchunks * 4); oBB = carve((size_t)g.nwin * (SL + 1) * 4); }
    int32_t r = ctx_reserve(ctx, ctx->tmp_d, off);
    if (r) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_d.p;
    uint16_t *D = (uint16_t *)(ws + oD);
    uint32_t *counts = (uint32_t *)(ws + oC), *base = (uint32_t *)(ws + oB), *sorted = (uint32_t *)(ws + oS), *buckets = (uint32_t *)(ws + oK);
    uint32_t *flags = (uint32_t *)(ws + oF), *totals = (uint32_t *)(ws + oT), *ord_hist = flags + 64, *perm = (uint32_t *)(ws + oPerm);
    static const bool overlap = [] { const char *e = getenv("C25519_SORT_OVERLAP"); return e && atoi(e) != 0; }();   // measured: no gain (MSM 3.15 -> 3.35 ms, verify neutral) -- every kernel already fills the chip
    if (!overlap && sort_stream && sort_stream != ctx->stream) {    // A/B knob: serialise (main waits for the scalars, sorts itself)
        HIPCHK(hipEventRecord(ctx->ev_sort, sort_stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_sort, 0));
        sort_stream = nullptr;
    }
    hipStream_t st = sort_stream ? sort_stream : ctx->stream;      // sort phase
    HIPCHK(hipMemsetAsync(flags, 0, 256 + 2048, st));
    hipLaunchKernelGGL(k_digits, dim3(div_up64(n, 256)), dim3(256), 0, st, d_scalars, n, g, D, flags);
    if (use_part) {
        uint32_t *P1 = (uint32_t *)(ws + oP1), *cc = (uint32_t *)(ws + oCC), *bin_base = (uint32_t *)(ws + oBB);
        const size_t lds1 = ((size_t)16 * SL + 2 * SL + 1 + PART_CHUNK) * 4, lds2 = ((size_t)2 * PART_BPS + PART_CAP) * 4;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(k_part_hist, dim3(g.nwin, pchunks), dim3(256), (size_t)4 * SL * 4, st, D, n, g, SL, cc);
        hipLaunchKernelGGL(k_part_scan, dim3(g.nwin), dim3(1024), 0, st, cc, SL, pchunks, g, bin_base, base);
        hipLaunchKernelGGL(k_part1, dim3(g.nwin, pchunks), dim3(1024), lds1, st, D, n, g, SL, cc, P1);
        hipLaunchKernelGGL(k_part2, dim3(g.nwin, SL), dim3(1024), lds2, st, P1, n, g, SL, bin_base, totals, base, sorted);
    } else {
    size_t lds = (size_t)g.half * 4;
    static int xswap = -1;   // tuning knob: C25519_XCD_SWAP = 0 | 1
    if (xswap < 0) { const char *e = getenv("C25519_XCD_SWAP"); xswap = e ? atoi(e) : 1; }
    if (lds > 48 * 1024) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hist<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hist<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (xswap) hipLaunchKernelGGL(k_hist<true>, dim3(g.nwin, nchunk), dim3(1024), lds, st, D, n, g, chunk, counts);
    else hipLaunchKernelGGL(k_hist<false>, dim3(nchunk, g.nwin), dim3(1024), lds, st, D, n, g, chunk, counts);
    hipLaunchKernelGGL(k_scan_chunks, dim3(div_up64(nb, 256)), dim3(256), 0, st, counts, nchunk, g, totals);
    hipLaunchKernelGGL(k_scan_buckets, dim3(g.nwin), dim3(1024), 0, st, totals, g, base);
    static const int sparts = [] { const char *e = getenv("C25519_SCATTER_PARTS"); int v = e ? atoi(e) : 8; return (v == 2 || v == 4 || v == 8 || v == 16) ? v : 1; }();
    const size_t lds_sliced = (size_t)g.half / sparts * 4 + (size_t)chunk * 2;
    if (sparts > 1 && g.half >= 1024 * sparts && lds_sliced <= 160 * 1024) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_sliced), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sliced));
        hipLaunchKernelGGL(k_scatter_sliced, dim3(g.nwin, nchunk), dim3(1024), lds_sliced, st, D, n, g, chunk, sparts, counts, base, sorted);
    } else if (xswap) hipLaunchKernelGGL(k_scatter<true>, dim3(g.nwin, nchunk), dim3(1024), lds, st, D, n, g, chunk, counts, base, sorted);
    else hipLaunchKernelGGL(k_scatter<false>, dim3(nchunk, g.nwin), dim3(1024), lds, st, D, n, g, chunk, counts, base, sorted);
    }
    if (sort_stream && sort_stream != ctx->stream) {                // join: the main stream continues once the lists exist
        HIPCHK(hipEventRecord(ctx->ev_sort, sort_stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_sort, 0));
    }
    st = ctx->stream;
    // Window groups: while group g+1 accumulates (VALU-bound, the whole chip), the latency-bound reduction
    // levels of group g (a few thousand lanes) run on the second stream.
    static int groups = -1;   // tuning knob: C25519_MSM_GROUPS = 1 | 2
    if (groups < 0) { const char *e = getenv("C25519_MSM_GROUPS"); groups = e ? atoi(e) : 1; if (groups != 2) groups = 1; }   // measured: 2 groups cost more (two kernel tails) than the overlap returns
    const int G = (groups == 2 && g.nwin >= 8 && n >= 65536) ? 2 : 1;
    static int pipe = -1;   // tuning knob (A/B on hardware): C25519_ACC_PIPE = 0 | 1 | 2 | 3 (LDS-DMA staging as in k_mul_base_wide was tried here: -15 %)
    if (pipe < 0) { const char *e = getenv("C25519_ACC_PIPE"); pipe = e ? atoi(e) : 3; if (pipe < 0 || pipe > 3) pipe = 3; }
    long_item *items = (long_item *)(ws + oLI);
    uint32_t *lgids = (uint32_t *)(ws + oLG), *lfirst = (uint32_t *)(ws + oLF), *segs = (uint32_t *)(ws + oLS);
    uint32_t *bufs[2] = {(uint32_t *)(ws + oR0), (uint32_t *)(ws + oR1)};
    const uint32_t *S_fin[2] = {nullptr, nullptr}, *P_fin[2] = {nullptr, nullptr};
    int k_lo[3] = {0, G == 2 ? g.nwin / 2 : g.nwin, g.nwin};
    if (ring) HIPCHK(hipEventRecord(ring[0], st));
    for (int grp = 0; grp < G; grp++) {
        const int k0 = k_lo[grp], k1 = k_lo[grp + 1], nw = k1 - k0;
        const uint64_t goff = (uint64_t)k0 * g.half, cnt = (uint64_t)nw * g.half;
        uint32_t *oh = ord_hist + 256 * grp, *pg = perm + goff, *counters = flags + 8 + 4 * grp;
        hipLaunchKernelGGL(k_order_hist, dim3(div_up64(cnt, 256)), dim3(256), 0, st, totals + goff, cnt, oh);
        hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(256), 0, st, oh);
        hipLaunchKernelGGL(k_order_scatter, dim3(div_up64(cnt, 256)), dim3(256), 0, st, totals + goff, cnt, (uint32_t)goff, oh, pg);
        // long buckets are independent of k_accumulate (which skips them): fold them on the second stream meanwhile
        HIPCHK(hipEventRecord(ctx->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
        hipLaunchKernelGGL(k_find_long, dim3(div_up64(cnt, 256)), dim3(256), 0, ctx->aux, base, g, goff, cnt, max_items, items, counters, lgids, lfirst);
        hipLaunchKernelGGL(k_long_segments, dim3(std::min<uint32_t>(max_items, 2048u)), dim3(64), 0, ctx->aux, d_pts, sorted, n, g, items, counters, max_items, segs);
        hipLaunchKernelGGL(k_long_combine, dim3(std::min<uint32_t>(max_long, 1024u)), dim3(64), 0, ctx->aux, base, g, counters, max_items, lgids, lfirst, segs, buckets);
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->aux));
        if (pipe == 0) hipLaunchKernelGGL(k_accumulate<0>, dim3(div_up64(cnt, 256)), dim3(256), 0, st, d_pts, sorted, base, pg, cnt, n, g, buckets);
        else if (pipe == 1) hipLaunchKernelGGL(k_accumulate<1>, dim3(div_up64(cnt, 256)), dim3(256), 0, st, d_pts, sorted, base, pg, cnt, n, g, buckets);
        else if (pipe == 3) hipLaunchKernelGGL(k_accumulate<3>, dim3(div_up64(cnt, 256)), dim3(256), 0, st, d_pts, sorted, base, pg, cnt, n, g, buckets);
        else hipLaunchKernelGGL(k_accumulate<2>, dim3(div_up64(cnt, 256)), dim3(256), 0, st, d_pts, sorted, base, pg, cnt, n, g, buckets);
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));     // long-bucket path (aux stream) done
        if (grp == G - 1 && ring) HIPCHK(hipEventRecord(ring[1], st));
        // reduction levels of this group: on the aux stream unless it is the last group
        hipStream_t rs = st;
        if (grp < G - 1) {
            rs = ctx->aux;
            HIPCHK(hipEventRecord(ctx->ev_fork2, st));
            HIPCHK(hipStreamWaitEvent(rs, ctx->ev_fork2, 0));
        }
        const uint32_t *S_in = buckets + goff * 40, *P_in = nullptr;
        const size_t gl = (size_t)nw * (plan.empty() ? 1 : plan[0].m_in / plan[0].L);       // points per level buffer of this group
        uint32_t *gb[2] = {bufs[0] + (size_t)k0 * (lvl_pts / g.nwin) * 80, bufs[1] + (size_t)k0 * (lvl_pts / g.nwin) * 80};
        int which = 0;
        for (size_t li = 0; li < plan.size(); li++) {
            int m_out = plan[li].m_in / plan[li].L;
            uint32_t *S_out = gb[which], *P_out = gb[which] + gl * 40;
            static const int coop = [] { const char *e = getenv("C25519_REDUCE_COOP"); return e ? atoi(e) : 1; }();
            if (coop && (uint64_t)nw * m_out * 8 <= (1u << 17))     // <= 2 waves per SIMD even with 8 lanes per segment: latency-bound
                hipLaunchKernelGGL(k_reduce_level_coop, dim3(div_up64((uint64_t)nw * m_out * 8, 128)), dim3(128), 0, rs, S_in, P_in, plan[li].m_in,
                                   plan[li].L, plan[li].shift, nw, S_out, P_out);
            else
                hipLaunchKernelGGL(k_reduce_level, dim3(div_up64((uint64_t)nw * m_out, 128)), dim3(128), 0, rs, S_in, P_in, plan[li].m_in, plan[li].L,
                                   plan[li].shift, nw, S_out, P_out);
            S_in = S_out; P_in = P_out; which ^= 1;
        }
        S_fin[grp] = S_in; P_fin[grp] = P_in;
        if (grp < G - 1) HIPCHK(hipEventRecord(ctx->ev_join2, rs));
    }
    HIPCHK(hipGetLastError());
    if (G == 2) HIPCHK(hipStreamWaitEvent(st, ctx->ev_join2, 0));
    // window totals -> host, Horner fold (pippenger.rs:159)
    // (pinned staging: three small copies queue back to back instead of three staged pageable copies)
    static_assert((size_t)MSM_MAX_WIN * 160 * 2 + 64 <= 20 * 1024, "h_msm too small");
    uint32_t *hS = (uint32_t *)ctx->h_msm, *hP = hS + (size_t)MSM_MAX_WIN * 40, *hflags = hP + (size_t)MSM_MAX_WIN * 40;
    hflags[0] = hflags[1] = 0;
    const bool haveP = !plan.empty();
    for (int grp = 0; grp < G; grp++) {
        const int k0 = k_lo[grp], nw = k_lo[grp + 1] - k0;
        HIPCHK(hipMemcpyAsync(hS + (size_t)k0 * 40, S_fin[grp], (size_t)nw * 160, hipMemcpyDeviceToHost, st));
        if (haveP) HIPCHK(hipMemcpyAsync(hP + (size_t)k0 * 40, P_fin[grp], (size_t)nw * 160, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipMemcpyAsync(hflags, flags, 8, hipMemcpyDeviceToHost, st));
    if (extra_bytes) HIPCHK(hipMemcpyAsync(extra_dst, extra_src, extra_bytes, hipMemcpyDeviceToHost, st));
    if (ring) HIPCHK(hipEventRecord(ring[2], st));
    HIPCHK(hipStreamSynchronize(st));
    if (hflags[0]) { ctx->err = "msm: a scalar has bit 255 set (Scalar invariant #1 violated)"; return -(int32_t)hipErrorInvalidValue; }
    ge_p3 total = ge_identity();
    for (int k = g.nwin - 1; k >= 0; k--) {
        ge_p3 col = host_p40(&hS[(size_t)k * 40]);                    // sum_b B_b
        if (haveP) col = ge_add(col, host_p40(&hP[(size_t)k * 40]));  // + sum_b b*B_b
        if (k != g.nwin - 1) total = ge_mul_by_pow_2(total, g.pos[k + 1] - g.pos[k]);
        total = ge_add(total, col);
    }
    R = total;
    return C25519_OK;
}

// points in any format -> packed affine Niels at d_pts[dst0..]; returns C25519_NONE if some point is invalid
int32_t prep_points(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_pts, uint64_t dst0, uint32_t *d_badcount) {
    hipStream_t st = ctx->stream;
    if (n == 0) return C25519_OK;
    if (in_fmt == C25519_FMT_EDWARDS_Y) HIPCHK(launch_prep_compressed(0, d_points, 1, n, d_pts, dst0, d_badcount, st));
    else if (in_fmt == C25519_FMT_RISTRETTO) HIPCHK(launch_prep_compressed(1, d_points, 1, n, d_pts, dst0, d_badcount, st));
    else if (in_fmt == C25519_FMT_RAW160) {
        int32_t r = ctx_reserve(ctx, ctx->prefix, n * 48);
        if (r) return r;
        constexpr int CH = 16;
        hipLaunchKernelGGL(k_prep_raw<CH>, dim3(div_up64((n + CH - 1) / CH, 256)), dim3(256), 0, st, d_points, n, (uint32_t *)ctx->prefix.p, d_pts, dst0);
    } else { ctx->err = "msm: bad in_fmt"; return -(int32_t)hipErrorInvalidValue; }
    HIPCHK(hipGetLastError());
    return C25519_OK;
}

// One bucket-method pass over at most MSM_PASS_MAX terms.
static int32_t msm_partial_pass(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, ge_p3 &R) {
    R = ge_identity();
    int32_t r = ctx_reserve(ctx, ctx->tmp_e, n * PTS_BYTES + 256);
    if (r) return r;
    uint32_t *d_pts = (uint32_t *)ctx->tmp_e.p;
    uint32_t *d_bad = (uint32_t *)ctx->d_flag;
    hipEvent_t *ring = ctx->ring[ctx->ncalls++ % c25519_ctx::RING];
    HIPCHK(hipMemsetAsync(d_bad, 0, 16, ctx->stream));
    // points are normalised on the main stream while the scalars are recoded and sorted on the second one
    HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    if ((r = prep_points(ctx, d_points, n, in_fmt, d_pts, 0, d_bad))) return r;
    r = msm_core(ctx, d_scalars, n, d_pts, R, ring, ctx->aux);
    if (r != C25519_OK) return r;
    uint32_t bad = 0;                                  // msm_core has synchronised: prep's counter is final
    HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return bad ? C25519_NONE : C25519_OK;
}
// The window width stops at c = 16 (the per-window histogram lives in LDS), so beyond ~2^22 terms the lists per
// bucket only get longer: larger inputs are cut into passes of about 2^21 terms -- the same decomposition the
// multi-GPU path uses across ranks -- whose partial sums are added on the host.  This also bounds
// the workspace (~0.5 GB) for any n.
static const int MSM_PASS_LOG2 = [] { const char *e = getenv("C25519_MSM_PASS_LOG2"); int v = e ? atoi(e) : 21; return v < 16 ? 16 : (v > 21 ? 21 : v); }();   // A/B knob
static const uint64_t MSM_PASS = 1ull << MSM_PASS_LOG2, MSM_PASS_MAX = 3ull << (MSM_PASS_LOG2 - 1);
static int pass_lanes() { static const int v = [] { const char *e = getenv("C25519_PASS_LANES"); int x = e ? atoi(e) : 2; return x < 1 ? 1 : (x > 4 ? 4 : x); }(); return v; }   // A/B knob
// run(c, first, step) on `lanes` contexts: the caller's and up to three peers (each the peer of the previous one)
template <class F>
static void run_on_lanes(c25519_ctx *ctx, uint64_t passes, F run) {
    c25519_ctx *cs[4] = {ctx, nullptr, nullptr, nullptr};
    int lanes = 1;
    const int want = (int)std::min<uint64_t>(passes, (uint64_t)pass_lanes());
    while (lanes < want) { c25519_ctx *p = ctx_peer(cs[lanes - 1]); if (!p) break; cs[lanes++] = p; }
    if (lanes > 1) hipStreamSynchronize(ctx->stream);               // the inputs are complete before other streams read them
    std::vector<std::thread> ts;
    for (int l = 1; l < lanes; l++) ts.emplace_back(run, cs[l], (uint64_t)l, (uint64_t)lanes);
    run(ctx, 0, (uint64_t)lanes);
    for (auto &t : ts) t.join();
}
static int32_t msm_partial_impl(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, ge_p3 &R) {
    HIPCHK(hipSetDevice(ctx->device));
    R = ge_identity();
    if (n == 0) return C25519_OK;
    if (n >= (1ull << 40)) { ctx->err = "msm: n must be < 2^40"; return -(int32_t)hipErrorInvalidValue; }
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    const uint64_t passes = n <= MSM_PASS_MAX ? 1 : (n + MSM_PASS - 1) / MSM_PASS, per = (n + passes - 1) / passes;
    const size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    // Passes are independent.  With several of them they are dealt round-robin to the caller's context and its peer
    // contexts (own streams and workspaces, same device), one host thread each: two thirds of a pass is low-VALU work
    // (normalise, sort, reduce, read-back) that now overlaps the accumulation of a neighbouring pass.
    std::vector<ge_p3> part(passes, ge_identity());
    std::vector<int32_t> st(passes, C25519_OK);
    auto run = [&](c25519_ctx *c, uint64_t first, uint64_t step) {
        hipSetDevice(c->device);
        for (uint64_t i = first; i < passes; i += step) {
            const uint64_t lo = i * per, cnt = std::min(per, n - lo);
            st[i] = msm_partial_pass(c, d_scalars + lo * 32, d_points + lo * psz, cnt, in_fmt, part[i]);
            if (st[i] < 0) break;
        }
    };
    run_on_lanes(ctx, passes, run);
    bool none = false;
    for (uint64_t i = 0; i < passes; i++) {
        if (st[i] < 0) { if (ctx->err.empty()) ctx->err = "msm: a pass failed on a peer context"; return st[i]; }
        if (st[i] == C25519_NONE) none = true;          // the status must not depend on the split
        else R = passes == 1 ? part[i] : ge_add(R, part[i]);
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    return none ? C25519_NONE : C25519_OK;
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
    size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, n * 32 + 16)) || (r = ctx_reserve(ctx, ctx->tmp_b, n * psz + 16))) return r;
    if (n) {
        HIPCHK(hipMemcpyAsync(ctx->tmp_a.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->tmp_b.p, points, n * psz, hipMemcpyHostToDevice, ctx->stream));
    }
    return c25519_msm_vartime_dev(ctx, (const uint8_t *)ctx->tmp_a.p, (const uint8_t *)ctx->tmp_b.p, n, in_fmt, out_fmt, out);
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

// One random-linear-combination check over at most VERIFY_PASS_MAX signatures (an MSM of 2n+1 terms).
// d_pk_points (may be NULL): the keys' decompressed points, n x 160 raw -- what VerifyingKey carries beside its bytes
// (verifying.rs:64-71), so that, like the reference (batch.rs:236), the batch does not decompress A_i again.
static int32_t verify_batch_pass(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off,
                                 const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, uint64_t n, uint32_t z_mode) {
    hipStream_t st = ctx->stream;
    const uint64_t m = 2 * n + 1;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_e, m * PTS_BYTES + 256))) return r;
    // tmp_f: hram (64n) | z16 (16n) | msm scalars (32m) | leaf/tree (64n + 64n/16 + ..) | partial sums
    const unsigned nblk = div_up64(n, 256);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t oH = carve(n * 64), oZ = carve((n + 4) * 16), oSc = carve(m * 32), oT0 = carve(n * 64), oT1 = carve((n / 16 + 1) * 64), oP = carve((size_t)nblk * 40);
    if ((r = ctx_reserve(ctx, ctx->tmp_f, off))) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_f.p;
    uint8_t *hram = ws + oH, *z16 = ws + oZ, *msc = ws + oSc, *t0 = ws + oT0, *t1 = ws + oT1;
    uint64_t *partial = (uint64_t *)(ws + oP);
    uint32_t *d_pts = (uint32_t *)ctx->tmp_e.p;
    uint32_t *d_cnt = (uint32_t *)ctx->d_flag;      // [0] bad A, [1] bad R, [2] bad s
    hipEvent_t *ring = ctx->ring[ctx->ncalls++ % c25519_ctx::RING];
    HIPCHK(hipMemsetAsync(d_cnt, 0, 16, st));
    // Two independent chains: (S) decompress A_i and R_i -- VALU-bound, ~30 % of the call; (A) hash,
    // derive z_i, batch scalars -- partly latency-bound (the Merkle levels).  They run on two streams
    // and join before the MSM.
    hipStream_t sa = ctx->aux;
    HIPCHK(hipEventRecord(ctx->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(sa, ctx->ev_fork, 0));
    // (S) points: [0] = B, [1..n] = R_i, [n+1..2n] = A_i     (batch.rs:235-244)
    hipLaunchKernelGGL(k_prep_basepoint, dim3(1), dim3(64), 0, st, d_pts, (uint64_t)0);
    if (d_pk_points) { if ((r = prep_points(ctx, d_pk_points, n, C25519_FMT_RAW160, d_pts, n + 1, d_cnt + 0))) return r; }
    else HIPCHK(launch_prep_compressed(0, d_pks, 1, n, d_pts, n + 1, d_cnt + 0, st));
    HIPCHK(launch_prep_compressed(0, d_sigs, 2, n, d_pts, 1, d_cnt + 1, st));                 // R_i = the first half of every 64-byte signature
    // (A)
    hipLaunchKernelGGL(k_hram, dim3(nblk), dim3(256), 0, sa, d_msgs, d_msg_off, d_sigs, d_pks, n, hram, d_cnt + 2);
    HIPCHK(hipGetLastError());
    if (z_mode == C25519_Z_TRANSCRIPT) {
        // the reference's sequential Merlin transcript (batch.rs:168-222), on the host
        std::vector<uint8_t> hh(n * 64), hs(n * 64), hz(n * 16);
        HIPCHK(hipMemcpyAsync(hh.data(), hram, n * 64, hipMemcpyDeviceToHost, sa));
        HIPCHK(hipMemcpyAsync(hs.data(), d_sigs, n * 64, hipMemcpyDeviceToHost, sa));
        HIPCHK(hipStreamSynchronize(sa));
        c25519_transcript_zs(hh.data(), hs.data(), n, hz.data());
        HIPCHK(hipMemcpyAsync(z16, hz.data(), n * 16, hipMemcpyHostToDevice, sa));
        HIPCHK(hipStreamSynchronize(sa));
    } else {
        uint64_t mm = (n + 15) / 16; uint8_t *a = t0, *b = t1;
        uint32_t level = 1;
        hipLaunchKernelGGL(k_ztree_first, dim3(div_up64(mm, 256)), dim3(256), 0, sa, hram, d_sigs, n, a);
        while (mm > 1024) {
            uint64_t mo = (mm + 3) / 4;
            hipLaunchKernelGGL(k_ztree, dim3(div_up64(mo, 256)), dim3(256), 0, sa, a, mm, level, b);
            mm = mo; level++; std::swap(a, b);
        }
        hipLaunchKernelGGL(k_ztree_tail, dim3(1), dim3(256), 0, sa, a, mm, level, b);      // leaves the 32-byte root at b
        std::swap(a, b);
        hipLaunchKernelGGL(k_zderive, dim3(div_up64((n + 3) / 4, 256)), dim3(256), 0, sa, a, (n + 3) / 4, z16);
        HIPCHK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_batch_scalars, dim3(nblk), dim3(256), 0, sa, hram, d_sigs, z16, n, msc, partial);
    HIPCHK(hipGetLastError());
    // the basepoint coefficient -sum z_i s_i (batch.rs:240): the per-block partial sums are folded by one more block on
    // the device, so the host does not have to wait for chain (A) before it can enqueue the MSM
    hipLaunchKernelGGL(k_bsum_finish, dim3(1), dim3(256), 0, sa, partial, nblk, msc);
    HIPCHK(hipGetLastError());
    if (ctx->h_pinned_cap < 64) {
        if (ctx->h_pinned) HIPCHK(hipHostFree(ctx->h_pinned));
        ctx->h_pinned = nullptr; ctx->h_pinned_cap = 0;
        HIPCHK(hipHostMalloc(&ctx->h_pinned, 4096, hipHostMallocDefault));
        ctx->h_pinned_cap = 4096;
    }
    uint32_t *cnt = (uint32_t *)ctx->h_pinned;          // pinned staging of the three counters
    // the MSM's digit/sort phase continues on the second stream while (S) is still decompressing
    ge_p3 R;
    r = msm_core(ctx, msc, m, d_pts, R, ring, sa, cnt, d_cnt, 16);      // the decode / canonical-s counters ride along with the last copy
    if (r != C25519_OK) return r;
    if (cnt[0]) return C25519_NONE;                     // a key that VerifyingKey::from_bytes rejects
    if (cnt[2]) return C25519_SCALAR_FORMAT;            // batch.rs:208-211
    if (cnt[1]) return C25519_VERIFY;                   // batch.rs:244 (R fails to decompress)
    return ge_is_identity(R) ? C25519_OK : C25519_VERIFY;   // batch.rs:246-250
}
// Batches beyond ~1.5 * 2^20 signatures are checked as several independent random linear combinations of about
// 2^20 signatures each (same reason as MSM_PASS_MAX; every pass derives its own z_i).  All passes run even after
// a failure so that the reference's precedence -- key decoding, then ScalarFormat for ANY non-canonical s
// (batch.rs:208-211), then Verify -- does not depend on where the batch was cut.
static const int VERIFY_PASS_LOG2 = [] { const char *e = getenv("C25519_VERIFY_PASS_LOG2"); int v = e ? atoi(e) : 20; return v < 15 ? 15 : (v > 20 ? 20 : v); }();   // A/B knob
static const uint64_t VERIFY_PASS = 1ull << VERIFY_PASS_LOG2, VERIFY_PASS_MAX = 3ull << (VERIFY_PASS_LOG2 - 1);
EXPORT int32_t ed25519_verify_batch_keys_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                             const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, uint64_t n, uint32_t z_mode) {
    (void)msgs_len;
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;                      // batch.rs: 1-term MSM 0*B = identity
    if (n >= (1ull << 40)) { ctx->err = "verify_batch: n too large"; return -(int32_t)hipErrorInvalidValue; }
    if (z_mode > 1) { ctx->err = "verify_batch: bad z_mode"; return -(int32_t)hipErrorInvalidValue; }
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    const uint64_t passes = n <= VERIFY_PASS_MAX ? 1 : (n + VERIFY_PASS - 1) / VERIFY_PASS, per = (n + passes - 1) / passes;
    std::vector<int32_t> st(passes, C25519_OK);
    auto run = [&](c25519_ctx *c, uint64_t first, uint64_t step) {              // see msm_partial_impl
        hipSetDevice(c->device);
        for (uint64_t i = first; i < passes; i += step) {
            const uint64_t lo = i * per;
            st[i] = verify_batch_pass(c, d_msgs, d_msg_off + lo, d_sigs + lo * 64, d_pks + lo * 32, d_pk_points ? d_pk_points + lo * 160 : nullptr,
                                      std::min(per, n - lo), z_mode);
            if (st[i] < 0) break;
        }
    };
    run_on_lanes(ctx, passes, run);
    bool seen[5] = {false, false, false, false, false};
    for (uint64_t i = 0; i < passes; i++) {
        if (st[i] < 0) { if (ctx->err.empty()) ctx->err = "verify_batch: a pass failed on a peer context"; return st[i]; }
        seen[st[i]] = true;
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    return seen[C25519_NONE] ? C25519_NONE : seen[C25519_SCALAR_FORMAT] ? C25519_SCALAR_FORMAT : seen[C25519_VERIFY] ? C25519_VERIFY : C25519_OK;
}

EXPORT int32_t ed25519_verify_batch_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                        const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, uint32_t z_mode) {
    return ed25519_verify_batch_keys_dev(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, nullptr, n, z_mode);
}
EXPORT int32_t ed25519_verify_batch_keys(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks,
                                         const uint8_t *pk_points, uint64_t n, uint32_t z_mode) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;
    uint64_t mlen = msg_off[n];
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, mlen + 64)) || (r = ctx_reserve(ctx, ctx->tmp_b, (n + 1) * 8)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 64)) ||
        (r = ctx_reserve(ctx, ctx->scratch, n * 32 + (pk_points ? n * 160 : 0) + 16)))
        return r;
    if (mlen) HIPCHK(hipMemcpyAsync(ctx->tmp_a.p, msgs, mlen, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->tmp_b.p, msg_off, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->tmp_c.p, sigs, n * 64, hipMemcpyHostToDevice, ctx->stream));
    uint8_t *d_pk = (uint8_t *)ctx->scratch.p, *d_pp = pk_points ? d_pk + n * 32 : nullptr;
    HIPCHK(hipMemcpyAsync(d_pk, pks, n * 32, hipMemcpyHostToDevice, ctx->stream));
    if (pk_points) HIPCHK(hipMemcpyAsync(d_pp, pk_points, n * 160, hipMemcpyHostToDevice, ctx->stream));
    return ed25519_verify_batch_keys_dev(ctx, (const uint8_t *)ctx->tmp_a.p, (const uint64_t *)ctx->tmp_b.p, mlen, (const uint8_t *)ctx->tmp_c.p,
                                         d_pk, d_pp, n, z_mode);
}
EXPORT int32_t ed25519_verify_batch(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks,
                                    uint64_t n, uint32_t z_mode) {
    return ed25519_verify_batch_keys(ctx, msgs, msg_off, sigs, pks, nullptr, n, z_mode);
}
