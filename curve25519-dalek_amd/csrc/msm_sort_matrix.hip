// Round 2's digit-matrix sort (u16 digit matrix D[window][term], histogram / scan / scatter or the two-pass partition through LDS).  It serves
//   * the merged layout of the precomputed static tables (extra.hip c25519_precomp_*: its terms are (window, scalar) pairs with up to 2^24 ids,
//     beyond the 23-bit term index of the chunk-local sort's entries), and
//   * plain MSM / verify_batch passes of 4 096 .. 65 535 terms.  Round 4 first routed those through the chunk-local sort as well (msm_sort.hip
//     runs correctly from 2 048 terms on, tested) -- and measured why not: its partition wants hundreds of 8 192-term chunks, and below 2^16
//     terms there are 1 - 8 of them (one block walks all ~36 windows of a 4 096-term input): 0.79 ms per call at 4 096 terms against 0.59 here,
//     0.92 against 0.66 at 16 384, 1.24 against 0.92 at 65 535 (profiles/r04_msm_midrange.txt).  Below 4 096 terms: small.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <stdexcept>
#include <string>
#include <vector>
#include <functional>
#include "../../include/c25519_hip.h"
#include "devio.h"
#include "sc_sha.h"
#include "sc28.h"
#include "kernels.h"
#include "ctx.h"
#include "msm_internal.h"
#include "msm_sort.h"
#include "ffi.h"

using namespace c25519;
#define EXPORT extern "C" __attribute__((visibility("default")))
#define HIPCHK(call)                                                \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return c25519_fail(ctx, _e, #call);   \
    } while (0)


namespace c25519 {

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
                if (part_entry((x[h >> 1] >> (16 * (h & 1))) & 0xffffu, k, g, g.bps_log2, 0u, sl, e)) atomicAdd(&sm[w * SL + sl], 1u);
            }
        }
    } else {
        for (u64 t = lo + threadIdx.x; t < hi; t += 256) {
            u32 sl, e;
            if (part_entry(Dk[t], k, g, g.bps_log2, 0u, sl, e)) atomicAdd(&sm[w * SL + sl], 1u);
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
            if (part_entry(D[(u64)k * n + t], k, g, g.bps_log2, (u32)t, sl, e)) { slc[r] = sl; ent[r] = e; atomicAdd(&cnt[w * SL + sl], 1u); }
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

}  // namespace c25519

using namespace c25519;

// the launches of the digit-matrix sort over the workspace msm_enqueue_sort carved (merged layout only)
int32_t msm_matrix_sort_enqueue(c25519_ctx *ctx, const msm_geom &g, const msm_merged *md, msm_plan &pl, const msm_matrix_sort_args &a, hipStream_t st) {
    const uint64_t n = a.n, nb = pl.nb;
    const int nchunk = a.nchunk, SL = a.SL, PART_CHUNK = a.PART_CHUNK, pchunks = a.pchunks;
    const uint64_t chunk = (n + nchunk - 1) / nchunk;
    uint16_t *D = a.D;
    uint32_t *counts = a.counts, *flags = a.flags, *totals = a.totals, *ord_hist = a.ord_hist, *base = pl.base, *sorted = pl.sorted, *perm = pl.perm;
    HIPCHK(hipMemsetAsync(flags, 0, 4096, st));
    if (md) hipLaunchKernelGGL(k_digits_merged, dim3(div_up64(md->ns, 256)), dim3(256), 0, st, a.d_scalars, a.n_scalars, md->ns, md->c, md->K, D, pl.bad_ws);
    else hipLaunchKernelGGL(k_digits, dim3(div_up64(n, 256)), dim3(256), 0, st, a.d_scalars, n, g, D, pl.bad_sticky ? pl.bad_sticky : pl.bad_ws);
    if (a.use_part) {
        uint32_t *P1 = a.P1, *cc = a.cc, *bin_base = a.bin_base;
        const size_t lds1 = ((size_t)16 * SL + 2 * SL + 1 + PART_CHUNK) * 4, lds2 = ((size_t)3 * PART_BPS_MAX + PART_CAP) * 4;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(k_part_hist, dim3(g.nwin, pchunks), dim3(256), (size_t)4 * SL * 4, st, D, n, g, SL, PART_CHUNK, cc);
        hipLaunchKernelGGL(k_part_scan, dim3(g.nwin), dim3(256), 0, st, cc, SL, pchunks, g, bin_base, base);
        hipLaunchKernelGGL(k_part1, dim3(g.nwin, pchunks), dim3(1024), lds1, st, D, n, g, SL, PART_CHUNK, cc, P1);
        hipLaunchKernelGGL(k_part2, dim3(g.nwin, SL), dim3(1024), lds2, st, P1, n, g, SL, bin_base, totals, base, sorted, ord_hist, pl.max_items, pl.items, pl.counters, pl.lgids, pl.lfirst);
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
    if (!a.use_part) hipLaunchKernelGGL(k_order_hist, dim3(div_up64(nb, 256)), dim3(256), 0, st, totals, base, g, (uint64_t)0, nb, ord_hist, pl.max_items, pl.items, pl.counters, pl.lgids, pl.lfirst);
    hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(256), 0, st, ord_hist);
    hipLaunchKernelGGL(k_order_scatter, dim3(div_up64(nb, 256)), dim3(256), 0, st, totals, nb, 0u, ord_hist, perm);
    HIPCHK(hipGetLastError());
    return C25519_OK;
}

void launch_merged_table(const uint8_t *in_raw, uint64_t ns, int c, int K, uint8_t *out_raw, hipStream_t st) {
    hipLaunchKernelGGL(k_merged_table, dim3(div_up64(ns, 256)), dim3(256), 0, st, in_raw, ns, c, K, out_raw);
}
