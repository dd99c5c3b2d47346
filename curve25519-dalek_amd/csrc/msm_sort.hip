// The sort of the Pippenger MSM: per window, the term indices grouped by bucket (the scatter-add "buckets[b] += P" of pippenger.rs:122-136
// as gather lists), straight from the SCALARS -- no digit matrix -- in four launches: chunk-local partition, bin totals, per-bin counting sort
// that gathers its runs from the chunks, bucket order.
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
// 250 us instead of 331 (363 at the start of the round, 560 in round 2), four launches instead of five.  (Round 4: 86 + 5 + 73 + 18 us per
// 1.68 M terms with 17-bit windows once every window has its own slice width and no bin is oversize -- msm_slice_params.)
// Deterministic like the kernels they replace (offsets come from exact counts, not from atomics on a global cursor).
// (eight words per scalar: s' = s + addk < 2^256 whenever bit 255 of s is clear, and a scalar with bit 255 set fails the call
//  anyway (bad_scalar); a term beyond n is loaded as s = 0, whose digits are all zero: s' = addk puts 2^(wid-1) into every signed
//  window and 0 into the unsigned ones -- it is skipped like any zero digit)
struct sweep_regs { u32 s[SWEEP_TPT][8]; };
typedef unsigned int sweep_u32x4 __attribute__((ext_vector_type(4)));
template <int THREADS>
__device__ __forceinline__ void sweep_load(const uint8_t *__restrict__ scalars, u64 n, u64 lo, const msm_geom &g, sweep_regs &R, u32 *__restrict__ bad_scalar, int nt) {
#pragma unroll
    for (int r = 0; r < SWEEP_TPT; r++) {
        const u64 t = lo + (u64)r * THREADS + threadIdx.x;
        u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t < n && nt) {                                     // (r6, A/B knob SWEEP_NT: the scalars are read once per pass -- streaming policy, see msm.hip k_prep_raw2)
            const sweep_u32x4 *q = reinterpret_cast<const sweep_u32x4 *>(scalars) + 2 * t;
            const sweep_u32x4 a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        } else if (t < n) load8(scalars, t, w);
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
// THREADS: 1024 (a chunk of 8192 terms per block: the shape that is fastest ALONE) or 256 (2048 terms: one wave per SIMD and 13 KB of LDS,
// the shape that FITS beside a k_accumulate held at two waves per SIMD -- profiles/r04_ab_sort_beside_accumulate.txt)
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_sweep_local(const uint8_t *__restrict__ scalars, u64 n, msm_geom g, int SL, u32 *__restrict__ lsg, u32 *__restrict__ bad_blk,
                                                               u32 *__restrict__ P1, u64 wstride, u32 *__restrict__ zero_words, int nzero, int nt) {
    C25519_PRIO_CHAIN();
    extern __shared__ u32 sm[];
    constexpr int NW = THREADS / 64, CHUNK = THREADS * SWEEP_TPT;
    u32 *cnt = sm;                         // [SL * NW], slice-major with the bank swizzle (above)
    u32 *cur = sm + NW * SL;               // [SL * NW]
    u32 *ls = sm + 2 * NW * SL;            // [SL + 1]: start of each slice in the staging buffer, then the number of entries
    u32 *stage = ls + SL + 1;              // [SWEEP_CHUNK]
    __shared__ u32 wtot[NW];
    __shared__ u32 sbad;
    const int j = blockIdx.x, nchunk = gridDim.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < nzero; i += THREADS) zero_words[i] = 0;
    if (threadIdx.x == 0) sbad = 0;
    const int total = NW * SL, ept = total >= THREADS ? total / THREADS : 1;
    const int base = (int)threadIdx.x * ept;
    for (int i = threadIdx.x; i < total; i += THREADS) cnt[i] = 0;
    __syncthreads();
    const u64 lo = (u64)j * CHUNK;
    sweep_regs R;
    sweep_load<THREADS>(scalars, n, lo, g, R, &sbad, nt);
#pragma unroll 1
    for (int k = 0; k < g.nwin; k++) {
        const int wd = g.wid[k], bps = g.bps[k];
        u32 ent[SWEEP_TPT], slc[SWEEP_TPT];
#pragma unroll
        for (int r = 0; r < SWEEP_TPT; r++) {
            const u32 v = sweep_take(R, r, wd);
            slc[r] = 0xffffffffu;
            u32 sl, e;
            if (part_entry(v, k, g, bps, (u32)lo + (u32)r * THREADS + threadIdx.x, sl, e)) {
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
        u32 *dst = P1 + (u64)k * wstride + (u64)j * CHUNK;
        for (u32 i = threadIdx.x; i < tot; i += THREADS) dst[i] = stage[i];
        for (int i = threadIdx.x; i <= SL; i += THREADS) lsg[((u64)k * (SL + 1) + i) * nchunk + j] = ls[i];      // (SL can be 256 = THREADS of the small shape)
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


// pass 2 of the chunk-local form: the bin's entries are gathered from the chunks' blocks.  Wave w takes chunks w, w + 16, ...; a run
// of a chunk is read in pieces of 64 entries (one 256-byte segment); the pieces of a wave are listed in LDS once and walked twice --
// to count, and (from L2 now) to place: 32 pieces in registers with static indices need more than the 64 VGPRs two 1024-thread
// blocks per compute unit leave a lane (42 spilled).  A bin with more than P2G_ITER pieces per wave or more than PART_CAP entries --
// heavily skewed digits -- walks its chunks without the list and places its entries straight into the sorted array.
// NW waves per block (16: two blocks per CU; 8: 512 threads, two waves per SIMD at 56 VGPRs -- fits beside k_accumulate at two waves per SIMD), ITER:
// pieces a wave lists; chunk: terms per chunk of the partition that produced P1 (SWEEP_TPT x its block size)
template <int NW, int ITER>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 2, NW / 2)))
k_part2g(const u32 *__restrict__ P1, u64 n, u64 wstride, u32 chunk, msm_geom g, int SL, int nchunk, const u32 *__restrict__ lsg, const u32 *__restrict__ binm,
         u32 *__restrict__ totals, u32 *__restrict__ base, u32 *__restrict__ sorted,
         u32 *__restrict__ ord_hist, u32 max_items, long_item *__restrict__ items, u32 *__restrict__ counters,
         u32 *__restrict__ long_gids, u32 *__restrict__ long_first) {
    C25519_PRIO_CHAIN();
    extern __shared__ u32 sm[];
    u32 *cnt = sm, *cur = sm + PART_BPS_MAX, *oh = sm + 2 * PART_BPS_MAX, *out = sm + 3 * PART_BPS_MAX, *wl = out + PART_CAP;      // wl[NW][ITER]
    __shared__ u32 red[NW], red_all[NW];
    const int k = blockIdx.x, sidx = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int PART_BPS = 1 << g.bps[k], sh = part_entry_shift(g);
    // Everything the block needs from memory before the gather is requested together: the bin totals (the bin's place in the window's
    // sorted list = the entries of the bins before it; SL <= 256 <= the block) and this wave's run starts -- the gather does not wait
    // for the prefix sum.
    const u32 *row0 = lsg + ((u64)k * (SL + 1) + sidx) * nchunk, *row1 = row0 + nchunk;
    const u32 *src = P1 + (u64)k * wstride;
    const u32 mine = tid < SL ? binm[(u64)k * SL + tid] : 0u;
    u32 part = tid < sidx ? mine : 0u, all = mine;        // entries of the bins before this one / of the whole window
    const u32 m = binm[(u64)k * SL + sidx];
    if (tid < PART_BPS) cnt[tid] = 0;
    if (tid < 256) oh[tid] = 0;
    // this wave's pieces: (offset in the window's P1 region) << 7 | entries in the piece
    int nslots = 0;
    for (int j0 = 0; j0 < nchunk; j0 += NW * 64) {
        const int j = j0 + w + NW * lane;
        u32 st = 0, len = 0;
        if (j < nchunk) { st = row0[j]; len = row1[j] - st; }
        const u32 np = (len + 63u) >> 6;
        u32 inc = np;
        for (int off = 1; off < 64; off <<= 1) { const u32 x = __shfl_up(inc, off, 64); if (lane >= off) inc += x; }
        const u32 first = (u32)nslots + inc - np;
        for (u32 p = 0; p < np; p++)
            if (first + p < (u32)ITER) wl[w * ITER + first + p] = (((u32)j * chunk + st + 64u * p) << 7) | (len - 64u * p < 64u ? len - 64u * p : 64u);
        nslots += (int)__shfl(inc, 63, 64);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { part += (u32)__shfl_xor((int)part, d, 64); all += (u32)__shfl_xor((int)all, d, 64); }
    if (lane == 0) { red[w] = part; red_all[w] = all; }
    const bool fits = !__syncthreads_or(nslots > ITER) && m <= (u32)PART_CAP;      // (the barrier: counters zeroed, wave sums of the prefix written)
    u32 b0 = 0, wtot = 0;
#pragma unroll
    for (int q = 0; q < NW; q++) { b0 += red[q]; wtot += red_all[q]; }
    if (sidx == SL - 1 && tid == 0) base[(u64)k * (g.half + 1) + g.half] = wtot;      // number of entries of the window
    {   // a window narrower than c bits leaves the buckets from SL << bps[k] on unused: empty lists at the end of the window's entries, this
        // block's share of them (the bucket order and the accumulation walk ALL half buckets of every window)
        const int per = (1 << g.bps_log2) - PART_BPS, first = (SL << g.bps[k]) + sidx * per;
        for (int i = tid; i < per; i += 64 * NW) { totals[(u64)k * g.half + first + i] = 0; base[(u64)k * (g.half + 1) + first + i] = wtot; }
        if (per > 0 && tid == 0) atomicAdd(&ord_hist[256 * msm_group_of(g, k) + 255], (u32)per);      // (length class of an empty list; one histogram per window group)
    }
    u32 *dst = sorted + (u64)k * n + b0;
    if (fits) {
#pragma unroll 1
        for (int t0 = 0; t0 < nslots; t0 += 8) {                         // eight pieces in flight
            u32 ev[8];
            bool ok[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const u32 d = t0 + q < nslots ? wl[w * ITER + t0 + q] : 0u;
                ok[q] = (u32)lane < (d & 127u);
                ev[q] = ok[q] ? src[(d >> 7) + lane] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) if (ok[q]) atomicAdd(&cnt[ev[q] >> sh], 1u);
        }
    } else {
        for (int j = w; j < nchunk; j += NW) {
            const u32 st = row0[j], en = row1[j];
            for (u32 o = st + lane; o < en; o += 64) atomicAdd(&cnt[src[(u64)j * chunk + o] >> sh], 1u);
        }
    }
    __syncthreads();
    if (tid < 64) {                                                    // exclusive scan of the bucket counts by one wave (8, 4, 2 or 1 per lane)
        const int per = PART_BPS >> 6;
        u32 c4[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sum = 0;
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
    if (tid < 256 && oh[tid]) atomicAdd(&ord_hist[256 * msm_group_of(g, k) + tid], oh[tid]);
    if (fits) {
#pragma unroll 1
        for (int t0 = 0; t0 < nslots; t0 += 8) {
            u32 ev[8];
            bool ok[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const u32 d = t0 + q < nslots ? wl[w * ITER + t0 + q] : 0u;
                ok[q] = (u32)lane < (d & 127u);
                ev[q] = ok[q] ? src[(d >> 7) + lane] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) if (ok[q]) out[atomicAdd(&cur[ev[q] >> sh], 1u)] = part_entry_final(ev[q], sh);
        }
        __syncthreads();
        for (u32 i = tid; i < m; i += 64 * NW) dst[i] = out[i];
    } else {
        // oversize bin = heavily skewed digits (e.g. one bucket holding most of the window).  Entries go straight to their final place;
        // lanes of a wave that share the first lane's bucket take their slots with ONE atomic.
        for (int j = w; j < nchunk; j += NW) {
            const u32 st = row0[j], en = row1[j];
            for (u32 o0 = st; o0 < en; o0 += 64) {
                const u32 o = o0 + lane;
                const bool have = o < en;
                const u32 ev = have ? src[(u64)j * chunk + o] : 0u, bk = ev >> sh;
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
                if (have) dst[pos] = part_entry_final(ev, sh);
            }
        }
    }
}

// the same with the scan inside: every block scans the 256-bin histogram itself (read-only) and takes its slots from a
// separate cursor array (zeroed by k_sweep_local) -- one launch less in the chain
// Window groups (msm_geom): every group has its own length histogram, its own cursors and its own stretch of perm -- the buckets of windows
// [gstart[q], gstart[q + 1]) in decreasing list length at perm[gstart[q] * half ..) -- so that k_accumulate can be launched group by group.  A block's
// buckets belong to ONE window (groups are only formed when half >= BS).
template <int BS>                                            // 256 bins, BS >= 256 threads: a block's buckets per bin take their slots with ONE global atomic per bin
__global__ void __launch_bounds__(BS) k_order_place(const u32 *__restrict__ totals, u64 nb, const u32 *__restrict__ ord_hist, u32 *__restrict__ ord_cursor, u32 *__restrict__ perm, msm_geom g) {
    C25519_PRIO_CHAIN();
    __shared__ u32 h[256], start[256], basep[256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int grp = g.ngroups > 1 ? msm_group_of(g, (int)(((u64)blockIdx.x * BS) / (u64)g.half)) : 0;
    ord_hist += 256 * grp; ord_cursor += 256 * grp; perm += (u64)g.gstart[grp] * (u64)g.half;
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

}  // namespace c25519

using namespace c25519;

// the bucket order alone (mid.hip): totals[gid] = list length, ord_hist = the call's 256-bin histogram of min(length, 255) (index 255 - length), ord_cursor = 256 zero words
void launch_order_place(const uint32_t *totals, uint64_t nb, const uint32_t *ord_hist, uint32_t *ord_cursor, uint32_t *perm, const c25519::msm_geom &g, hipStream_t st) {
    hipLaunchKernelGGL(k_order_place<1024>, dim3(div_up64(nb, 1024)), dim3(1024), 0, st, totals, nb, ord_hist, ord_cursor, perm, g);
}

// ================================================================================================
// host side: workspace carve-up and the launches of the sort
// ================================================================================================
// sort / long-bucket parameters that depend on the number of terms per window (n) and buckets per window (g.half)
// slices per window of the chunk-local sort: SL = half >> bps_log2 for every window, of 2^bps[k] buckets each.  A signed window of w bits has
// digits in [-2^(w-1), 2^(w-1)): 2^(w-1) buckets; an unsigned one of w bits 2^w - 1.  With one slice width for all windows (rounds 2-3) the
// windows one bit narrower than c -- up to two signed ones and the unsigned one below the overflow window -- filled only the lower half of
// their slices with twice the entries: 128 - 192 bins per pass beyond the LDS capacity of k_part2g, whose global-memory path was a third
// (16-bit windows) to more than half (17-bit) of that kernel's time (profiles/r04_sort_alone.txt).
void msm_slice_params(msm_geom &g) {
    int lsl = 0; while ((g.half >> g.bps_log2) > (1 << lsl)) lsl++;          // log2(SL)
    for (int k = 0; k < MSM_MAX_WIN; k++) {
        const int bits = k < g.nwin ? (k < g.first_unsigned ? g.wid[k] - 1 : g.wid[k]) : 0;     // log2 of the window's bucket range (rounded up)
        int b = bits - lsl;
        if (b < 6) b = 6;                                  // k_part2g scans its bucket counts with one wave: at least 64 buckets per slice
        if (b > g.bps_log2) b = g.bps_log2;
        g.bps[k] = (unsigned char)b;
    }
}
void msm_sort_params(uint64_t n, msm_geom &g) {
    // slices of 2^bps_log2 buckets such that a (window, slice) bin holds at most ~16 K entries (PART_CAP with 12 % headroom)
    // (17-bit windows: 512 buckets per slice keep the 128 slices per window -- and with them the partition's LDS footprint and the number of
    //  per-bin blocks -- of the 16-bit layout; the 22-bit term index that leaves is enough for a pass)
    g.bps_log2 = (g.half > (1 << 15) && n <= (1ull << 22)) ? 9 : 8;
    while (g.bps_log2 > 6 && (n << g.bps_log2) / (uint64_t)g.half > 16500) g.bps_log2--;
    if ((1 << g.bps_log2) > g.half) { g.bps_log2 = 0; while ((2 << g.bps_log2) <= g.half) g.bps_log2++; }
    const uint64_t mean = n / (uint64_t)g.half + 1;
    g.long_cap = (u32)std::max<uint64_t>(LONG_CAP_MIN, (mean * 5 + 1) / 2);
}

// md (may be null): merged layout -- d_scalars holds n_scalars scalars, the sort runs over md->K * md->ns digit-terms (digit-matrix sort)
// n_carve (0 = n): the number of terms the workspace is carved for -- passes that CONTINUE each other's bucket sums (msm_record_enqueue)
// must find the buckets at the same address although the last pass is shorter
int32_t msm_enqueue_sort(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n_scalars, const msm_geom &g, uint32_t *d_slot, hipStream_t sort_stream, msm_plan &pl,
                         const msm_merged *md, uint64_t n_carve, hipEvent_t lists_free, int parity) {
    (void)d_slot;
    const uint64_t n = md ? (uint64_t)md->K * md->ns : n_scalars;
    const uint64_t nc = n_carve > n ? n_carve : n;
    // The chunk-local sort serves every plain MSM pass: its per-bin counting sort scans 2^bps_log2 >= 64 buckets per wave, i.e. windows of
    // c >= 7 bits (n >= 2048 terms; inputs below 4096 terms never get here, msm_small_enqueue), and its entries keep a 23-bit term index.
    if (!md && (g.half < 64 || n > (1ull << (part_entry_shift(g) - 1)))) { ctx->err = "msm: internal error (a pass outside the range of the chunk-local sort)"; return -(int32_t)hipErrorInvalidValue; }
    // which sort: the digit-matrix sort for the merged layout and for plain passes below 2^16 terms (the chunk-local partition wants hundreds of
    // chunks: msm_sort_matrix.hip has the numbers); C25519_SORT_CHUNK_LOCAL_MIN lowers the boundary (tests run the chunk-local sort from 2 048 terms)
    static const uint64_t chunk_local_min = (uint64_t)C25519_KNOB("SORT_CHUNK_LOCAL_MIN", 1 << 16);
    const bool matrix = md != nullptr || nc < chunk_local_min;
    // every region BEFORE the buckets (oK) must be sized from nc, the number of terms the call's passes are carved for, never from
    // this pass's own n: a shorter last pass that CONTINUES its predecessor's bucket sums has to find them at the same offset
    // (round 3 derived nchunk from n: with C25519_MSM_PASS_LOG2 = 21 / 22 a last pass one chunk shorter moved oC .. oK)
    int nchunk = std::max(1, std::min(64, 512 / g.nwin));
    while (nchunk > 1 && nc / nchunk < 4096) nchunk /= 2;
    if ((nc + nchunk - 1) / nchunk > 65536) nchunk = (int)((nc + 65535) / 65536);   // a chunk's digits must fit LDS (k_scatter_sliced)
    const uint64_t nb = (uint64_t)g.nwin * g.half;
    const int nseg = red_nseg(g.half);
    // workspace carve-up (tmp_d): [digit matrix | counts] (merged layout only) | base | sorted | buckets | segment pairs | flags | perm | long-bucket lists | sort scratch
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    // parity >= 0 (the passes of a multi-pass call): what the accumulation READS of the sort -- bucket bases, gather lists, bucket order -- exists
    // twice, and the pass uses copy `parity`: the sort of the next pass on this workspace can then run beside this pass's accumulation
    const int copies = parity >= 0 ? 2 : 1, cp = parity >= 0 ? (parity & 1) : 0;
    const size_t szB = ((size_t)g.nwin * (g.half + 1) * 4 + 255) & ~(size_t)255, szS = ((size_t)g.nwin * nc * 4 + 255) & ~(size_t)255, szPerm = (nb * 4 + 255) & ~(size_t)255;
    const size_t oD = carve(matrix ? (size_t)g.nwin * nc * 2 : 0), oC = carve(matrix ? (size_t)g.nwin * nchunk * g.half * 4 : 0), oB = carve(szB * copies) + szB * cp;
    const size_t oS = carve(szS * copies) + szS * cp, oK = carve(nb * 160), oT = carve(nb * 4);
    const size_t oSW = carve((size_t)g.nwin * nseg * 2 * 160), oF = carve(16384), oPerm = carve(szPerm * copies) + szPerm * cp;
    // long-bucket path: at most (#entries / LONG_SEG + #long buckets) work items; a long bucket has > LONG_CAP entries
    const uint64_t entries = (uint64_t)g.nwin * nc;
    const uint32_t max_long = (uint32_t)std::min<uint64_t>(nb, entries / g.long_cap + 1);
    const uint32_t max_items = (uint32_t)(entries / LONG_SEG + max_long + 1);
    const size_t oLI = carve((size_t)max_items * sizeof(long_item)), oLG = carve((size_t)max_long * 4), oLF = carve((size_t)max_long * 4);
    const size_t oLS = carve((size_t)max_items * 160);
    const bool use_part = matrix && g.c >= 13 && n <= (1ull << 23) && n >= (1ull << 16);          // the two-pass partition of the digit-matrix sort
    // block shapes of the chunk-local sort: the large ones are the fastest alone; the small ones fit beside a k_accumulate held at two waves
    // per SIMD (one wave per SIMD at 128 VGPRs / two at 56; profiles/r04_ab_sort_beside_accumulate.txt).  Read once per process.
    static const int small_blocks = C25519_KNOB("SORT_SMALL", 0);
    // C25519_SWEEP_THREADS=512: the partition in 512-thread blocks (chunks of 4096 terms: two waves per SIMD at 64 VGPRs and 25 KB of LDS) with the
    // per-bin sort in its large shape (four waves per SIMD at 32 VGPRs, 76 KB) -- the pairing that fits beside a two-wave k_accumulate without
    // the 16-entry runs of the all-small arm
    static const int sweep_threads = small_blocks ? 256 : (C25519_KNOB("SWEEP_THREADS", SWEEP_THREADS) == 512 ? 512 : SWEEP_THREADS);
    const int sweep_chunk = sweep_threads * SWEEP_TPT;
    const int SL = std::max(1, g.half >> g.bps_log2), PART_CHUNK = matrix ? part_chunk(SL) : sweep_chunk, pchunks = (int)((n + PART_CHUNK - 1) / PART_CHUNK), pchunks_c = (int)((nc + PART_CHUNK - 1) / PART_CHUNK);
    // chunk-local form: P1 holds whole chunk blocks, oCC the slice starts [window][SL + 1][chunk], oBB the bin totals and the chunks' flags
    const size_t p1_words = matrix ? (size_t)g.nwin * nc : (size_t)g.nwin * pchunks_c * sweep_chunk;
    size_t oP1 = 0, oCC = 0, oBB = 0;
    if (!matrix || use_part) { oP1 = carve(p1_words * 4); oCC = carve((size_t)g.nwin * (SL + 1) * pchunks_c * 4); oBB = carve((size_t)g.nwin * (SL + 1) * 4 + (size_t)pchunks_c * 4); }
    int32_t r = ctx_reserve(ctx, ctx->tmp_d, off);
    if (r) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_d.p;
    uint32_t *base = (uint32_t *)(ws + oB), *sorted = (uint32_t *)(ws + oS), *buckets = (uint32_t *)(ws + oK);
    // small words of the chain (u32 index): [8..] long-bucket counters, [64..319] bucket-order histogram, [320..575] its cursors,
    // [576] "a scalar has bit 255 set" (ORed into the result slot by the bucket reduction: the sort itself never touches the slot)
    uint32_t *flags = (uint32_t *)(ws + oF), *totals = (uint32_t *)(ws + oT), *ord_hist = flags + 64, *perm = (uint32_t *)(ws + oPerm);
    // (one 256-bin length histogram and one set of cursors per window group: 64 + 512 G words, G <= MSM_MAX_GROUPS, zeroed by the partition kernel)
    const int ngr = g.ngroups > 1 ? g.ngroups : 1;
    uint32_t *ord_cursor = flags + 64 + 256 * ngr, *bad_ws = flags + 64 + 512 * ngr;
    const int ZERO_WORDS = 64 + 512 * ngr;
    if (ngr > 1 && (matrix || g.half < 1024)) { ctx->err = "msm: internal error (window groups outside the chunk-local sort)"; return -(int32_t)hipErrorInvalidValue; }
    pl.g = g; pl.n = n; pl.nb = nb; pl.nseg = nseg; pl.max_items = max_items; pl.max_long = max_long;
    pl.base = base; pl.sorted = sorted; pl.buckets = buckets; pl.perm = perm; pl.SW = (uint32_t *)(ws + oSW); pl.counters = flags + 8; pl.bad_ws = bad_ws;
    pl.items = (long_item *)(ws + oLI); pl.lgids = (uint32_t *)(ws + oLG); pl.lfirst = (uint32_t *)(ws + oLF); pl.segs = (uint32_t *)(ws + oLS);
    pl.sort_stream = sort_stream;
    hipStream_t st = sort_stream ? sort_stream : ctx->stream;
    if (matrix) {
        if (lists_free) HIPCHK(hipStreamWaitEvent(st, lists_free, 0));     // (the digit-matrix sort is not split: all of it waits)
        msm_matrix_sort_args a;
        a.d_scalars = d_scalars; a.n_scalars = n_scalars; a.n = n; a.nchunk = nchunk; a.use_part = use_part; a.SL = SL; a.PART_CHUNK = PART_CHUNK; a.pchunks = pchunks;
        a.D = (uint16_t *)(ws + oD); a.counts = (uint32_t *)(ws + oC); a.P1 = (uint32_t *)(ws + oP1); a.cc = (uint32_t *)(ws + oCC); a.bin_base = (uint32_t *)(ws + oBB);
        a.flags = flags; a.totals = totals; a.ord_hist = ord_hist;
        return msm_matrix_sort_enqueue(ctx, g, md, pl, a, st);
    }
    uint32_t *P1 = (uint32_t *)(ws + oP1), *lsg = (uint32_t *)(ws + oCC), *binm = (uint32_t *)(ws + oBB), *bad_blk = binm + (size_t)g.nwin * (SL + 1);
    const uint64_t wstride = (uint64_t)pchunks_c * sweep_chunk;
    constexpr int ITER_SMALL = 160;                                          // pieces per wave: up to 1024 chunks of 2048 terms over 8 waves
    const int nw1 = sweep_chunk / SWEEP_TPT / 64;
    const size_t lds1 = ((size_t)2 * nw1 * SL + SL + 1 + sweep_chunk) * 4;
    const size_t lds2 = ((size_t)3 * PART_BPS_MAX + PART_CAP + (small_blocks ? 8 * ITER_SMALL : 16 * P2G_ITER)) * 4;
    static const int sweep_nt = C25519_KNOB("SWEEP_NT", 0);
    if (small_blocks) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep_local<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part2g<8, ITER_SMALL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(k_sweep_local<256>, dim3(pchunks), dim3(256), lds1, st, d_scalars, n, g, SL, lsg, bad_blk, P1, wstride, flags, ZERO_WORDS, sweep_nt);
    } else if (sweep_threads == 512) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part2g<16, P2G_ITER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(k_sweep_local<512>, dim3(pchunks), dim3(512), lds1, st, d_scalars, n, g, SL, lsg, bad_blk, P1, wstride, flags, ZERO_WORDS, sweep_nt);
    } else {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep_local<SWEEP_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part2g<16, P2G_ITER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL(k_sweep_local<SWEEP_THREADS>, dim3(pchunks), dim3(SWEEP_THREADS), lds1, st, d_scalars, n, g, SL, lsg, bad_blk, P1, wstride, flags, ZERO_WORDS, sweep_nt);
    }
    hipLaunchKernelGGL(k_bin_totals, dim3((g.nwin * SL + 3) / 4), dim3(256), 0, st, lsg, pchunks, SL, g.nwin * SL, binm, bad_blk, bad_ws, pl.bad_sticky);
    if (pl.ev_partition) HIPCHK(hipEventRecord(pl.ev_partition, st));
    // lists_free: the partition above writes only the sort's own scratch (chunk blocks, slice starts, bin totals, the small counters of the chain);
    // the gather lists, bucket bases, totals and the bucket order -- what the accumulation of the PREVIOUS pass on this workspace still reads --
    // are written from here on
    if (lists_free) HIPCHK(hipStreamWaitEvent(st, lists_free, 0));
    if (small_blocks) hipLaunchKernelGGL((k_part2g<8, ITER_SMALL>), dim3(g.nwin, SL), dim3(512), lds2, st, P1, n, wstride, (u32)sweep_chunk, g, SL, pchunks, lsg, binm, totals, base, sorted, ord_hist, max_items, pl.items, pl.counters, pl.lgids, pl.lfirst);
    else hipLaunchKernelGGL((k_part2g<16, P2G_ITER>), dim3(g.nwin, SL), dim3(1024), lds2, st, P1, n, wstride, (u32)sweep_chunk, g, SL, pchunks, lsg, binm, totals, base, sorted, ord_hist, max_items, pl.items, pl.counters, pl.lgids, pl.lfirst);
    hipLaunchKernelGGL(k_order_place<1024>, dim3(div_up64(nb, 1024)), dim3(1024), 0, st, totals, nb, ord_hist, ord_cursor, perm, g);
    HIPCHK(hipGetLastError());
    return C25519_OK;
}
