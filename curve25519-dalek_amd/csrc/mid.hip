// The MID-SIZE multiscalar multiplication (round 6): 6144 .. 2^18 terms (verify_batch: from 2048 signatures; both from 12 288 terms until late in the round) in SIX short
// launches on ONE stream.
//
// Reference: the same algorithm as msm.hip -- backend/serial/scalar_mul/pippenger.rs:67-160 (signed digits, buckets, running-sum reduction, Horner fold), for the sizes
// where the reference's heaviest real callers live (edwards.rs:1002-1031 vartime_multiscalar_mul of a Bulletproofs verification; ed25519-dalek/src/batch.rs:225-244
// verify_batch of 2^13 .. 2^17 signatures).
//
// Why a third path.  Between the small path (small.hip, up to 6143 terms) and the throughput regime (2^20 terms and beyond) the bucket pipeline of msm.hip is a
// chain of 15 - 27 short kernels on two streams: the HOST's launch loop (5 - 8 us per launch, a dozen event operations) put k_accumulate 150 us into a 330 us call
// at 2^14 terms, the batched inversion of the normaliser is a 78 us latency chain on 64 waves, and the eleven sort kernels of 2 - 9 us each are separated by as much
// again (profiles/r05_timeline_msm_2p14.txt; whole call 0.03 - 0.17 of the multiplier roof).  Here:
//
//   k_mid_front   one lane per term: the window digits of s' = s + addk as a u16 matrix D[window][term], and -- raw points -- the point as a PROJECTIVE Niels
//                 record (Y+X, Y-X, Z, 2dT) of (XZ : YZ : Z^2 : XY): five field operations, NO inversion.  The accumulation then costs 8 M per addition instead
//                 of 7 M, which is nothing beside a 265-operation inversion chain while the machine is not full (the small path made the same choice).
//                 Also zeroes the small counters of the kernels behind it (it always precedes them on the stream: nothing relies on a previous call's clean-up).
//   k_mid_sort    one block per (window, slice of <= 4096 buckets): count -> scan -> place over the window's row of D, straight into the final gather lists
//                 (two reads of a row that sits in L2; no partition pass, no digit re-derivation), the bucket bases, the list lengths with their histogram, and the
//                 work items of over-long lists.
//   k_order_place (msm_sort.hip) all buckets of the call in the order of their list lengths, longest first: the accumulation's makespan is its longest lists, and
//                 they must start first (a block-local order -- the first version -- cost 132 against 79 us at 2^16 terms).
//   k_mid_acc_long  one lane per bucket walking its list (8 M additions on projective Niels records; prepared records -- verify_batch / compressed inputs -- take
//                 accum.hip k_accumulate_long: 7 M mixed additions on the affine records a decompression left); blocks IN FRONT of the bucket lanes fold the
//                 over-long lists segment by segment (mid_long.h), the wave that finishes a bucket's last segment sums the segments (no separate combine launch, no
//                 second stream).  Which lists are over-long: mid_pick_cap -- the accumulation lasts as long as the longest list a lane keeps.
//   k_reduce_a4, k_reduce_b4pub (reduce.hip)  the two levels of the bucket reduction; the level-B block that finishes the last window writes the record's header --
//                 and, for a caller that reads the record on the host right away, publishes it into page-locked host memory and releases the sequence word the
//                 host polls (the small path's mechanism, msm.hip wait_published, with the same recovery).
//
// Same window layout (msm_layout), same record format, same host fold as the other two paths: a partial-result record of this path folds with any other.
// Variable time like the paths beside it (digits index gather lists): vartime_multiscalar_mul and verify_batch only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <stdlib.h>
#include <string>
#define C25519_CHAIN 1
#include "../../include/c25519_hip.h"
#include "devio.h"
#include "ctx.h"
#include "msm_internal.h"
#include "msm_sort.h"
#include "fe26x.h"
#include "mid_long.h"

using namespace c25519;
#define HIPCHK(call)                                                \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return c25519_fail(ctx, _e, #call);   \
    } while (0)

namespace c25519 {


// ---- front: digits + records --------------------------------------------------------------------------------------------------------------------------------
// SRC 0: raw 160-byte points -> projective Niels records (the same X, Y, Z-only reading as the other two paths: T is whatever the caller stored and is not used);
// SRC 1: the records exist already (affine Niels, made by a decompression): digits only.
template <int SRC>
__global__ void __launch_bounds__(256) k_mid_front(const uint8_t *__restrict__ scalars, const uint8_t *__restrict__ points, u64 n, msm_geom g, uint16_t *__restrict__ D, u64 dstride,
                                                   u32 *__restrict__ recs, u32 *__restrict__ zero_words, int nzero, u32 *__restrict__ blockflags) {
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < (u32)nzero; i += gridDim.x * 256u) zero_words[i] = 0;      // (all blocks: with the caps of mid_pick_cap the per-bucket counters of the long path are tens of thousands of words)
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    int bad = 0;
    if (t < dstride) {
        // (the rows of D are padded to a multiple of eight terms with the scalar 0, whose digits are all zero -- s' = addk puts 2^(wid-1) into every signed window
        //  and 0 into the unsigned ones: the sort reads whole 16-byte vectors without a bound check)
        u32 s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t < n) load8(scalars, t, s);
        bad = (int)(s[7] >> 31);
        u32 carry = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { const u64 v = (u64)s[i] + g.addk[i] + carry; s[i] = (u32)v; carry = (u32)(v >> 32); }
        // the windows are contiguous from bit 0 (msm_layout): window k is the low wid[k] bits, then the scalar moves down -- no dynamic register index
#pragma unroll 1
        for (int k = 0; k < g.nwin; k++) {
            const int wd = g.wid[k];
            D[(u64)k * dstride + t] = (uint16_t)(s[0] & ((1u << wd) - 1u));
#pragma unroll
            for (int i = 0; i < 7; i++) s[i] = __funnelshift_r(s[i], s[i + 1], (u32)wd);
            s[7] >>= wd;
        }
        if (SRC == 0 && t < n) {
            const feT X = raw160_fe(points, t, 0), Y = raw160_fe(points, t, 1), Z = raw160_fe(points, t, 2);
            const feT x = fe_mul(X, Z), y = fe_mul(Y, Z), zz = fe_sq(Z), tt = fe_mul(fe_mul(X, Y), fe_d2());
            const feT a = fe_carry(fe_add(y, x)), b = fe_carry(fe_sub(y, x));
            u32 w[40];
#pragma unroll
            for (int i = 0; i < 10; i++) { w[i] = a.v[i]; w[10 + i] = b.v[i]; w[20 + i] = zz.v[i]; w[30 + i] = tt.v[i]; }
            uint4 *q = reinterpret_cast<uint4 *>(recs) + MID_REC_Q * t;
#pragma unroll
            for (int i = 0; i < MID_REC_Q; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
        }
    }
    const int any = __syncthreads_or(bad);
    if (threadIdx.x == 0) blockflags[blockIdx.x] = (u32)any;      // "a scalar has bit 255 set", one word per block (ORed by the reduction's last block: no cleared word needed)
}

// ---- sort: one block per (window, slice) --------------------------------------------------------------------------------------------------------------------------
// D row of window k (dstride u16, a multiple of 8, padded with zero digits): read twice as 16-byte vectors.  Pass 1 counts the bucket occupancies of THIS slice, the
// entries of the slices before it (= where this slice's lists start in the window's sorted array) and of the whole window; then scan, bases, bucket order, long-list
// items; pass 2 places term | sign << 31 at its final position (LDS cursors).  Any digit distribution is correct (all terms in one bucket: one long list, LDS atomics
// on one counter).
// Every block of a window reads the window's whole row, so a block's time is n / 1024 digits per thread and pass whatever the slicing, and the digit loop is what it
// costs: ~13 instructions per digit, and an LDS atomic only for the digits of THIS slice -- an LDS atomic costs by its active lanes (~1 lane per cycle), so "one atomic
// per digit, into a spare counter when the digit is outside the slice" (the branch-free form) was 64 us at 2^16 terms; a branch per condition with ~50 instructions
// per digit and every cursor atomic of pass 2 behind its own wait: 74 us at 2^16, 235 us at 2^18 terms (profiles/r06_timeline_mid_first.txt).
// Bucket order: the kernel leaves every list's length (totals) and the call's 256-bin length histogram; k_order_place (msm_sort.hip) then orders ALL buckets of the
// call by length, longest first.  The accumulation's makespan is its longest lists and they must start first: a slice-by-slice order (long lists also in the last
// blocks to be dispatched) cost the same k_accumulate 132 against 79 us at 2^16 terms; interleaving the slices' ranks repaired that but mixes the narrow windows'
// lists -- twice as long in half the buckets -- into every wave: 323 against 200 us at 2^18 (profiles/r06_timeline_mid_first.txt).
constexpr int MID_BPS_MAX = 4096, MID_SORT_THREADS = 1024, MID_PER_MAX = MID_BPS_MAX / MID_SORT_THREADS;
// The digit tests on the STORED value v (d = v - hw; bucket = |d| - 1; hw = 2^(wid-1) for a signed window, 0 for an unsigned one, whose values above `half` -- only a
// scalar with bit 255 set has one, and the front kernel has flagged it -- are dropped like digit_of drops them): with r = s * BPS the slice's first bucket,
//   in this slice      <=>  |d| in [r + 1, r + BPS]   <=>  (v - (hw + r + 1)) <u BPS  or  ((hw - r - 1) - v) <u BPS
//   in a slice before  <=>  0 < |d| <= r              <=>  v != hw  and  (v - (hw - r)) <=u 2 r
// -- compares on block-uniform constants instead of a decode per digit (first versions: ~40 instructions per digit, 60 us at 2^16 terms).
struct mid_slice_consts { u32 hw, lo_pos, lo_neg, lo_before, span_before, vmax, bps_count; };
__global__ void __launch_bounds__(MID_SORT_THREADS) k_mid_sort(const uint16_t *__restrict__ D, u64 n, u64 dstride, msm_geom g, int bps, u32 *__restrict__ sorted, u32 *__restrict__ base,
                                                               u32 *__restrict__ totals, u32 *__restrict__ ord_hist, u32 max_items, mid_item *__restrict__ items, u32 *__restrict__ counters,
                                                               u32 *__restrict__ long_gids) {
    __shared__ u32 cnt[MID_BPS_MAX], cur[MID_BPS_MAX], oh[256], red[3][16];
    const int k = blockIdx.x, s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int BPS = 1 << bps, PER = BPS > MID_SORT_THREADS ? BPS / MID_SORT_THREADS : 1;      // consecutive buckets per thread (1, 2 or 4)
    const bool uns = k >= g.first_unsigned;
    mid_slice_consts K;
    {
        const u32 r = (u32)s * (u32)BPS;
        K.hw = uns ? 0u : 1u << (g.wid[k] - 1);
        K.vmax = uns ? (u32)g.half : 0xffffu;
        K.bps_count = (u32)BPS;
        K.lo_pos = K.hw + r + 1u;                 // v - lo_pos = bucket - r for a positive digit
        K.lo_neg = K.hw - r - 1u;                 // lo_neg - v = bucket - r for a negative digit (wraps far above BPS for an unsigned window or a positive digit)
        K.lo_before = K.hw - r;                   // |d| <= r  <=>  v - (hw - r) <= 2 r   (r = 0: only v = hw, excluded by the non-zero test)
        K.span_before = 2u * r;
        if (uns) { K.lo_neg = 0x80000000u; K.lo_before = 0u; K.span_before = r; }      // unsigned window: no negative digits; |d| <= r <=> v <= r
    }
    for (int i = tid; i < BPS; i += MID_SORT_THREADS) cnt[i] = 0;
    if (tid < 256) oh[tid] = 0;
    __syncthreads();
    const uint4 *row = reinterpret_cast<const uint4 *>(D + (u64)k * dstride);
    const u32 nvec = (u32)(dstride / 8);
    u32 before = 0, nz = 0;
    // four 16-byte loads in flight per thread
#pragma unroll 1
    for (u32 i0 = tid; i0 < nvec; i0 += 4 * MID_SORT_THREADS) {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const u32 i = i0 + (u32)q * MID_SORT_THREADS; v[q] = row[i < nvec ? i : nvec - 1]; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool live = i0 + (u32)q * MID_SORT_THREADS < nvec;
            const u32 x[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
            for (int h = 0; h < 8; h++) {
                const u32 vv = (h & 1) ? x[h >> 1] >> 16 : x[h >> 1] & 0xffffu;
                const bool ok = live && vv <= K.vmax;
                const u32 dp = vv - K.lo_pos, dn = K.lo_neg - vv;
                const bool on = ok && vv != K.hw;
                nz += on ? 1u : 0u;
                before += (on && (vv - K.lo_before) <= K.span_before) ? 1u : 0u;
                if (ok && (dp < K.bps_count || dn < K.bps_count)) atomicAdd(&cnt[dp < K.bps_count ? dp : dn], 1u);      // (an LDS atomic costs by its ACTIVE lanes)
            }
        }
    }
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) { before += (u32)__shfl_xor((int)before, dd, 64); nz += (u32)__shfl_xor((int)nz, dd, 64); }
    if (lane == 0) { red[0][w] = before; red[1][w] = nz; }
    __syncthreads();                                             // the counts are complete
    before = 0; nz = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) { before += red[0][q]; nz += red[1][q]; }
    // exclusive scan of the slice's bucket counts: thread tid owns buckets tid * PER .. tid * PER + PER - 1
    u32 c[MID_PER_MAX], csum = 0;
#pragma unroll
    for (int j = 0; j < MID_PER_MAX; j++) { c[j] = (j < PER && tid * PER + j < BPS) ? cnt[tid * PER + j] : 0u; csum += c[j]; }
    u32 inc = csum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 y = (u32)__shfl_up((int)inc, off, 64); if (lane >= off) inc += y; }
    if (lane == 63) red[2][w] = inc;
    // the slice's share of the call's length histogram (255 - min(c, 255): longest first)
#pragma unroll
    for (int j = 0; j < MID_PER_MAX; j++)
        if (j < PER && tid * PER + j < BPS) atomicAdd(&oh[255u - (c[j] > 255u ? 255u : c[j])], 1u);
    __syncthreads();
    u32 wb = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) wb += q < w ? red[2][q] : 0u;
    u32 lo[MID_PER_MAX];
    {
        u32 run = before + wb + inc - csum;                      // start of this thread's first list in the window's sorted array
#pragma unroll
        for (int j = 0; j < MID_PER_MAX; j++) {
            lo[j] = run; run += c[j];
            if (j < PER && tid * PER + j < BPS) {
                cur[tid * PER + j] = lo[j];
                base[(u64)k * (g.half + 1) + (u64)s * BPS + tid * PER + j] = lo[j];
            }
        }
    }
    if (s == (int)gridDim.y - 1 && tid == 0) base[(u64)k * (g.half + 1) + g.half] = nz;
    __syncthreads();
    if (tid < 256 && oh[tid]) atomicAdd(&ord_hist[tid], oh[tid]);
#pragma unroll
    for (int j = 0; j < MID_PER_MAX; j++) {
        if (!(j < PER && tid * PER + j < BPS)) continue;
        const u64 G = (u64)k * g.half + (u64)s * BPS + tid * PER + j;          // global bucket index
        totals[G] = c[j];
        if (c[j] > g.long_cap) {                                // an over-long list: work items of MID_LONG_SEG entries each (k_mid_long)
            const u32 nseg = (c[j] + MID_LONG_SEG - 1) / MID_LONG_SEG;
            const u32 first = atomicAdd(&counters[0], nseg);
            const u32 lb = atomicAdd(&counters[1], 1u);
            long_gids[lb] = (u32)G;
            for (u32 sg = 0; sg < nseg && first + sg < max_items; sg++) {
                mid_item it;
                it.gid = (u32)G; it.lo = lo[j] + sg * MID_LONG_SEG; it.hi = it.lo + MID_LONG_SEG < lo[j] + c[j] ? it.lo + MID_LONG_SEG : lo[j] + c[j]; it.first = first; it.lb = lb; it.nseg = nseg; it.pad0 = it.pad1 = 0;
                items[first + sg] = it;
            }
        }
    }
    // pass 2: place.  The cursor atomics of a vector's digits first, then the stores (the waits for the returned positions then overlap).
    u32 *dst = sorted + (u64)k * n;
#pragma unroll 1
    for (u32 i0 = tid; i0 < nvec; i0 += 4 * MID_SORT_THREADS) {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const u32 i = i0 + (u32)q * MID_SORT_THREADS; v[q] = row[i < nvec ? i : nvec - 1]; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 i = i0 + (u32)q * MID_SORT_THREADS;
            const bool live = i < nvec;
            const u32 x[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
            u32 pos[8], ent[8];
            bool put[8];
#pragma unroll
            for (int h = 0; h < 8; h++) {
                const u32 vv = (h & 1) ? x[h >> 1] >> 16 : x[h >> 1] & 0xffffu;
                const u32 dp = vv - K.lo_pos, dn = K.lo_neg - vv;
                const bool isp = dp < K.bps_count, isn = dn < K.bps_count;
                put[h] = live && vv <= K.vmax && (isp || isn);
                ent[h] = (8u * i + (u32)h) | (isp ? 0u : 0x80000000u);
                pos[h] = 0;
                if (put[h]) pos[h] = atomicAdd(&cur[isp ? dp : dn], 1u);
            }
#pragma unroll
            for (int h = 0; h < 8; h++) if (put[h]) dst[pos[h]] = ent[h];
        }
    }
}

// ---- accumulate ----------------------------------------------------------------------------------------------------------------------------------------------------
// over-long lists as a launch of their own (raw points: behind k_mid_acc<0> on the same stream; mid_long.h has the work loop)
template <int FMT>
__global__ void __launch_bounds__(256) k_mid_long(const u32 *__restrict__ recs, const u32 *__restrict__ sorted, u64 n, msm_geom g, u32 *__restrict__ buckets, u32 max_items,
                                                  const mid_item *__restrict__ items, const u32 *__restrict__ counters, u32 *__restrict__ seg_sums, u32 *__restrict__ long_done) {
    C25519_PRIO_LONG();
    mid_long_body<FMT>(recs, sorted, n, g, buckets, max_items, items, counters, seg_sums, long_done, blockIdx.x * 4u + (threadIdx.x >> 6), gridDim.x * 4u);
}
// one lane per bucket (in the order of perm).  No record is prefetched across the addition: 136 VGPRs, three waves per SIMD, which is what counts once the buckets
// outnumber the machine's lanes (from ~2^15 terms: the first version kept the next record in registers, 210 VGPRs, and took 131 us at 2^16 terms against 73 now);
// and below, where every bucket has its lane at once, a prefetching variant (176 VGPRs, two waves) measured level at every size -- a lane's list is a chain of 8 M
// additions at ~4 us each for a nearly lone wave, not of loads (profiles/r06_ab_mid_prefetch.txt; removed).
template <int FMT>
__device__ __forceinline__ void mid_acc_body(const u32 *__restrict__ recs, const u32 *__restrict__ sorted, const u32 *__restrict__ base, const u32 *__restrict__ perm, u64 count, u64 n,
                                             const msm_geom &g, u32 *__restrict__ buckets, u32 block) {
    const u64 tid = (u64)block * 256 + threadIdx.x;
    const bool in_range = tid < count;
    const u64 gid = in_range ? perm[tid] : 0;
    const int k = (int)(gid / g.half), b = (int)(gid % g.half);
    const u32 lo = base[(u64)k * (g.half + 1) + b], hi = base[(u64)k * (g.half + 1) + b + 1];
    const bool mine = in_range && hi - lo <= g.long_cap;
    const u32 *list = sorted + (u64)k * n;
    const u32 len = mine ? hi - lo : 0u;
    ge_p3 acc = ge_identity();
    u32 e = len > 0 ? list[lo] : 0u;
    u32 sgn = 0;      // (A/B arm C25519_MID_SIGN_LAZY: the digit's sign lazily on the accumulator, as in accum.hip.  Measured LEVEL here -- 8192 terms 0.166 - 0.173 against
                      //  0.167 - 0.169 ms, 2^18 0.530 - 0.542 against 0.522 - 0.535, profiles/r06_ab_lazy_sign.txt: a lane's list is a latency chain, and the negation sits ON it
                      //  where the selects on the record do not -- so the operand selection stays)
#pragma unroll 1
    for (u32 it = 0; it < len; it++) {
        mid_rec<FMT> cur;
        cur.load(recs, e & 0x7fffffffu);
#ifndef C25519_MID_SIGN_LAZY
        const bool neg = (e >> 31) != 0;
        if (it + 1 < len) e = list[lo + it + 1];
        acc = cur.add_to(acc, neg);
#else
        const u32 me = (u32)((int)e >> 31), flip = me ^ sgn;
        sgn = me;
        if (it + 1 < len) e = list[lo + it + 1];
        acc = cur.add_to_lazy(acc, flip);
#endif
        ge_pin(acc);
    }
#ifdef C25519_MID_SIGN_LAZY
    acc.X = fe_carry(feW(fe_cond_neg(acc.X, sgn))); acc.T = fe_carry(feW(fe_cond_neg(acc.T, sgn)));
#endif
    (void)sgn;
    if (mine) p40_store(buckets, gid, acc);
}
template <int FMT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_mid_acc(const u32 *__restrict__ recs, const u32 *__restrict__ sorted, const u32 *__restrict__ base, const u32 *__restrict__ perm, u64 count, u64 n,
                                                 msm_geom g, u32 *__restrict__ buckets) {
    mid_acc_body<FMT>(recs, sorted, base, perm, count, n, g, buckets, blockIdx.x);
}
// (r6, late) ... and with the over-long lists IN the launch: L.blocks blocks in front of the bucket lanes fold them (mid_long.h; the form accum.hip k_accumulate_long gave
// verify_batch).  This is what lets the cap on a lane's list come down to where it belongs: the accumulation of a mid-size call is not throughput, it is its LONGEST
// list -- a chain of dependent additions at 3 - 5 us each (2^17 terms: lists of up to 34 entries, 140 us, where the machine needs 100 for all 2.4 M additions; 2^18
// terms: 51 entries, 238 us against 180) -- and a list handed to a wave costs a ~27 us shuffle tree, beside the bucket lanes instead of behind them.
struct mid_long_args { const mid_item *items; const u32 *counters; u32 *seg_sums; u32 *long_done; u32 max_items, blocks; };
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_mid_acc_long(const u32 *__restrict__ recs, const u32 *__restrict__ sorted, const u32 *__restrict__ base, const u32 *__restrict__ perm, u64 count, u64 n,
                                                 msm_geom g, u32 *__restrict__ buckets, mid_long_args L) {
    if (blockIdx.x < L.blocks) {
        C25519_PRIO_LONG();
        mid_long_body<0>(recs, sorted, n, g, buckets, L.max_items, L.items, L.counters, L.seg_sums, L.long_done, blockIdx.x * 4u + (threadIdx.x >> 6), L.blocks * 4u);
        return;
    }
    mid_acc_body<0>(recs, sorted, base, perm, count, n, g, buckets, blockIdx.x - L.blocks);
}

// (r6, late) The same with the WAVE-COOPERATIVE GATHER of accum.hip k_accumulate, for the sizes where the buckets outnumber the machine's lanes (2^16 terms and up):
// k_mid_acc<0> at 2^18 terms is 66 % VALU-busy and its waves wait 1.6x as long as they issue (profiles/r06_mid_acc_pmc.txt) -- every lane fetches its own 160-byte
// record with ten 16-byte loads, 640 cache-line look-ups per addition and wave, and nothing is in flight during the addition (a register prefetch costs the third wave
// per SIMD: r06_ab_mid_prefetch.txt).  Here the records of addition i + 1 travel by DMA into LDS during addition i: pieces 0 .. 7 of a record by the eight lanes
// 8j .. 8j + 7 (eight instructions of eight lines each, the layout of k_accumulate: piece c of record r at position (c + r) mod 8 of the record's eight slots), pieces 8
// and 9 by lane pairs (two instructions of 32 lines), 128 look-ups per addition; every lane reads its own record back (ten ds_read_b128) before the next DMA is issued.
// 40 KB of LDS per block, three blocks per compute unit, the register budget of k_mid_acc.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_mid_acc_coop(const u32 *__restrict__ recs, const u32 *__restrict__ sorted, const u32 *__restrict__ base, const u32 *__restrict__ perm, u64 count, u64 n,
                                                 msm_geom g, u32 *__restrict__ buckets) {
    __shared__ uint4 stage[(256 / 64) * MID_REC_Q * 64];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
    const bool in_range = tid < count;                     // every lane of a wave keeps loading for the others
    const u64 gid = in_range ? perm[tid] : 0;
    const int k = (int)(gid / g.half), b = (int)(gid % g.half);
    const u32 lo = base[(u64)k * (g.half + 1) + b], hi = base[(u64)k * (g.half + 1) + b + 1];
    const bool mine = in_range && hi - lo <= g.long_cap;
    const u32 *list = sorted + (u64)k * n;
    const u32 lane = threadIdx.x & 63u;
    uint4 *slot_a = stage + (threadIdx.x >> 6) * (MID_REC_Q * 64), *slot_b = slot_a + 8 * 64;
    const uint4 *my_a = slot_a + lane * 8, *my_b = slot_b + lane * 2;
    const u32 sub = lane >> 3, coff = ((lane & 7u) - sub) & 7u;
    const u32 len = mine ? hi - lo : 0u;
    u32 wmax = len;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { u32 o = (u32)__shfl_xor((int)wmax, d, 64); wmax = o > wmax ? o : wmax; }
    ge_p3 acc = ge_identity();
    u32 e = 0, e1 = 0;                                   // entries of iterations it and it + 1 (a finished lane keeps a valid index)
    if (len > 0) e = list[lo];
    if (len > 1) e1 = list[lo + 1];
#define C25519_MID_COOP_ISSUE(ent)                                                                                              \
    {                                                                                                                           \
        _Pragma("unroll") for (int kk = 0; kk < 8; kk++) {                                                                     \
            const u32 idx = (u32)__shfl((int)(ent), (int)(8 * kk + sub), 64) & 0x7fffffffu;                                     \
            const uint4 *src = reinterpret_cast<const uint4 *>(recs) + (u64)MID_REC_Q * idx + coff;                             \
            __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(slot_a + kk * 64), 16, 0, 0);                        \
        }                                                                                                                       \
        _Pragma("unroll") for (int m = 0; m < 2; m++) {                                                                        \
            const u32 idx = (u32)__shfl((int)(ent), (int)(32 * m + (lane >> 1)), 64) & 0x7fffffffu;                             \
            const uint4 *src = reinterpret_cast<const uint4 *>(recs) + (u64)MID_REC_Q * idx + 8 + (lane & 1u);                  \
            __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(slot_b + m * 64), 16, 0, 0);                         \
        }                                                                                                                       \
    }
    if (wmax > 0) C25519_MID_COOP_ISSUE(e)
#pragma unroll 1
    for (u32 it = 0; it < wmax; it++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        mid_rec<0> cur;
#pragma unroll
        for (int c = 0; c < 8; c++) cur.q[c] = my_a[(c + lane) & 7u];
        cur.q[8] = my_b[0]; cur.q[9] = my_b[1];
        const bool neg = (e >> 31) != 0, active = it < len;
        const u32 e_next = e1;
        if (it + 2 < len) e1 = list[lo + it + 2];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (it + 1 < wmax) C25519_MID_COOP_ISSUE(e_next)
        if (active) acc = cur.add_to(acc, neg);
        e = e_next;
    }
#undef C25519_MID_COOP_ISSUE
    if (mine) p40_store(buckets, gid, acc);
}

}  // namespace c25519

// ---- host ----------------------------------------------------------------------------------------------------------------------------------------------------------------
// upper end of the path (A/B knob MSM_MID_MAX of the tuning build; 0 = off: the bucket pipeline from 12 288 terms as in round 5)
// upper ends of the path (A/B knobs of the tuning build; 0 = off: the bucket pipeline from 12 288 terms as in round 5): raw points up to 2^18 terms (200 000 terms 0.47 against 0.54 ms, 2^18 0.57 against 0.58, 400 000
// 0.78 against 0.71: profiles/r06_ab_mid_upper_end.txt); prepared records -- verify_batch's 2n + 1 terms, whose sort is a third of the bucket pipeline's call -- up to 2^18 + 1
uint64_t msm_mid_max() { static const uint64_t v = (uint64_t)C25519_KNOB_LL("MSM_MID_MAX", 1 << 18); return v; }
static uint64_t msm_mid_max_records() { static const uint64_t v = (uint64_t)C25519_KNOB_LL("MSM_MID_MAX_RECORDS", (1 << 18) + 1); return v; }
bool msm_mid_serves_terms(uint64_t n) { return n > verify_small_max() && n <= msm_mid_max_records(); }      // (verify_batch's prepared records; whatever width the caller is about to choose)
// (prepared records below msm_small_max() terms: only with a layout that is not the small path's -- verify_batch from verify_small_max() + 1 terms chooses one)
bool msm_mid_serves(uint64_t n, const msm_geom &g, bool prepared) {
    return (n > msm_small_max() || (prepared && n > verify_small_max() && g.half > 64)) && n <= (prepared ? msm_mid_max_records() : msm_mid_max()) && g.c >= 8 && g.c <= 16 && g.half >= 64 &&
           g.ngroups <= 1;
}

// The cap on a bucket lane's list (longer lists go to the waves of the long path).  A lane walks its list as a chain of dependent additions, so the accumulation lasts as
// long as the longest list below the cap; every list above it costs a wave a ~27 us shuffle tree, which is free beside the bucket lanes as long as there are a few
// hundred of them.  For uniform digits the list lengths of window k are Poisson(n / buckets_k) -- the windows of c - 1 and c - 2 bits have lists twice as long as the
// others -- so: the smallest cap (>= 8) that leaves an EXPECTED `target` lists above it, never above the rule it replaces (`upper`).  Skewed inputs simply have more
// long lists than expected; the long path takes whatever it is given.
static double poisson_tail(double lam, int cap) {           // P(X > cap), X ~ Poisson(lam)
    double term = std::exp(-lam), cdf = term;
    for (int i = 1; i <= cap; i++) { term *= lam / i; cdf += term; }
    return cdf >= 1.0 ? 0.0 : 1.0 - cdf;
}
// Measured (profiles/r06_ab_mid_cap.txt; device-resident calls, raw points, with the long lists inside the accumulation's launch): target 64 against the rule it replaces
// 12 288 terms 0.208 -> 0.197 ms, 2^14 0.225 -> 0.207, 2^15 0.250 -> 0.241, 2^16 0.294 -> 0.287, 2^17 0.378 -> 0.374, 2^18 0.541 -> 0.553 (the accumulation there is
// throughput: the old cap stays); target 256 the same up to 2^14 and level or worse above; 1024 worse everywhere (a long list costs its wave ten times the
// additions of the list).  verify_batch: 2^13 signatures 0.333 -> 0.312, level within +-5 us above: the old cap from 2^15 terms.
static uint32_t mid_pick_cap(uint64_t n, const msm_geom &g, uint32_t upper, bool prepared) {
    static const int target = C25519_KNOB("MID_LONG_TARGET", 64);       // A/B knob of the tuning build: 0 = the rule of the first mid path, max(48, 3 x mean)
    static const int force = C25519_KNOB("MID_LONG_TARGET_ALWAYS", 0);  // ... at every size
    if (target <= 0) return upper;
    if (!force && n >= (prepared ? (1ull << 15) : (1ull << 18))) return upper;
    for (uint32_t cap = 8; cap < upper; cap++) {
        double expect = 0;
        for (int k = 0; k + 1 < g.nwin; k++) {                // (the overflow window holds a handful of entries)
            const double buckets = (double)(1u << (k + 2 < g.nwin ? g.wid[k] - 1 : g.wid[k]));
            expect += buckets * poisson_tail((double)n / buckets, (int)cap);
        }
        if (expect <= (double)target) return cap;
    }
    return upper;
}

// The whole pass: digits (+ records), sort, accumulation, reduction; column sums (and, hdr != 0, the record header) to d_slot -- or, ctx->direct_seq != 0, the
// record published into the context's page-locked host slot.  src_fmt 0: raw 160-byte points at `points`; 1: affine Niels records at `points` (a decompression made them).
// hdr 0: the slot was initialised by k_slot_init and carries counters of its own (verify_batch); 1: this pass writes the header (MSM).
// ring (may be null): [0] / [1] bracket k_mid_acc, [2] end of the pass.
int32_t msm_mid_enqueue(c25519_ctx *ctx, const uint8_t *d_scalars, const void *points, int src_fmt, uint64_t n, const msm_geom &g_in, uint32_t *d_slot, int hdr, uint64_t terms, hipEvent_t *ring,
                        const mid_run *run) {
    // The cap beyond which a list leaves the bucket lanes for k_mid_long: max(48, 3 x mean) here (the bucket pipeline: max(192, 2.5 x mean), set for passes of millions of
    // terms).  A mid-size call is as long as its longest list: verify_batch of 2^15 signatures with 15-bit windows puts the top 8 bits of the 128-bit z_i into 256 buckets
    // of ~128 entries each -- just under 192 -- and k_accumulate walked them for 368 us where the call of 2^16 signatures (256 entries each: over the cap) took 85
    // (profiles/r06_timeline_mid_verify_2p15_long_cap.txt).  One wave per 256 entries handles such a list in ~30 us.
    msm_geom g = g_in;
    g.long_cap = mid_pick_cap(n, g, (u32)std::max<uint64_t>(48, 3 * (n / (uint64_t)g.half + 1)), src_fmt != 0);
    if (!msm_mid_serves(n, g, src_fmt != 0) || n >= (1ull << 31)) { ctx->err = "msm: internal error (mid path outside its range)"; return -(int32_t)hipErrorInvalidValue; }
    if (run && src_fmt == 0) { ctx->err = "msm: internal error (mid path: a stream override with raw points)"; return -(int32_t)hipErrorInvalidValue; }
    hipStream_t st = run && run->stream ? run->stream : ctx->stream;
    const uint64_t dstride = (n + 7) & ~(uint64_t)7;
    const uint64_t nb = (uint64_t)g.nwin * g.half;
    const int nseg = red_nseg(g.half);
    // slices per window: enough blocks for the machine (A/B knob MID_SORT_BLOCKS, 128) with 256 .. 4096 buckets each -- k_mid_sort has the reasoning
    static const int want_blocks = C25519_KNOB("MID_SORT_BLOCKS", 128);
    int bps = 0; while ((1 << bps) < g.half && (1 << bps) < MID_BPS_MAX) bps++;
    while (bps > 8 && g.nwin * (g.half >> bps) < want_blocks) bps--;
    const int SL = g.half >> bps;
    const unsigned nfront = div_up64(dstride, 256);
    const uint64_t entries = (uint64_t)g.nwin * n;
    const uint32_t max_long = (uint32_t)std::min<uint64_t>(nb, entries / g.long_cap + 1);
    const uint32_t max_items = (uint32_t)(entries / MID_LONG_SEG + max_long + 1);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t oD = carve((size_t)g.nwin * dstride * 2), oB = carve((size_t)g.nwin * (g.half + 1) * 4), oS = carve((size_t)g.nwin * n * 4), oK = carve(nb * 160), oPerm = carve(nb * 4), oT = carve(nb * 4);
    const size_t oSW = carve((size_t)g.nwin * nseg * 2 * 160), oLI = carve((size_t)max_items * sizeof(mid_item)), oLG = carve((size_t)max_long * 4), oLS = carve((size_t)max_items * 160);
    const size_t oBF = carve((size_t)nfront * 4);
    // small words zeroed by k_mid_front: [0] long items, [1] long buckets, [2] finished windows (k_reduce_b4pub), [64 .. 320) the bucket-order histogram, [320 .. 576) its
    // cursors, [576 .. 576 + max_long) finished segments per long bucket
    const int nzero = 576 + (int)max_long;
    const size_t oZ = carve((size_t)nzero * 4);
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_d, off))) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_d.p;
    uint16_t *D = (uint16_t *)(ws + oD);
    uint32_t *base = (uint32_t *)(ws + oB), *sorted = (uint32_t *)(ws + oS), *buckets = (uint32_t *)(ws + oK), *perm = (uint32_t *)(ws + oPerm), *SW = (uint32_t *)(ws + oSW), *totals = (uint32_t *)(ws + oT);
    mid_item *items = (mid_item *)(ws + oLI);
    uint32_t *lgids = (uint32_t *)(ws + oLG), *segs = (uint32_t *)(ws + oLS), *blockflags = (uint32_t *)(ws + oBF), *zw = (uint32_t *)(ws + oZ);
    const uint32_t *recs = (const uint32_t *)points;
    // raw points: up to MID_PROJ_MAX terms the records are PROJECTIVE (no inversion; 8 M additions in k_mid_acc<0>); above, the batched normaliser (msm.hip
    // k_prep_raw2) makes affine records on this stream WHILE the digits and the sort run on the second one, and the accumulation is the bucket pipeline's
    // k_accumulate (7 M mixed additions, wave-cooperative gathers): its 130 us at 2^18 terms hide behind the sort, and the accumulation of 2^18 terms is
    // throughput, not latency
    static const uint64_t proj_max = (uint64_t)C25519_KNOB_LL("MID_PROJ_MAX", 1 << 18);      // (profiles/r06_ab_mid_knobs.txt: 2^16 terms 0.318 against 0.371 ms, 2^17 0.388 against 0.440; r06_ab_mid_upper_end.txt: 2^18 0.572 against 0.618)
    const bool proj = src_fmt == 0 && n <= proj_max, norm = src_fmt == 0 && !proj;
    hipStream_t ss = st;                                            // the stream of the digits and the sort
    if (src_fmt == 0 && (r = ctx_reserve(ctx, ctx->tmp_e, n * 160 + 256))) return r;
    if (norm) {
        ss = ctx->aux;
        HIPCHK(hipEventRecord(ctx->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(ss, ctx->ev_fork, 0));
        if ((r = prep_points(ctx, (const uint8_t *)points, n, C25519_FMT_RAW160, (uint32_t *)ctx->tmp_e.p, 0, nullptr))) return r;
        recs = (const uint32_t *)ctx->tmp_e.p;
    }
    if (proj) {
        recs = (const uint32_t *)ctx->tmp_e.p;
        hipLaunchKernelGGL(k_mid_front<0>, dim3(nfront), dim3(256), 0, ss, d_scalars, (const uint8_t *)points, n, g, D, dstride, (uint32_t *)ctx->tmp_e.p, zw, nzero, blockflags);
    } else hipLaunchKernelGGL(k_mid_front<1>, dim3(nfront), dim3(256), 0, ss, d_scalars, (const uint8_t *)nullptr, n, g, D, dstride, (uint32_t *)nullptr, zw, nzero, blockflags);
    hipLaunchKernelGGL(k_mid_sort, dim3(g.nwin, SL), dim3(MID_SORT_THREADS), 0, ss, D, n, dstride, g, bps, sorted, base, totals, zw + 64, max_items, items, zw, lgids);
    launch_order_place(totals, nb, zw + 64, zw + 320, perm, g, ss);
    if (norm) {
        HIPCHK(hipEventRecord(ctx->ev_sort, ss));
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_sort, 0));
    }
    if (ring) HIPCHK(hipEventRecord(ring[0], st));
    const unsigned nacc = div_up64(nb, 256), nlong = 256;      // (one wave per item, four per block: 1024 items in flight)
    // Over-long lists.  Prepared records (verify_batch: the top bits of the 128-bit z_i fill a few buckets with n / 2 .. n / 256 terms each -- a ~50 us chain of a few
    // dozen waves): blocks IN FRONT of the bucket lanes of the same launch (accum.hip k_accumulate_long), which skip those buckets.  (Until the end of round 6 a launch
    // of its own on the second stream beside the accumulation: the two cross-stream hand-overs cost 7 + 12 us of a 460 us call, profiles/r06_timeline_mid_verify_2p14.txt;
    // A/B knob MID_LONG_BESIDE of the tuning build.)  Raw points (random scalars have none: the kernel finds an empty work list): behind the accumulation on this stream.
    // (2^13 .. 2^15 signatures: -12 .. -25 us; 2^16: level; 2^17: +12 us, the launch of their own stays there -- profiles/r06_ab_verify_mid_launches.txt)
    static const int long_beside = C25519_KNOB("MID_LONG_BESIDE", 0);
    const bool fused = src_fmt != 0 && !long_beside && n <= (1ull << 17) + 1, beside = src_fmt != 0 && !fused;
    hipStream_t sl = beside ? (st == ctx->aux ? ctx->stream : ctx->aux) : st;
    // the records of a pass that runs on the stream of its scalars: made on the other stream (and signed here, behind them)
    if (run && run->recs_ready) HIPCHK(hipStreamWaitEvent(st, run->recs_ready, 0));
    if (run && run->sign_z16 && run->sign_count) launch_apply_sign((uint32_t *)recs, run->sign_first, run->sign_z16, run->sign_count, st);
    if (beside) {
        HIPCHK(hipEventRecord(ctx->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(sl, ctx->ev_fork, 0));
        hipLaunchKernelGGL(k_mid_long<1>, dim3(nlong), dim3(256), 0, sl, recs, sorted, n, g, buckets, max_items, items, zw, segs, zw + 576);
        HIPCHK(hipEventRecord(ctx->ev_join, sl));
    }
    if (proj) {
        ctx->kname[0] = "c25519::k_mid_acc<0> (mid path: one lane per bucket, 8 M additions on projective Niels records)";
        // from 2^16 terms (A/B knob MID_COOP_MIN of the tuning build; 0 = never) the cooperative gather
        // (the cooperative gather: measured level with the separate launch, profiles/r06_ab_mid_coop.txt; A/B knob MID_COOP_MIN of the tuning build, 0 = never)
        static const uint64_t coop_min = (uint64_t)C25519_KNOB_LL("MID_COOP_MIN", 0);
        static const int raw_fused = C25519_KNOB("MID_RAW_FUSED", 1);      // A/B knob: 0 = k_mid_long behind the accumulation on the same stream
        if (coop_min && n >= coop_min) {
            hipLaunchKernelGGL(k_mid_acc_coop, dim3(nacc), dim3(256), 0, st, recs, sorted, base, perm, nb, n, g, buckets);
            hipLaunchKernelGGL(k_mid_long<0>, dim3(nlong), dim3(256), 0, st, recs, sorted, n, g, buckets, max_items, items, zw, segs, zw + 576);
        } else if (raw_fused) {
            ctx->kname[0] = "c25519::k_mid_acc_long (mid path: one lane per bucket, 8 M additions on projective Niels records; over-long lists in front)";
            const mid_long_args L = {items, zw, segs, zw + 576, max_items, 128u};
            hipLaunchKernelGGL(k_mid_acc_long, dim3(nacc + L.blocks), dim3(256), 0, st, recs, sorted, base, perm, nb, n, g, buckets, L);
        } else {
            hipLaunchKernelGGL(k_mid_acc<0>, dim3(nacc), dim3(256), 0, st, recs, sorted, base, perm, nb, n, g, buckets);
            hipLaunchKernelGGL(k_mid_long<0>, dim3(nlong), dim3(256), 0, st, recs, sorted, n, g, buckets, max_items, items, zw, segs, zw + 576);
        }
    } else if (fused) {
        ctx->kname[0] = launch_accumulate_long(recs, sorted, base, perm, nb, n, g, buckets, items, zw, segs, zw + 576, max_items, 64, st);
    } else {
        ctx->kname[0] = launch_accumulate(recs, sorted, base, perm, nb, n, g, buckets, 0, st);      // affine records: the bucket pipeline's accumulation (accum.hip)
        if (!beside) hipLaunchKernelGGL(k_mid_long<1>, dim3(nlong), dim3(256), 0, st, recs, sorted, n, g, buckets, max_items, items, zw, segs, zw + 576);
    }
    if (beside) HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
    if (ring) { HIPCHK(hipEventRecord(ctx->ev_acc, st)); HIPCHK(hipEventRecord(ring[1], st)); }
    reduce_publish pub = {0, nullptr, 0, (uint32_t)terms, (uint32_t)(terms >> 32), (uint32_t)g.c, hdr, hdr ? nullptr : slot_flags(d_slot)};
    uint32_t *out = d_slot;
    if (ctx->direct_seq) {
        out = ctx->hd_msm + (size_t)C25519_MAX_SLOTS * C25519_SLOT_U32;
        pub.on = 1; pub.host_flag = ctx->hd_msm + (size_t)(C25519_MAX_SLOTS + 1) * C25519_SLOT_U32; pub.seq = ctx->direct_seq;
        static const int lose_every = C25519_KNOB("FAULT_LOSE_PUBLICATION", 0);      // (tuning build only: small.hip has the note)
        const uint64_t nth = ++ctx->counters[C25519_CTR_PUBLISH_DIRECT];
        if (lose_every > 0 && nth % (uint64_t)(lose_every > 0 ? lose_every : 1) == 0) pub.seq ^= 0x40000000u;
    }
    launch_bucket_reduce_pub(buckets, g, nseg, SW, out, blockflags, (int)nfront, zw + 2, pub, st);
    HIPCHK(hipGetLastError());
    if (ring) HIPCHK(hipEventRecord(ring[2], st));
    if (st != ctx->stream) {                                   // the context's main stream continues behind the pass (nothing of this call waits for that)
        HIPCHK(hipEventRecord(ctx->ev_join, st));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
    return C25519_OK;
}
