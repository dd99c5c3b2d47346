// Bucket accumulation of the Pippenger MSM (pippenger.rs:122-136 as gather lists): one lane per (window, bucket).
// Its own translation unit: the field arithmetic here is the chained-carry form with the three + four independent
// products of a mixed addition issued column by column in LOCKSTEP (fe26x.h) -- 855 quarter-rate instructions per addition
// against 954 of the ten-column form.  (Round 2 also built the ten-column and the one-product-after-another chained forms
// and four other gather schedules as A/B arms: k_accumulate 1.13 ms per 2^21 terms in this form against 1.17 - 1.19;
// DESIGN.md section 4 keeps the numbers.)
#include <hip/hip_runtime.h>
#include <stdlib.h>
#define C25519_CHAIN 1
#include "devio.h"
#include "msm_internal.h"
#include "fe26x.h"
#include "msm_sort.h"
#include "mid_long.h"

namespace c25519 {

#if defined(__HIP_DEVICE_COMPILE__)
#define ge_madd_acc ge_madd_signed_p3_lockstep
#else
#define ge_madd_acc ge_madd_signed_p3
#endif

// Registers: 168 VGPRs (capped: 170 unconstrained, two more than three waves per SIMD allow; no scratch) -- accumulator
// point 40, the prefetched record 32, column sums, operands and pre-scaled limbs.  Three waves per SIMD are what saturates
// the multiplier (a lone wave issues one v_mad_u64_u32 per 11.7 cycles at best); measured on the ten-column form with
// amdgpu_waves_per_eu budgets: 128 VGPRs (4 waves, 10 scratch accesses per addition) 1.73 ms, 96 (5 waves, 32) 3.6 ms,
// 80 (6 waves, 105) 7.1 ms against 1.23 ms -- occupancy has nothing to give and any spill costs more than it hides.
//
// WAVE-COOPERATIVE GATHER.  A lane that fetches its own 128-byte record with eight 16-byte loads makes the
// texture/L1 path look up 64 different cache lines per instruction, 512 per addition and wave -- and that path,
// not the multiplier, is what k_accumulate shares with the normaliser and the sort of the next pass.  Here the
// eight lanes 8j..8j+7 fetch the eight 16-byte pieces of ONE record per instruction (8 lines per instruction, 64
// per addition), straight into LDS (global_load_lds_dwordx4: destination = wave base + 16 * lane), and every lane
// then reads its own record back.  Piece c of record r sits at position (c + r) mod 8 of the record's 128 bytes,
// so that the read-back of a piece touches all 32 banks once per 8 lanes.  The records of addition i+1 are in
// flight during addition i; the wave's own vmcnt(0) orders DMA -> ds_read, lgkmcnt(0) orders ds_read -> next DMA.
#ifndef C25519_ACC_WAVES
#define C25519_ACC_WAVES 3       // A/B arm (profiles/r04_ab_accumulate_occupancy.txt): 2 leaves a third of every SIMD's registers to other kernels
#endif
// (the body: `block` is the block's index among the bucket blocks, `stage` the block's 32 KB of LDS)
__device__ __forceinline__ void accumulate_body(const u32 *__restrict__ pts, const u32 *__restrict__ sorted, const u32 *__restrict__ base,
             const u32 *__restrict__ perm, u64 count, u64 n, const msm_geom &g, u32 *__restrict__ buckets, int cont, u32 block, uint4 *stage) {
    const u64 tid = (u64)block * blockDim.x + threadIdx.x;
    const bool in_range = tid < count;                     // every lane of a wave keeps loading for the others
    const u64 gid = in_range ? perm[tid] : 0;
    const int k = (int)(gid / g.half), b = (int)(gid % g.half);
    const u32 lo = base[(u64)k * (g.half + 1) + b], hi = base[(u64)k * (g.half + 1) + b + 1];
    const bool mine = in_range && hi - lo <= g.long_cap;   // long lists belong to k_long_segments
    const u32 *list = sorted + (u64)k * n;
    // cont: the bucket sums of the previous pass on this stream set are the starting point (one bucket reduction per call
    // instead of one per pass: msm.hip msm_record_enqueue)
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    const u32 lane = threadIdx.x & 63u;
    // (r6, last) the wave's slot from a SCALAR wave index: the LDS destination of every gather (M0) is then scalar arithmetic -- it was a v_or + v_readfirstlane per gather
#ifdef C25519_ACC_GATHER64      // A/B arm (tools/build_variant.sh): the addressing until the end of round 6
    uint4 *wave_slot = stage + (threadIdx.x >> 6) * (8 * 64);
#else
    uint4 *wave_slot = stage + (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (8 * 64);
#endif
    const uint4 *my_rec = wave_slot + lane * 8;
    const u32 sub = lane >> 3, coff = ((lane & 7u) - sub) & 7u;
    const u32 len = mine ? hi - lo : 0u;
    u32 wmax = len;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { u32 o = (u32)__shfl_xor((int)wmax, d, 64); wmax = o > wmax ? o : wmax; }
    // a wave whose lists are all empty leaves continued bucket sums where they are (the buckets a narrow window never uses -- a fifth of a
    // 17-bit layout -- and whatever a short pass did not touch): no 160-byte load and store per lane
    if (cont && wmax == 0) return;
    ge_p3 acc = (cont && mine) ? p40_load(buckets, gid) : ge_identity();
    u32 e = 0, e1 = 0;                                   // entries of iterations it and it + 1 (a finished lane keeps a valid index)
    u32 sgn = 0;                                         // (lazy sign: see the addition below)
    if (len > 0) e = list[lo];
    if (len > 1) e1 = list[lo + 1];
#ifdef C25519_ACC_GATHER64
#define C25519_COOP_ISSUE(ent)                                                                                                  \
    _Pragma("unroll") for (int kk = 0; kk < 8; kk++) {                                                                         \
        const u32 idx = (u32)__shfl((int)(ent), (int)(8 * kk + sub), 64) & 0x7fffffffu;                                         \
        const uint4 *src = reinterpret_cast<const uint4 *>(pts) + PTS_Q * (u64)idx + coff;                                      \
        __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(wave_slot + kk * 64), 16, 0, 0);                         \
    }
#else
#define C25519_COOP_ISSUE(ent)                                                                                                  \
    _Pragma("unroll") for (int kk = 0; kk < 8; kk++) {                                                                         \
        /* the piece's BYTE offset in 32 bits (a pass has at most 2^25 records: launch_accumulate checks), added to the scalar base by the load itself:   \
           one v_lshl_or_b32 per gather where the 64-bit address took a mask, a 64-bit shift and a 64-bit add (the shift drops the sign bit of the entry) */ \
        const u32 boff = ((u32)__shfl((int)(ent), (int)(8 * kk + sub), 64) << 7) | (coff << 4);                                 \
        const char *src = reinterpret_cast<const char *>(pts) + boff;                                                            \
        __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(wave_slot + kk * 64), 16, 0, 0);                         \
    }
#endif
    if (wmax > 0) { C25519_COOP_ISSUE(e) }
#pragma unroll 1
    for (u32 it = 0; it < wmax; it++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 q[PTS_Q];
#pragma unroll
        for (int c = 0; c < PTS_Q; c++) q[c] = my_rec[(c + lane) & 7u];
        const bool neg = (e >> 31) != 0, active = it < len;
        const u32 e_next = e1;
        if (it + 2 < len) e1 = list[lo + it + 2];          // (a non-temporal load here -- to keep the lists out of the MALL -- loses the line reuse of a lane's consecutive entries: k_accumulate 1.12 against 1.00 ms)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (it + 1 < wmax) { C25519_COOP_ISSUE(e_next) }
        // (the loop counter is wave-uniform here, so the first entry of a list could be CONVERTED -- 1 M instead of 7 M --
        //  behind a uniform branch: 76 scratch accesses in the addition at the 168-register budget.  Round 4 PEELED it in front of the loop
        //  instead (kernel-uniform on `cont`; 168 VGPRs, no scratch): level from 2^12 to 2^20 terms and at 2^24, 1.978 against 1.907 ms at 2^21
        //  (k_accumulate 1.21 against 1.12 ms), profiles/r04_ab_first_entry.txt -- not adopted)
#if defined(C25519_ACC_SIGN_SELECT) || !defined(__HIP_DEVICE_COMPILE__)      // A/B arm (tools/build_variant.sh): the sign by operand selection, rounds 2 - 6
        if (active) acc = ge_madd_acc(acc, pts_from_q(q), neg);
#else
        // (r6, last) the sign LAZILY on the accumulator (fe26x.h ge_madd_lazy_p3_lockstep): sgn = all ones while the stored point is MINUS the bucket sum
        if (active) {
            const u32 me = (u32)((int)e >> 31), flip = me ^ sgn;
            sgn = me;
            acc = ge_madd_lazy_p3_lockstep(acc, pts_from_q(q), flip);
        }
#endif
        e = e_next;
    }
#undef C25519_COOP_ISSUE
    if (!mine) return;
#if !defined(C25519_ACC_SIGN_SELECT) && defined(__HIP_DEVICE_COMPILE__)
    acc.X = fe_carry(feW(fe_cond_neg(acc.X, sgn))); acc.T = fe_carry(feW(fe_cond_neg(acc.T, sgn)));      // back to the true sum (tight limbs: p40_store's format)
#endif
    p40_store(buckets, gid, acc);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C25519_ACC_WAVES, C25519_ACC_WAVES)))
k_accumulate(const u32 *__restrict__ pts, const u32 *__restrict__ sorted, const u32 *__restrict__ base,
             const u32 *__restrict__ perm, u64 count, u64 n, msm_geom g, u32 *__restrict__ buckets, int cont) {
    __shared__ uint4 stage[(256 / 64) * 8 * 64];
    accumulate_body(pts, sorted, base, perm, count, n, g, buckets, cont, blockIdx.x, stage);
}
// (r6) The mid path's accumulation of prepared records (mid.hip: verify_batch of 2^13 .. 2^17 signatures): the same bucket lanes, and IN FRONT of them L.blocks blocks
// whose waves fold the over-long lists (mid_long.h).  One launch instead of k_accumulate on the main stream with k_mid_long beside it on the second: the two
// cross-stream hand-overs around that pair were 7 + 12 us of a 460 us call (profiles/r06_timeline_mid_verify_2p14.txt), and the long lists -- the call's longest
// chains -- are dispatched first.
struct acc_long_args { const mid_item *items; const u32 *counters; u32 *seg_sums; u32 *long_done; u32 max_items, blocks; };
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C25519_ACC_WAVES, C25519_ACC_WAVES)))
k_accumulate_long(const u32 *__restrict__ pts, const u32 *__restrict__ sorted, const u32 *__restrict__ base,
                  const u32 *__restrict__ perm, u64 count, u64 n, msm_geom g, u32 *__restrict__ buckets, acc_long_args L) {
    __shared__ uint4 stage[(256 / 64) * 8 * 64];
    if (blockIdx.x < L.blocks) {
        C25519_PRIO_LONG();
        mid_long_body<1>(pts, sorted, n, g, buckets, L.max_items, L.items, L.counters, L.seg_sums, L.long_done, blockIdx.x * 4u + (threadIdx.x >> 6), L.blocks * 4u);
        return;
    }
    accumulate_body(pts, sorted, base, perm, count, n, g, buckets, 0, blockIdx.x - L.blocks, stage);
}

}  // namespace c25519

const char *launch_accumulate(const uint32_t *pts, const uint32_t *sorted, const uint32_t *base, const uint32_t *perm, uint64_t count, uint64_t n, const c25519::msm_geom &g, uint32_t *buckets, int cont, hipStream_t st) {
    using namespace c25519;
    hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, pts, sorted, base, perm, count, n, g, buckets, cont);
    return "c25519::k_accumulate (lockstep field products, wave-cooperative gather)";
}
const char *launch_accumulate_long(const uint32_t *pts, const uint32_t *sorted, const uint32_t *base, const uint32_t *perm, uint64_t count, uint64_t n, const c25519::msm_geom &g, uint32_t *buckets,
                                   const void *items, const uint32_t *counters, uint32_t *seg_sums, uint32_t *long_done, uint32_t max_items, uint32_t long_blocks, hipStream_t st) {
    using namespace c25519;
    acc_long_args L = {(const mid_item *)items, counters, seg_sums, long_done, max_items, long_blocks};
    hipLaunchKernelGGL(k_accumulate_long, dim3((unsigned)((count + 255) / 256) + long_blocks), dim3(256), 0, st, pts, sorted, base, perm, count, n, g, buckets, L);
    return "c25519::k_accumulate_long (mid path: bucket lanes of k_accumulate behind the blocks of the over-long lists)";
}
