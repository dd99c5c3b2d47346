// Bucket accumulation of the Pippenger MSM (pippenger.rs:122-136 as gather lists): one lane per (window, bucket).
// Its own translation unit because it is built three times, once per form of the field arithmetic (C25519_ACC_CHAIN picks):
//   accum0.o  -DC25519_CHAIN=0                      ten independent column sums per product (fe26.h)
//   accum1.o  -DC25519_CHAIN=1                      chained carries, one product after another
//   accum2.o  -DC25519_CHAIN=1 -DC25519_LOCKSTEP=1  chained carries, the three + four independent products of a mixed
//             addition issued column by column in lockstep (fe26x.h) -- the DEFAULT: 855 quarter-rate instructions per
//             addition against 954 of the ten-column form, k_accumulate 1.13 against 1.18 ms per 2^21 terms
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "devio.h"
#include "msm_internal.h"
#if defined(C25519_LOCKSTEP)
#include "fe26x.h"
#endif

namespace c25519 {
namespace ACCUM_NS {

#if defined(C25519_LOCKSTEP) && defined(__HIP_DEVICE_COMPILE__)
#define ge_madd_acc ge_madd_signed_p3_lockstep
#define ACCUM_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))   // 170 VGPRs unconstrained, two more than three waves allow
#else
#define ge_madd_acc ge_madd_signed_p3
#define ACCUM_ATTR
#endif

// Registers: 151 VGPRs in the ten-column form, 168 (capped, no scratch) in the lockstep form, i.e. THREE waves per SIMD
// either way (rocprofv3 prints the count halved) -- accumulator point 40, the prefetched record 32, column sums,
// operands and pre-scaled limbs.  Measured in round 2 on the ten-column form with
// amdgpu_waves_per_eu budgets: 128 VGPRs (4 waves, 10 scratch accesses per addition) 1.73 ms, 96 (5 waves, 32) 3.6 ms,
// 80 (6 waves, 105) 7.1 ms against 1.23 ms for this form -- at three waves the kernel already issues at 93 % of its
// instruction bound, so occupancy has nothing to give and any spill costs more than it hides.
template <int PIPE>   // PIPE 0: plain loop; 1: next index prefetched; 2: next index and next point prefetched; 3: point i+1 and index i+2; 4: wave-cooperative gather through LDS
__global__ void __launch_bounds__(256) ACCUM_ATTR k_accumulate(const u32 *__restrict__ pts, const u32 *__restrict__ sorted, const u32 *__restrict__ base,
                                                    const u32 *__restrict__ perm, u64 count, u64 n, msm_geom g, u32 *__restrict__ buckets) {
    u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (PIPE != 4 && tid >= count) return;
    const bool in_range = tid < count;                     // (PIPE 4: every lane of a wave keeps loading for the others)
    const u64 gid = in_range ? perm[tid] : 0;
    int k = (int)(gid / g.half), b = (int)(gid % g.half);
    u32 lo = base[(u64)k * (g.half + 1) + b], hi = base[(u64)k * (g.half + 1) + b + 1];
    const bool mine = in_range && hi - lo <= g.long_cap;   // long lists belong to k_long_segments
    if (PIPE != 4 && !mine) return;
    const u32 *list = sorted + (u64)k * n;
    ge_p3 acc = ge_identity();
    if (PIPE == 0) {
#pragma unroll 1
        for (u32 i = lo; i < hi; i++) {
            u32 e = list[i];
            acc = ge_madd_acc(acc, pts_load(pts, e & 0x7fffffffu), (e >> 31) != 0);
        }
    } else if (PIPE == 1) {
        u32 e_next = lo < hi ? list[lo] : 0u;
#pragma unroll 1
        for (u32 i = lo; i < hi; i++) {
            u32 e = e_next;
            if (i + 1 < hi) e_next = list[i + 1];
            acc = ge_madd_acc(acc, pts_load(pts, e & 0x7fffffffu), (e >> 31) != 0);
        }
    } else if (PIPE == 3) {
        // point i+1 AND index i+2 in flight during addition i: the gather of the next point never waits for its index.
        // (Peeling the first entry of a list -- a conversion, 1 M, instead of an addition to the identity, 7 M: 1.5 % of
        //  the multiplications at 64 entries per bucket -- puts two scratch accesses into the loop at the 168-register
        //  budget of the lockstep form, which costs more than it saves.)
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
            acc = ge_madd_acc(acc, A, neg);
        }
    } else if (PIPE == 4) {
        // WAVE-COOPERATIVE GATHER.  A lane that fetches its own 128-byte record with eight 16-byte loads makes the
        // texture/L1 path look up 64 different cache lines per instruction, 512 per addition and wave -- and that path,
        // not the multiplier, is what k_accumulate shares with the normaliser and the sort of the next pass.  Here the
        // eight lanes 8j..8j+7 fetch the eight 16-byte pieces of ONE record per instruction (8 lines per instruction, 64
        // per addition), straight into LDS (global_load_lds_dwordx4: destination = wave base + 16 * lane), and every lane
        // then reads its own record back.  Piece c of record r sits at position (c + r) mod 8 of the record's 128 bytes,
        // so that the read-back of a piece touches all 32 banks once per 8 lanes.  The records of addition i+1 are in
        // flight during addition i; the wave's own vmcnt(0) orders DMA -> ds_read, lgkmcnt(0) orders ds_read -> next DMA.
        __shared__ uint4 stage[(256 / 64) * 8 * 64];
        typedef __attribute__((address_space(3))) void lds_void;
        typedef const __attribute__((address_space(1))) void gbl_void;
        const u32 lane = threadIdx.x & 63u;
        uint4 *wave_slot = stage + (threadIdx.x >> 6) * (8 * 64);
        const uint4 *my_rec = wave_slot + lane * 8;
        const u32 sub = lane >> 3, coff = ((lane & 7u) - sub) & 7u;
        const u32 len = mine ? hi - lo : 0u;
        u32 wmax = len;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { u32 o = (u32)__shfl_xor((int)wmax, d, 64); wmax = o > wmax ? o : wmax; }
        u32 e = 0, e1 = 0;                                   // entries of iterations it and it + 1 (a finished lane keeps a valid index)
        if (len > 0) e = list[lo];
        if (len > 1) e1 = list[lo + 1];
#define C25519_COOP_ISSUE(ent)                                                                                                  \
        _Pragma("unroll") for (int kk = 0; kk < 8; kk++) {                                                                     \
            const u32 idx = (u32)__shfl((int)(ent), (int)(8 * kk + sub), 64) & 0x7fffffffu;                                     \
            const uint4 *src = reinterpret_cast<const uint4 *>(pts) + PTS_Q * (u64)idx + coff;                                  \
            __builtin_amdgcn_global_load_lds((gbl_void *)src, (lds_void *)(wave_slot + kk * 64), 16, 0, 0);                     \
        }
        if (wmax > 0) { C25519_COOP_ISSUE(e) }
#pragma unroll 1
        for (u32 it = 0; it < wmax; it++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint4 q[PTS_Q];
#pragma unroll
            for (int c = 0; c < PTS_Q; c++) q[c] = my_rec[(c + lane) & 7u];
            const bool neg = (e >> 31) != 0, active = it < len;
            const u32 e_next = e1;
            if (it + 2 < len) e1 = list[lo + it + 2];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (it + 1 < wmax) { C25519_COOP_ISSUE(e_next) }
            // (the loop counter is wave-uniform here, so the first entry of a list could be CONVERTED -- 1 M instead of 7 M --
            //  behind a uniform branch: 76 scratch accesses in the addition at the 168-register budget)
            if (active) acc = ge_madd_acc(acc, pts_from_q(q), neg);
            e = e_next;
        }
#undef C25519_COOP_ISSUE
        if (!mine) return;
    } else {
        uint4 q[PTS_Q];
        u32 e = 0;
        if (lo < hi) { e = list[lo]; const uint4 *src = reinterpret_cast<const uint4 *>(pts) + PTS_Q * (u64)(e & 0x7fffffffu); for (int j = 0; j < PTS_Q; j++) q[j] = src[j]; }
#pragma unroll 1
        for (u32 i = lo; i < hi; i++) {
            const ge_aniels A = pts_from_q(q);
            const bool neg = (e >> 31) != 0;
            if (i + 1 < hi) { e = list[i + 1]; const uint4 *src = reinterpret_cast<const uint4 *>(pts) + PTS_Q * (u64)(e & 0x7fffffffu); for (int j = 0; j < PTS_Q; j++) q[j] = src[j]; }
            acc = ge_madd_acc(acc, A, neg);
        }
    }
    p40_store(buckets, gid, acc);
}

}  // namespace ACCUM_NS
}  // namespace c25519

void ACCUM_LAUNCH(int pipe, const uint32_t *pts, const uint32_t *sorted, const uint32_t *base, const uint32_t *perm, uint64_t count, uint64_t n, const c25519::msm_geom &g, uint32_t *buckets, hipStream_t st) {
    using namespace c25519;
    using namespace c25519::ACCUM_NS;
    const dim3 grid((unsigned)((count + 255) / 256)), blk(256);
    static const unsigned lds = [] { const char *e = getenv("C25519_ACC_LDS"); return e ? (unsigned)atoi(e) : 0u; }();   // A/B knob: LDS reservation = occupancy cap
    if (pipe == 0) hipLaunchKernelGGL(k_accumulate<0>, grid, blk, lds, st, pts, sorted, base, perm, count, n, g, buckets);
    else if (pipe == 1) hipLaunchKernelGGL(k_accumulate<1>, grid, blk, lds, st, pts, sorted, base, perm, count, n, g, buckets);
    else if (pipe == 3) hipLaunchKernelGGL(k_accumulate<3>, grid, blk, lds, st, pts, sorted, base, perm, count, n, g, buckets);
    else if (pipe == 4) hipLaunchKernelGGL(k_accumulate<4>, grid, blk, lds, st, pts, sorted, base, perm, count, n, g, buckets);
    else hipLaunchKernelGGL(k_accumulate<2>, grid, blk, lds, st, pts, sorted, base, perm, count, n, g, buckets);
}
