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
template <int PIPE>   // PIPE 0: plain loop; 1: next index prefetched; 2: next index and next point prefetched; 3: point i+1 and index i+2
__global__ void __launch_bounds__(256) ACCUM_ATTR k_accumulate(const u32 *__restrict__ pts, const u32 *__restrict__ sorted, const u32 *__restrict__ base,
                                                    const u32 *__restrict__ perm, u64 count, u64 n, msm_geom g, u32 *__restrict__ buckets) {
    u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= count) return;
    const u64 gid = perm[tid];
    int k = (int)(gid / g.half), b = (int)(gid % g.half);
    u32 lo = base[(u64)k * (g.half + 1) + b], hi = base[(u64)k * (g.half + 1) + b + 1];
    if (hi - lo > g.long_cap) return;
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
    else hipLaunchKernelGGL(k_accumulate<2>, grid, blk, lds, st, pts, sorted, base, perm, count, n, g, buckets);
}
