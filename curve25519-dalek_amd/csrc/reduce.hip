// Bucket reduction col_k = sum_b (b + 1) B_b (pippenger.rs:146-151) with FOUR WAVES PER POINT OPERATION.
//
// The reduction is a latency chain, not a throughput problem: 1.1 M complete additions per 2^21-term pass are 35 us of multiplier time
// on this GPU, but as a running sum they are ~50 dependent point operations, and a lone wave issues a v_mad_u64_u32 every 6.4 cycles when it is
// independent of its predecessors and every 14 when it is not (profiles/r04_instruction_rates.txt) -- 9 M x 100 products per addition make
// ~4.4 us per link of the chain, 0.25 ms for rounds 2-3's k_reduce_a / k_reduce_b (one lane per point, msm.hip).  Nothing hides beside it: it
// is the tail of every call (and of every verify_batch).
//
// The reference's own answer to "one addition is too slow" is its parallel formulas (docs/parallel-formulas.md:106-213, the AVX2 backend:
// the four products of each half of an addition are independent).  Here the four lanes are four WAVES of a block: the 64 lanes of a wave
// still own 64 different points (so the wave-wide scans of the running sum keep their shape: "shuffles" become LDS indexing), the points
// live in LDS, and every addition is
//     stage 1   wave 0: (Y1-X1)(Y2-X2)   wave 1: (Y1+X1)(Y2+X2)   wave 2: T1 T2 2d   wave 3: 2 Z1 Z2      -> LDS, barrier
//     stage 2   wave 0: X3 = (B-A)(D-C)  wave 1: Y3 = (D+C)(B+A)  wave 2: T3 = (B-A)(B+A)  wave 3: Z3 = (D-C)(D+C)   -> LDS, barrier
// i.e. 2-3 multiplications deep instead of 9 (a doubling: one squaring and one multiplication deep instead of 7-8).  Same formulas, same
// results as ge_add / ge_dbl (ge26.h) -- the operand classes (tight / loose / wide) of every product are those of ge_add_cached and
// ge_p1p1_to_p3.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string>
// chained-carry products (fe26.h): 92 VGPRs instead of 172 with the ten-column form -- five blocks per compute unit, i.e. all 1088 segment
// blocks of a 2^21-term pass resident at once (with two per CU the chain would run twice)
#ifndef C25519_CHAIN
#define C25519_CHAIN 1
#endif
#include "../../include/c25519_hip.h"
#include "devio.h"
#include "ctx.h"
#include "msm_internal.h"
#include "msm_sort.h"

using namespace c25519;

namespace c25519 {

// LDS point array of 64 points (X 0-9, Y 10-19, Z 20-29, T 30-39).  Round 5 measured the alternative layout -- the 40 words of logical lane l at arr[42 l ..], a
// coordinate as five ds_read_b64 / ds_write_b64 (twice the bytes per LDS cycle; stride 42: 42 d mod 64 is never 0 or +-1 for 0 < d < 32, conflict-free for any lane
// permutation) -- on the suspicion that the ~100 words a point operation moves per lane through the LDS were ~40 % of it at 4.25 blocks per compute unit: level
// at every size (profiles/r05_ab_reduce_lds.txt).  The chain is latency (two dependent products + two barriers per operation), not LDS bandwidth.  The arm stays
// behind -DC25519_RC_LANEMAJOR.
#ifndef C25519_RC_LANEMAJOR      // default: word i of lane l at arr[i * 64 + l], ten ds_read_b32 per coordinate
constexpr int RC_WORDS = 40 * 64;
__device__ __forceinline__ feT rc_get(const u32 *arr, int lane, int c) {
    feT r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = arr[(c * 10 + i) * 64 + lane];
    return r;
}
__device__ __forceinline__ void rc_put(u32 *arr, int lane, int c, const feT &a) {
#pragma unroll
    for (int i = 0; i < 10; i++) arr[(c * 10 + i) * 64 + lane] = a.v[i];
}
#else
constexpr int RC_STRIDE = 42, RC_WORDS = RC_STRIDE * 64;
__device__ __forceinline__ feT rc_get(const u32 *arr, int lane, int c) {
    const uint2 *q = reinterpret_cast<const uint2 *>(arr + lane * RC_STRIDE + c * 10);
    feT r;
#pragma unroll
    for (int i = 0; i < 5; i++) { const uint2 v = q[i]; r.v[2 * i] = v.x; r.v[2 * i + 1] = v.y; }
    return r;
}
__device__ __forceinline__ void rc_put(u32 *arr, int lane, int c, const feT &a) {
    uint2 *q = reinterpret_cast<uint2 *>(arr + lane * RC_STRIDE + c * 10);
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = make_uint2(a.v[2 * i], a.v[2 * i + 1]);
}
#endif
__device__ __forceinline__ feT rc_ident(int c) { return (c == 1 || c == 2) ? fe_one() : fe_zero(); }
// coordinate c of point idx of a p40 array in global memory (40 consecutive u32 per point; a coordinate is only 8-byte aligned)
__device__ __forceinline__ feT rc_global(const u32 *base, u64 idx, int c) {
    const uint2 *q = reinterpret_cast<const uint2 *>(base + idx * 40 + c * 10);
    feT r;
#pragma unroll
    for (int i = 0; i < 5; i++) { const uint2 v = q[i]; r.v[2 * i] = v.x; r.v[2 * i + 1] = v.y; }
    return r;
}
__device__ __forceinline__ void rc_global_put(u32 *base, u64 idx, int c, const feT &a) {
    uint2 *q = reinterpret_cast<uint2 *>(base + idx * 40 + c * 10);
#pragma unroll
    for (int i = 0; i < 5; i++) q[i] = make_uint2(a.v[2 * i], a.v[2 * i + 1]);
}
// which coordinate of the RESULT a wave produces in stage 2 (and owns when points are copied): wave 0 X, 1 Y, 2 T, 3 Z
__device__ __forceinline__ int rc_coord(int role) { return role == 0 ? 0 : role == 1 ? 1 : role == 2 ? 3 : 2; }

// dst[dl] = A + B for all 64 logical lanes at once.  getA / getB: coordinate c (0 X, 1 Y, 2 Z, 3 T) of this lane's operands (LDS, global
// memory or the identity).  role is wave-uniform.  Two barriers; every thread of the block must call it.
struct rc_nohook { __device__ __forceinline__ void operator()() const {} };
// after: called between stage 1 and its barrier (the operands have been consumed: the place to issue the loads of the next operand)
template <class GA, class GB, class HOOK = rc_nohook>
__device__ __forceinline__ void rc_add(int role, int lane, GA getA, GB getB, u32 *scratch, u32 *dst, int dl, HOOK after = HOOK()) {
    // The factor 2d of C = 2d T1 T2 would be a third multiplication on wave 2's path.  d = -121665 / 121666 (the curve constant as the
    // reference's AVX2 backend uses it, backend/vector/avx2/edwards.rs: scaled "cached" coordinates): all four products are scaled by
    // 121666 instead -- A' = 121666 A, B' = 121666 B, D' = 2 * 121666 ZZ, K = 2 * 121665 TT = -C' -- by ten products each instead of a
    // hundred, which scales the result (X : Y : Z : T) by 121666^2: the same point.
    feT prod;
    if (role == 0) prod = fe_mul_small(fe_mul(fe_sub(getA(1), getA(0)), fe_sub(getB(1), getB(0))), 121666u);
    else if (role == 1) prod = fe_mul_small(fe_mul(fe_add(getA(1), getA(0)), fe_add(getB(1), getB(0))), 121666u);
    else if (role == 2) prod = fe_mul_small(fe_mul(getA(3), getB(3)), 243330u);
    else prod = fe_mul_small(fe_mul(getA(2), getB(2)), 243332u);
    rc_put(scratch, lane, role, prod);
    after();
    __syncthreads();
    const feT A = rc_get(scratch, lane, 0), B = rc_get(scratch, lane, 1), K = rc_get(scratch, lane, 2), D = rc_get(scratch, lane, 3);
    const feL E = fe_sub(B, A), H = fe_add(B, A);
    const feL F = fe_add(D, K), G = fe_sub(D, K);          // F = D - C, G = D + C with C = -K
    feT out;
    if (role == 0) out = fe_mul(feW(F), E);
    else if (role == 1) out = fe_mul(feW(G), H);
    else if (role == 2) out = fe_mul(feW(E), H);
    else out = fe_mul(feW(F), G);
    rc_put(dst, dl, rc_coord(role), out);
    __syncthreads();
}
// arr[lane] = 2 * arr[lane] for all lanes (ge_dbl + ge_p1p1_to_p3)
__device__ __forceinline__ void rc_dbl(int role, int lane, u32 *arr, u32 *scratch) {
    feT sq;
    if (role == 0) sq = fe_sq(rc_get(arr, lane, 0));
    else if (role == 1) sq = fe_sq(rc_get(arr, lane, 1));
    else if (role == 2) sq = fe_sq(fe_add(rc_get(arr, lane, 0), rc_get(arr, lane, 1)));
    else sq = fe_sq(rc_get(arr, lane, 2));
    rc_put(scratch, lane, role, sq);
    __syncthreads();
    const feT XX = rc_get(scratch, lane, 0), YY = rc_get(scratch, lane, 1), S = rc_get(scratch, lane, 2), ZZ = rc_get(scratch, lane, 3);
    const feL YpX = fe_add(YY, XX), YmX = fe_sub(YY, XX), ZZ2 = fe_twice(ZZ);
    const feL cX = fe_sub(S, fe_carry(YpX));
    const feW cT = fe_sub_w(ZZ2, YmX);
    feT out;
    if (role == 0) out = fe_mul(cT, cX);
    else if (role == 1) out = fe_mul(feW(YmX), YpX);
    else if (role == 2) out = fe_mul(feW(cX), YpX);
    else out = fe_mul(cT, YmX);
    rc_put(arr, lane, rc_coord(role), out);
    __syncthreads();
}

__device__ __forceinline__ feT rc_tot(const u32 *tot, int c) { feT r; for (int i = 0; i < 10; i++) r.v[i] = tot[c * 10 + i]; return r; }
// in: S[l], W[l] for the 64 logical lanes.  out: S[0] = sum_l W_l + 2^shift * sum_l l * S_l (in every lane unless scaled_tot); tot = sum_l S_l (one point, 40 words).
// (wave_weighted_sum of msm.hip: suffix scan, then sum_l l S_l = sum_{l >= 1} T_l, then a butterfly)
// scaled_tot (r6, late; level A of a two-level reduction): lane 63 of W ends with 2^6 * tot.  Level B weights segment j by j * 2^(lb + 6); with the segment totals
// arriving pre-multiplied by 2^6 it doubles lb times instead of lb + 6 -- six point operations (~13 us) off the tail of EVERY call of 12 288 terms and more -- and the
// six doublings cost nothing here: in the butterfly only lane 0's result is used, whose cone of operands at the step of distance d is lanes 0 .. d - 1, so lane 63's
// result is never used, and its threads double the total instead -- in lane 63's slot of W, which is dead by then -- step after step, as a complete addition
// of a point to itself (edwards.rs:795-800 is complete: P + P is the doubling).
// plus_tot: the result is sum_l W_l + 2^shift * sum_l l * S_l + tot -- tot rides in lane 0, whose own S (T_0, weight 0) has just been replaced by the identity, from
// the addition of the W_l on: no step of its own.
// nl (a power of two, <= 64): lanes nl .. 63 hold the identity in S and W (level B of a window with fewer than 64 segments): the scan and the butterfly skip their levels
__device__ __forceinline__ void rc_weighted_sum(int role, int lane, u32 *S, u32 *W, u32 *tot, u32 *scratch, int shift, bool scaled_tot = false, bool plus_tot = false, int nl = 64) {
#pragma unroll 1
    for (int d = 1; d < nl; d <<= 1) {
        const bool in = lane + d < 64;
        const int o = in ? lane + d : lane;
        rc_add(role, lane, [&](int c) { return rc_get(S, lane, c); }, [&](int c) { return in ? rc_get(S, o, c) : rc_ident(c); }, scratch, S, lane);
    }
    // lane 0 holds the total: keep it, and take it out of the weighted part (T_0 has weight 0)
    const int mc = rc_coord(role);
    if (lane == 0) {
        const feT t = rc_get(S, 0, mc);
        for (int i = 0; i < 10; i++) tot[mc * 10 + i] = t.v[i];
        rc_put(S, 0, mc, rc_ident(mc));
    }
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < shift; i++) rc_dbl(role, lane, S, scratch);
    // (the per-lane special cases below are COPIES into the lane's ordinary slot and per-lane array / index selects -- a conditional operand inside rc_add makes every
    //  wave read both alternatives: level A 47 -> 52 us at 2^14 terms that way)
    if (plus_tot) {                                          // lane 0's slot holds the identity: the total goes there, with weight 1, in front of the W_l
        if (lane == 0) rc_put(S, 0, mc, rc_tot(tot, mc));
        __syncthreads();
    }
    rc_add(role, lane, [&](int c) { return rc_get(S, lane, c); }, [&](int c) { return rc_get(W, lane, c); }, scratch, S, lane);
    if (scaled_tot) {                                        // W is dead from here: lane 63's slot of it carries the doubling chain of the total
        if (lane == 63) rc_put(W, 63, mc, rc_tot(tot, mc));
        __syncthreads();
    }
    const bool dbl = scaled_tot && lane == 63;
    u32 *arr = dbl ? W : S;
#pragma unroll 1
    for (int d = scaled_tot ? 32 : nl / 2; d > 0; d >>= 1) {
        const int o = dbl ? 63 : lane ^ d;
        rc_add(role, lane, [&](int c) { return rc_get(arr, lane, c); }, [&](int c) { return rc_get(arr, o, c); }, scratch, arr, lane);
    }
}
// level A: block = segment `seg` (64 x 2^lb buckets, 2^lb per logical lane: 512 / 8, or 1024 / 16 for 17-bit windows) of window k.  direct: the window has a single segment, write col_k itself.
// k0: first window of the launch (a window group, msm_geom: the blocks of the launch are the segments of windows k0 ..)
__global__ void __launch_bounds__(256) k_reduce_a4(const u32 *__restrict__ buckets, int half, int nseg, int lb, u32 *__restrict__ SW, u32 *__restrict__ cols, int direct,
                                                   const u32 *__restrict__ bad_ws, int k0) {
    C25519_PRIO_SIDE();
    __shared__ __attribute__((aligned(16))) u32 S[RC_WORDS], W[RC_WORDS], scratch[RC_WORDS], tot[40];
    // (the roles rotate with the block index: the waves of the ~4 blocks that share a SIMD then play different roles, whose loads differ)
    const int role = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) + blockIdx.x) & 3), lane = threadIdx.x & 63;
    const int bid = k0 * nseg + (int)blockIdx.x, k = bid / nseg, seg = bid % nseg;
    if (bad_ws && bid == 0 && threadIdx.x == 0 && *bad_ws) atomicOr(cols + MSM_MAX_WIN * 40, 1u);
    const int LB = 1 << lb, b0 = (seg * 64 + lane) * LB;
    const u32 *B = buckets + (u64)k * half * 40;
    auto bucket = [&](int b) { return [=](int c) { return b < half ? rc_global(B, (u64)b, c) : rc_ident(c); }; };
    {   // run = acc = B[b0 + 7]
        const int mc = rc_coord(role);
        const feT v = bucket(b0 + LB - 1)(mc);
        rc_put(S, lane, mc, v); rc_put(W, lane, mc, v);
    }
    __syncthreads();
#pragma unroll 1
    for (int j = LB - 2; j >= 0; j--) {
        rc_add(role, lane, [&](int c) { return rc_get(S, lane, c); }, bucket(b0 + j), scratch, S, lane);                                          // run += B_j
        if (j > 0) rc_add(role, lane, [&](int c) { return rc_get(W, lane, c); }, [&](int c) { return rc_get(S, lane, c); }, scratch, W, lane);   // acc += run
    }
    // S[0] = W_seg = sum (b - segment base + 1) B_b -- the "+ 1" of the weights b + 1 rides along segment by segment (until late in round 6: added once per window, in
    // level B, as a step of its own) -- and, two levels, W[63] = 2^6 S_seg
    rc_weighted_sum(role, lane, S, W, tot, scratch, lb, !direct, true);
    const int mc = rc_coord(role);
    if (direct) {
        if (lane == 0) rc_global_put(cols, (u64)k, mc, rc_get(S, 0, mc));
    } else if (lane == 0) {
        rc_global_put(SW, 2 * (u64)bid, mc, rc_get(W, 63, mc));            // 2^6 S_seg
        rc_global_put(SW, 2 * (u64)bid + 1, mc, rc_get(S, 0, mc));
    }
}
// level B: one block per window over its nseg <= 64 segment pairs (weight 2^(lb + 6) per segment, of which level A has applied 2^6)
__global__ void __launch_bounds__(256) k_reduce_b4(const u32 *__restrict__ SW, int nseg, int lb, u32 *__restrict__ cols, int k0) {
    C25519_PRIO_SIDE();
    __shared__ __attribute__((aligned(16))) u32 S[RC_WORDS], W[RC_WORDS], scratch[RC_WORDS], tot[40];
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int k = k0 + (int)blockIdx.x, mc = rc_coord(role);
    rc_put(S, lane, mc, lane < nseg ? rc_global(SW, 2 * ((u64)k * nseg + lane), mc) : rc_ident(mc));
    rc_put(W, lane, mc, lane < nseg ? rc_global(SW, 2 * ((u64)k * nseg + lane) + 1, mc) : rc_ident(mc));
    __syncthreads();
    int nl = 1; while (nl < nseg) nl <<= 1;
    rc_weighted_sum(role, lane, S, W, tot, scratch, lb, false, false, nl);        // (S_j arrives as 2^6 S_j, W_j with the weights b + 1: no further doublings, no final addition)
    if (lane == 0) rc_global_put(cols, (u64)k, mc, rc_get(S, 0, mc));
}

// ---- (r6) level B with the record's publication (the mid path, mid.hip) -------------------------------------------------------------------------------------------
// k_reduce_b4, and then: the block that finishes the LAST window writes the record's header (hdr) / ORs the "a scalar has bit 255 set" flag of the front kernel's blocks
// into the slot, and -- pub.on -- the columns having gone straight into the context's page-locked host slot, releases the sequence word the host polls (msm.hip
// wait_published; the same mechanism, and the same recovery, as the small path's small_publish).  *done_cnt: zeroed by k_mid_front, the kernel that always precedes
// this one on the stream.
// Measured and NOT adopted: both levels in ONE launch (a level-A block fences its segment pair to the device and counts itself on the window's counter; the block that
// finishes a window's last segment runs level B in place).  One launch and one inter-kernel gap less -- and 137 against 47 + 42 us at 2^14 terms, 220 against 89 + 44 at
// 2^16: an agent-scope release on this GPU writes the L2 back, and 2800 - 4900 of them (four storing lanes in each of 700 - 1200 blocks) queue up on the eight L2s at
// ~0.15 us each (profiles/r06_timeline_mid_first.txt).  Here only the 18 - 24 level-B blocks fence.
__global__ void __launch_bounds__(256) k_reduce_b4pub(const u32 *__restrict__ SW, int nwin, int nseg, int lb, u32 *__restrict__ cols, const u32 *__restrict__ blockflags, int nflags,
                                                      u32 *__restrict__ done_cnt, reduce_publish pub) {
    C25519_PRIO_SIDE();
    __shared__ __attribute__((aligned(16))) u32 S[RC_WORDS], W[RC_WORDS], scratch[RC_WORDS], tot[40];
    __shared__ int s_last;
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int k = (int)blockIdx.x, mc = rc_coord(role);
    rc_put(S, lane, mc, lane < nseg ? rc_global(SW, 2 * ((u64)k * nseg + lane), mc) : rc_ident(mc));
    rc_put(W, lane, mc, lane < nseg ? rc_global(SW, 2 * ((u64)k * nseg + lane) + 1, mc) : rc_ident(mc));
    __syncthreads();
    int nl = 1; while (nl < nseg) nl <<= 1;
    rc_weighted_sum(role, lane, S, W, tot, scratch, lb, false, false, nl);        // (as in k_reduce_b4)
    if (lane == 0) {
        rc_global_put(cols, (u64)k, mc, rc_get(S, 0, mc));
        if (pub.on) __threadfence_system(); else __threadfence();      // (every storing lane fences its own stores, then the block counts itself)
    }
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(done_cnt, 1u) == (u32)nwin - 1u;
    __syncthreads();
    if (!s_last) return;
    int bad = 0;
    for (int i = threadIdx.x; i < nflags; i += 256) bad |= (int)blockflags[i];
    const int any_bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        u32 *f = cols + MSM_MAX_WIN * 40;
        if (pub.hdr) {
            for (int i = 0; i < 16; i++) f[i] = 0;
            f[0] = (u32)any_bad; f[REC_TERMS_LO] = pub.terms_lo; f[REC_TERMS_HI] = pub.terms_hi; f[REC_PASSES] = 1; f[REC_MAGIC] = REC_MAGIC_VALUE; f[REC_C] = pub.c;
        } else if (pub.on && pub.dev_flags) {
            for (int i = 0; i < 16; i++) f[i] = pub.dev_flags[i];      // (written by kernels that precede this one on its stream, or that it waited for: the hash chain, the decompressions)
            f[0] |= (u32)any_bad;
        } else if (any_bad) atomicOr(f, 1u);
        if (pub.on) {
            __threadfence_system();
            __hip_atomic_store(pub.host_flag, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// the flags and the header alone (one block): layouts whose windows have a single segment (level A writes the column sums itself)
__global__ void __launch_bounds__(256) k_mid_finish(u32 *__restrict__ cols, const u32 *__restrict__ blockflags, int nflags, reduce_publish pub) {
    int bad = 0;
    for (int i = threadIdx.x; i < nflags; i += 256) bad |= (int)blockflags[i];
    const int any_bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        u32 *f = cols + MSM_MAX_WIN * 40;
        if (pub.hdr) {
            for (int i = 0; i < 16; i++) f[i] = 0;
            f[0] = (u32)any_bad; f[REC_TERMS_LO] = pub.terms_lo; f[REC_TERMS_HI] = pub.terms_hi; f[REC_PASSES] = 1; f[REC_MAGIC] = REC_MAGIC_VALUE; f[REC_C] = pub.c;
        } else if (pub.on && pub.dev_flags) {
            for (int i = 0; i < 16; i++) f[i] = pub.dev_flags[i];      // (written by kernels that precede this one on its stream, or that it waited for: the hash chain, the decompressions)
            f[0] |= (u32)any_bad;
        } else if (any_bad) atomicOr(f, 1u);
        if (pub.on) {
            __threadfence_system();
            __hip_atomic_store(pub.host_flag, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace c25519

// the bucket reduction of the mid path: level A, then level B with the record's header / publication (a single-segment layout: level A writes the columns, then the flags alone)
void launch_bucket_reduce_pub(const uint32_t *buckets, const c25519::msm_geom &g, int nseg, uint32_t *SW, uint32_t *out, const uint32_t *blockflags, int nflags, uint32_t *done_cnt,
                              const c25519::reduce_publish &pub, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_a4, dim3((unsigned)(g.nwin * nseg)), dim3(256), 0, st, buckets, g.half, nseg, red_lb_log2(g.half), SW, out, nseg == 1 ? 1 : 0, (const u32 *)nullptr, 0);
    if (nseg > 1) hipLaunchKernelGGL(k_reduce_b4pub, dim3((unsigned)g.nwin), dim3(256), 0, st, SW, g.nwin, nseg, red_lb_log2(g.half), out, blockflags, nflags, done_cnt, pub);
    else hipLaunchKernelGGL(k_mid_finish, dim3(1), dim3(256), 0, st, out, blockflags, nflags, pub);
}

// the bucket reduction of a pass (level A over the segments, level B over the windows) on stream st
void launch_bucket_reduce4(const uint32_t *buckets, const c25519::msm_geom &g, int nseg, uint32_t *SW, uint32_t *d_slot, const uint32_t *bad_ws, hipStream_t st, int k0, int k1) {
    if (k1 < 0) k1 = g.nwin;
    if (k1 <= k0) return;
    // (bad_ws is folded into the slot by the block of window 0, segment 0: the launch of the first group)
    hipLaunchKernelGGL(k_reduce_a4, dim3((unsigned)((k1 - k0) * nseg)), dim3(256), 0, st, buckets, g.half, nseg, red_lb_log2(g.half), SW, d_slot, nseg == 1 ? 1 : 0, k0 == 0 ? bad_ws : nullptr, k0);
    if (nseg > 1) hipLaunchKernelGGL(k_reduce_b4, dim3((unsigned)(k1 - k0)), dim3(256), 0, st, SW, nseg, red_lb_log2(g.half), d_slot, k0);
}
