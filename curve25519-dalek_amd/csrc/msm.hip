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
#include <stdio.h>
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
#include "msm_sort.h"
#include "host51.h"
#include "ffi.h"
#include <functional>

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
// raw 160-byte points: Montgomery-trick normalisation, CH points per lane (cf. k_compress_p32).  The kernel moves 1.1 GB per
// 2^21 points (Z, then X, Y, Z again, 48-byte prefix products out and back, 128-byte records out): 0.30 ms against a
// memory floor of ~0.27 ms.  Tried in round 2 and dropped: one inversion per BLOCK through an LDS tree (-45 % field
// operations, but one wave inverts while three wait: 0.35 ms; 0.40 ms when LLVM moved the wave-uniform inversion to the
// scalar unit), CH = 32 / 8, fetching every record one step ahead (+12 VGPRs, 0.34 ms), and three launches (lane products,
// ONE batched inversion over the lane totals, unwind: -68 % field operations, 0.083 + 0.107 + 0.192 ms -- the two
// memory passes run at 3.6 - 5.2 TB/s and then contend with the sort on the second stream: no gain end to end), and
// wave-coalesced record I/O transposed through LDS (8x fewer cache-line requests per instruction, but 40 + 30 + 32 LDS
// dword accesses and four barriers per point: 0.47 ms).  Round 4: the way up fetching only Z (the three 16-byte pieces 5 .. 7 of every record,
// 3 DMA instructions into a compact layout instead of 10): level at every size (2^24 terms 13.09 - 13.20 against 13.14 - 13.15 ms, 2^21
// 1.88 - 1.90 against 1.89 - 1.93) -- a record's Z shares its 128-byte lines with X and Y; profiles/r04_ab_prep_z_only.txt.
// WAVE-COALESCED memory accesses.  Lane t owns points t, t + T, ...: the 64 lanes of a wave own 64 CONSECUTIVE points at
// every step; if each lane fetched its own 40-byte coordinates and stored its own 128-byte record (round 1's k_prep_raw),
// every memory instruction would look up 64 different cache lines -- 1664 look-ups per point and wave, on
// the texture/L1 path that the accumulation of the previous pass (one gather per addition) and the sort also live on.
// Here the wave DMAs the whole 10 KB block of its 64 points into LDS (global_load_lds_dwordx4, 8 lines per instruction),
// the prefix products live in a [step][piece][lane] layout (8 lines per instruction), and the records go out through an
// LDS transpose (piece c of record r at position (c + r) mod 8: conflict-free both ways) as eight fully coalesced
// stores: 272 look-ups per point and wave.  The block of step j+1 (j-1 on the way back) is in flight during step j.
// NT (round 6, A/B knob PREP_NT): the normaliser's STREAMING traffic -- the raw points it reads (twice) and the prefix products it writes and reads back, 2.7 GB per
// 2^24-term call -- with the non-temporal cache policy, so that it stops competing for the MALL with the 215 MB of gather records a pass's accumulation lives on (the
// records it WRITES keep the default policy: they are what the accumulation wants resident).  profiles/r06_ab_mall.txt has the measurement.
typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4)));
template <int NT> __device__ __forceinline__ void prep_store16(uint4 *p, const uint4 &v) {
    if (NT) __builtin_nontemporal_store((nt_u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<nt_u32x4 *>(p)); else *p = v;
}
template <int NT> __device__ __forceinline__ uint4 prep_load16(const uint4 *p) {
    if (NT) { const nt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
    return *p;
}
// SPEC (round 6): points that are EXPECTED to be affine -- the decompressed points VerifyingKeys carry (verifying.rs:64-71; c25519_decompress_batch writes Z = 1).
// A wave first runs ONE pass over its points: reads X, Y, Z, writes the record of (X, Y), notes whether every Z was 1 -- no prefix products, no second read, no
// inversion: 288 instead of 544 bytes per point -- and returns if they all were; a wave that met another Z goes through the general algorithm below from the start
// (its records are rewritten).  verify_batch of 2^20 signatures with cached key points: the normalisation of the keys is the head of the main stream's chain, 0.35 of
// its 1.32 ms (profiles/r06_timeline_verify_blake2b.txt); 0.11 this way.  (As two launches -- this pass, then the general kernel returning at once on a device flag --
// the second launch still took 60 - 70 us beside k_hram: its 128 blocks of 72 KB LDS wait for room.  profiles/r06_ab_prep_affine.txt)
template <int CH, int WPB, int NT = 0, int SPEC = 0>        // points per lane, waves per block, streaming accesses, speculative affine pass
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
    // (the clamp of the last, partial block is decided per WAVE, not as twenty per-lane selects per step)
#define C25519_PREP_ISSUE(j)                                                                                                   \
    {                                                                                                                          \
        const u64 b4 = (w0 + (u64)(j) * T) * 10;                                                                               \
        if (b4 + 639 <= last4) {                                                                                               \
            _Pragma("unroll") for (int i = 0; i < 10; i++)                                                                    \
                __builtin_amdgcn_global_load_lds((gbl_void *)(in4 + b4 + (u64)(i * 64) + lane), (lds_void *)(sin + i * 64), 16, 0, NT ? 2 : 0); \
        } else {                                                                                                               \
            _Pragma("unroll") for (int i = 0; i < 10; i++) {                                                                  \
                u64 a = b4 + (u64)(i * 64) + lane;                                                                             \
                a = a > last4 ? last4 : a;                                                                                     \
                __builtin_amdgcn_global_load_lds((gbl_void *)(in4 + a), (lds_void *)(sin + i * 64), 16, 0, NT ? 2 : 0);        \
            }                                                                                                                  \
        }                                                                                                                      \
    }
    const uint4 *my4 = sin + lane * 10;
    const uint2 *my2 = reinterpret_cast<const uint2 *>(sin) + lane * 20;
    uint4 *pre4 = reinterpret_cast<uint4 *>(prefix) + (w0 / 64) * (u64)(CH * 3 * 64) + lane;
    if (SPEC) {
        const u32 sub = lane >> 3, coff = ((lane & 7u) - sub) & 7u;
        bool aff = true;
        C25519_PREP_ISSUE(0)
#pragma unroll 1
        for (int j = 0; j < nj; j++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint4 x0 = my4[0], x1 = my4[1], y1 = my4[3], y2 = my4[4], z0 = my4[5], z1 = my4[6];
            const uint2 x2 = my2[4], y0 = my2[5], z2 = my2[14];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (j + 1 < nj) C25519_PREP_ISSUE(j + 1)
            const u64 lx[5] = {x0.x | (u64)x0.y << 32, x0.z | (u64)x0.w << 32, x1.x | (u64)x1.y << 32, x1.z | (u64)x1.w << 32, x2.x | (u64)x2.y << 32};
            const u64 ly[5] = {y0.x | (u64)y0.y << 32, y1.x | (u64)y1.y << 32, y1.z | (u64)y1.w << 32, y2.x | (u64)y2.y << 32, y2.z | (u64)y2.w << 32};
            const u64 lz[5] = {z0.x | (u64)z0.y << 32, z0.z | (u64)z0.w << 32, z1.x | (u64)z1.y << 32, z1.z | (u64)z1.w << 32, z2.x | (u64)z2.y << 32};
            const bool in = t + (u64)j * T < n;
            aff = aff && (!in || ((lz[0] == 1) && ((lz[1] | lz[2] | lz[3] | lz[4]) == 0)));
            uint4 q[PTS_Q];
            pts_pieces(fe_from_limbs51(lx), fe_from_limbs51(ly), q);
#pragma unroll
            for (int i = 0; i < PTS_Q; i++) sout[lane * 8 + ((i + lane) & 7u)] = q[i];
            uint4 *dst = reinterpret_cast<uint4 *>(pts) + PTS_Q * (dst0 + w0 + (u64)j * T);
#pragma unroll
            for (int i = 0; i < PTS_Q; i++) {
                const u32 r = 8u * i + sub;
                const uint4 v = sout[i * 64 + lane];
                if (w0 + (u64)j * T + r < n) dst[r * 8 + coff] = v;
            }
        }
        if (__ballot(!aff) == 0ull) return;                  // every Z of this wave's points was 1: the records are written
    }
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
        {
            // (a lane whose point of this step is out of range works on the clamped block: its prefix slot is its own, its running product is
            //  kept by an explicit select)
            const bool in = t + (u64)j * T < n;
            const u64 l[5] = {z0.x | (u64)z0.y << 32, z0.z | (u64)z0.w << 32, z1.x | (u64)z1.y << 32, z1.z | (u64)z1.w << 32, z2.x | (u64)z2.y << 32};
            affine = affine && (!in || ((l[0] == 1) && ((l[1] | l[2] | l[3] | l[4]) == 0)));
            prep_store16<NT>(&pre4[(j * 3 + 0) * 64], make_uint4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]));
            prep_store16<NT>(&pre4[(j * 3 + 1) * 64], make_uint4(acc.v[4], acc.v[5], acc.v[6], acc.v[7]));
            prep_store16<NT>(&pre4[(j * 3 + 2) * 64], make_uint4(acc.v[8], acc.v[9], 0u, 0u));
            acc = fe_select_m(acc, fe_mul(acc, fe_from_limbs51(l)), lane_mask(in));
        }
    }
    feT inv = fe_one();
    if (__ballot(!affine) != 0ull) inv = fe_select_m(inv, fe_invert(acc), lane_mask(!affine));      // (wave-uniform branch, explicit select)
    const u32 sub = lane >> 3, coff = ((lane & 7u) - sub) & 7u;
    uint4 pa = make_uint4(0, 0, 0, 0), pb = pa, pc = pa;     // prefix product of the step about to be unwound
    if (nj > 0) { pa = prep_load16<NT>(&pre4[((nj - 1) * 3 + 0) * 64]); pb = prep_load16<NT>(&pre4[((nj - 1) * 3 + 1) * 64]); pc = prep_load16<NT>(&pre4[((nj - 1) * 3 + 2) * 64]); }
    if (nj > 0) C25519_PREP_ISSUE(nj - 1)
#pragma unroll 1
    for (int j = nj - 1; j >= 0; j--) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4 x0 = my4[0], x1 = my4[1], y1 = my4[3], y2 = my4[4], z0 = my4[5], z1 = my4[6];
        const uint2 x2 = my2[4], y0 = my2[5], z2 = my2[14];
        const uint4 a = pa, b = pb, c = pc;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j > 0) {
            pa = prep_load16<NT>(&pre4[((j - 1) * 3 + 0) * 64]); pb = prep_load16<NT>(&pre4[((j - 1) * 3 + 1) * 64]); pc = prep_load16<NT>(&pre4[((j - 1) * 3 + 2) * 64]);
            C25519_PREP_ISSUE(j - 1)
        }
        const u64 lx[5] = {x0.x | (u64)x0.y << 32, x0.z | (u64)x0.w << 32, x1.x | (u64)x1.y << 32, x1.z | (u64)x1.w << 32, x2.x | (u64)x2.y << 32};
        const u64 ly[5] = {y0.x | (u64)y0.y << 32, y1.x | (u64)y1.y << 32, y1.z | (u64)y1.w << 32, y2.x | (u64)y2.y << 32, y2.z | (u64)y2.w << 32};
        const u64 lz[5] = {z0.x | (u64)z0.y << 32, z0.z | (u64)z0.w << 32, z1.x | (u64)z1.y << 32, z1.z | (u64)z1.w << 32, z2.x | (u64)z2.y << 32};
        // (no zero-filling and no merging of per-lane branches here: lanes whose point is out of range compute on the clamped block and are
        //  masked at the store; the per-lane choices are explicit selects on lane masks, fe26.h)
        uint4 q[PTS_Q];
        const bool in = t + (u64)j * T < n;
        feT x = fe_from_limbs51(lx), y = fe_from_limbs51(ly);
        if (__ballot(!affine) != 0ull) {                    // wave-uniform: a wave of VerifyingKey points (Z = 1 throughout) skips the unwinding
            feT pre;
            pre.v[0] = a.x; pre.v[1] = a.y; pre.v[2] = a.z; pre.v[3] = a.w; pre.v[4] = b.x; pre.v[5] = b.y; pre.v[6] = b.z; pre.v[7] = b.w;
            pre.v[8] = c.x; pre.v[9] = c.y;
            const feT zi = fe_mul(inv, pre);
            const lanemask proj = lane_mask(in && !affine);
            inv = fe_select_m(inv, fe_mul(inv, fe_from_limbs51(lz)), proj);
            x = fe_select_m(x, fe_mul(x, zi), proj); y = fe_select_m(y, fe_mul(y, zi), proj);
        }
        pts_pieces(x, y, q);
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


// long list alone (skewed inputs: e.g. the +1 carry digit of every unsigned 128-bit z_i in verify_batch lands
// ~n/2 terms in ONE bucket; identical scalars do the same in every window).
// (k_accumulate lives in accum.hip: the chained-carry products in lockstep)
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
// level A: block (one wave) = segment `seg` of window k.  direct: the window has a single segment, write col_k itself.
// bad_ws (may be null): the sort's "a scalar has bit 255 set" word, ORed into the slot's flag 0 (the sort does not touch the slot)
__global__ void __launch_bounds__(64) k_reduce_a(const u32 *__restrict__ buckets, int half, int nseg, int lb, u32 *__restrict__ SW, u32 *__restrict__ cols, int direct,
                                                 const u32 *__restrict__ bad_ws) {
    C25519_PRIO_SIDE();
    const int k = blockIdx.x / nseg, seg = blockIdx.x % nseg, lane = threadIdx.x;
    if (bad_ws && blockIdx.x == 0 && lane == 0 && *bad_ws) atomicOr(cols + MSM_MAX_WIN * 40, 1u);
    const int LB = 1 << lb, b0 = (seg * 64 + lane) * LB;
    const u32 *B = buckets + (u64)k * half * 40;
    const ge_p3 id = ge_identity();
    ge_p3 run = (b0 + LB - 1 < half) ? p40_load(B, b0 + LB - 1) : id;
    ge_p3 acc = run;
#pragma unroll 1
    for (int j = LB - 2; j >= 1; j--) {
        run = ge_add(run, (b0 + j < half) ? p40_load(B, b0 + j) : id);
        acc = ge_add(acc, run);
    }
    run = ge_add(run, (b0 < half) ? p40_load(B, b0) : id);
    wave_weighted_sum(run, acc, lb, lane);                // run (lane 0) = S_seg, acc = W_seg = sum (b - seg base) B_b
    if (lane == 0) {
        if (direct) p40_store(cols, k, ge_add(acc, run));
        else { p40_store(SW, 2 * (u64)blockIdx.x, run); p40_store(SW, 2 * (u64)blockIdx.x + 1, acc); }
    }
}
// level B: one wave per window over its nseg <= 64 segment pairs
__global__ void __launch_bounds__(64) k_reduce_b(const u32 *__restrict__ SW, int nseg, int lb, u32 *__restrict__ cols) {
    C25519_PRIO_SIDE();
    const int k = blockIdx.x, lane = threadIdx.x;
    const ge_p3 id = ge_identity();
    ge_p3 S = lane < nseg ? p40_load(SW, 2 * ((u64)k * nseg + lane)) : id;
    ge_p3 W = lane < nseg ? p40_load(SW, 2 * ((u64)k * nseg + lane) + 1) : id;
    wave_weighted_sum(S, W, lb + 6, lane);                    // 64 x 2^lb buckets per segment
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
// zero a slot and write its header; pre (may be null): counters a caller computed beforehand (whole-batch hashing in the
// transcript z-mode: [0] non-canonical s, [1] bad message offsets), merged into flags [4] and [5]
__global__ void __launch_bounds__(256) k_slot_init(u32 *__restrict__ slot, u32 terms_lo, u32 terms_hi, u32 passes, u32 c, const u32 *__restrict__ pre) {
    for (int i = threadIdx.x; i < C25519_SLOT_U32; i += 256) {
        u32 v = 0;
        const int f = i - MSM_MAX_WIN * 40;
        if (f == REC_TERMS_LO) v = terms_lo;
        else if (f == REC_TERMS_HI) v = terms_hi;
        else if (f == REC_PASSES) v = passes;
        else if (f == REC_MAGIC) v = REC_MAGIC_VALUE;
        else if (f == REC_C) v = c;
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


}  // namespace c25519
// ================================================================================================
// host orchestration
// ================================================================================================

ge_p3 host_p40(const uint32_t *t) {
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
    else { const h51 zi = h51_invert(h51_from_fe(R.Z)); ge_affine_compress(h51_to_fe(h51_mul(h51_from_fe(R.X), zi)), h51_to_fe(h51_mul(h51_from_fe(R.Y), zi)), w); }
    memcpy(out, w, 32);
}

// window width for n terms.  c = log2(n) - 4 balances n additions per window against 2 x 2^(c-1) for its bucket reduction: the choice for
// THROUGHPUT, from 2^20 terms.  Below that a call is a chain of latencies, and the longest link is k_accumulate's serial walk over a bucket's
// list (2^(5) = 32 dependent additions at c = log2 n - 4, whatever n): wider windows shorten the lists faster than they lengthen the reduction
// -- 2^16 terms 0.65 -> 0.45 ms, 2^14 0.55 -> 0.41 (profiles/r04_ab_midrange_windows.txt: +3 bits from 2^13 to 2^16 terms, +2 at 2^12 and 2^17,
// +1 at 2^18 and 2^19).  At most cmax -- 17 for the plain layout (2^16 buckets per window: 512-bucket slices in the sort, 1024-bucket segments
// in the reduction), 16 for the merged layout (u16 digit matrix) and verify_batch.
static int pick_window(uint64_t n, int cmax) {
    int lg = 0; while ((1ull << (lg + 1)) <= n) lg++;
    static const int mid = C25519_KNOB("MSM_MIDRANGE_WINDOWS", 1);      // A/B knob: 0 = c = log2 n - 4 throughout (rounds 1-3)
    int c = lg - 4;
    if (mid && lg >= 12 && lg <= 19) c += lg <= 12 ? 2 : lg <= 16 ? 3 : lg == 17 ? 2 : 1;
    if (c < 5) c = 5;
    if (c > cmax) c = cmax;
    return c;
}

// window layout for n terms (see msm_geom): signed windows share the low 254 - c bits evenly, then an unsigned window of c - 2 bits up to
// bit 251, then the overflow window, bits 252..255 (a canonical scalar leaves at most the recoding carry there).  (Rounds 1-3: unsigned
// window of c - 1 bits up to bit 252, overflow window 253..255 -- the same signed windows; bit 252 of a canonical scalar is clear, so that
// unsigned window used only the lower half of its buckets, cf. msm_slice_params.)
void msm_layout(uint64_t n, msm_geom &g, int cmax_call, int c_exact) {
    static const int cmax_env = std::min(17, std::max(12, C25519_KNOB("MSM_CMAX", 17)));      // A/B knob: 16 = rounds 1-3 (profiles/r04_ab_window_17.txt)
    static const int cforce = C25519_KNOB("MSM_CFORCE", 0);       // A/B knob: this width for every plain layout of the process (0 = choose)
    // (a forced width only where the kernels behind it were built for it: the small path's tables hold windows of 5 .. 7 bits, the chunk-local and
    //  digit-matrix sorts scan at least 64 buckets per slice, i.e. windows of >= 7 bits)
    const bool force_ok = cforce >= 5 && cforce <= 17 && !cmax_call && (n <= msm_small_max() ? cforce <= 7 : cforce >= 7);
    // the small path (small.hip): 5-bit windows below 1024 terms (pick_window's lower clamp), 6-bit ones above (A/B knob MSM_SMALL_C; rounds 4 and early 5: 7 bits from
    // 2048 terms and the bucket pipeline from 4096 -- profiles/r05_ab_small_path_range.txt: 7-bit tables are 40 KB of LDS per block, four blocks per compute unit).
    // (r6) The choice lives HERE, for plain layouts only: inside pick_window it also narrowed msm_merged_layout's windows (the precomputed-static
    // MSM never takes the small path) from 7 .. 12 bits to 6 for 121 .. 722 static points -- twice the digit-terms, all in 32 buckets.
    const bool small_c = n >= 1024 && n <= msm_small_max();
    g.c = c_exact ? c_exact : force_ok ? cforce : small_c ? C25519_KNOB("MSM_SMALL_C", 6) : pick_window(n, cmax_call ? std::min(cmax_call, cmax_env) : cmax_env);
    // (r6) the first sizes of the mid path (mid.hip) want wider windows than the rule, which was tuned on the bucket pipeline in round 4: 13 bits for 8192 .. 16 383 terms
    // (profiles/r06_ab_mid_window.txt: 12 288 terms 0.231 -> 0.213 ms) and 12 bits below (the rule: 10; profiles/r06_ab_small_mid_boundary.txt: 6144 terms 0.229 -> 0.165 ms,
    // 11 bits 0.177, 13 bits 0.175); every other size of the path is at its optimum under the rule
    if (!c_exact && !force_ok && !small_c && n > msm_small_max() && n < 16384) {
        if (g.c == 12) g.c = 13;
        else if (g.c == 10 && n < 8192) g.c = 12;
    }
    g.half = 1 << (g.c - 1);
    const int low_bits = 253 - (g.c - 1), nsig = (low_bits + g.c - 1) / g.c, wbase = low_bits / nsig, wrem = low_bits % nsig;
    uint32_t a[9] = {0};
    int bit = 0;
    for (int k = 0; k < nsig; k++) {
        g.pos[k] = (unsigned char)bit; g.wid[k] = (unsigned char)(wbase + (k < wrem ? 1 : 0));
        bit += g.wid[k];
        a[(bit - 1) >> 5] |= 1u << ((bit - 1) & 31);
    }
    g.pos[nsig] = (unsigned char)bit; g.wid[nsig] = (unsigned char)(g.c - 2);          // bit == 254 - c
    g.pos[nsig + 1] = 252; g.wid[nsig + 1] = 4;
    g.nwin = nsig + 2;
    g.first_unsigned = nsig;
    msm_sort_params(n, g);
    for (int k = g.nwin; k < MSM_MAX_WIN; k++) { g.pos[k] = 0; g.wid[k] = 1; }
    msm_slice_params(g);
    for (int i = 0; i < 8; i++) g.addk[i] = a[i];
    g.ngroups = 1; g.gstart[0] = 0;
    for (int i = 1; i <= MSM_MAX_GROUPS; i++) g.gstart[i] = (unsigned char)g.nwin;
}
// Window groups of a single-pass call (msm_geom): `groups` groups of consecutive windows; `last` (0 = equal shares) = content windows of the final
// group, whose bucket reduction is the exposed tail of the call.  The overflow window (a handful of entries) rides with the final group.  Only layouts
// whose windows have at least 1024 buckets (a block of the bucket-order kernel must not straddle groups) and at least two windows per group.
void msm_set_groups(msm_geom &g, int groups, int last) {
    g.ngroups = 1; g.gstart[0] = 0;
    for (int i = 1; i <= MSM_MAX_GROUPS; i++) g.gstart[i] = (unsigned char)g.nwin;
    groups = std::min(groups, MSM_MAX_GROUPS);
    const int content = g.nwin - 1;
    if (groups < 2 || g.half < 1024 || content < 2 * groups) return;
    if (last < 1 || last > content - (groups - 1)) last = 0;
    int at = 0;
    for (int q = 0; q < groups - 1; q++) {
        const int left = content - at - last, share = last ? (left + (groups - 2 - q)) / (groups - 1 - q) : (content - at + (groups - 1 - q)) / (groups - q);
        at += share;
        g.gstart[q + 1] = (unsigned char)at;
    }
    g.ngroups = (unsigned char)groups;
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

// diagnostics: the sort of one pass alone (scalars on the device; layout as for `layout_terms` terms), `reps` times back to back on the context's
// stream -- for kernel traces of the sort without an accumulation beside it (tools/sort_only.py)
EXPORT int32_t c25519_debug_sort(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, uint64_t layout_terms, int32_t reps) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n <= msm_small_max() || n > (1ull << 22)) { ctx->err = "debug_sort: n outside the range of a bucket-method pass"; return -(int32_t)hipErrorInvalidValue; }
    msm_geom g;
    msm_layout(layout_terms ? layout_terms : n, g);
    for (int i = 0; i < reps; i++) {
        msm_plan pl;
        pl.bad_sticky = nullptr;
        int32_t r = msm_enqueue_sort(ctx, d_scalars, n, g, dslot(ctx, 0), nullptr, pl);
        if (r) return r;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
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

// An MSM is enqueued in two halves so that a caller can put other work between them:
//   msm_enqueue_sort  digits, counting sort, bucket order and the long-bucket work list -- needs only the SCALARS; runs on
//                     sort_stream (nullptr = the context's main stream).  Memory-bound.
//   msm_enqueue_acc   joins sort_stream into the main stream, then accumulation (+ the long-bucket path on the second
//                     stream) and bucket reduction -- needs the POINTS (packed affine Niels records).  VALU-bound.
// The column sums go to d_slot.  ring (may be null): [0] / [1] bracket k_accumulate, [2] = end of the pass.
// wait_acc (may be null): an event the accumulation waits for -- the previous pass's accumulation on the other stream
// set: two accumulations side by side only share the multipliers, while a sort beside an accumulation is free.
// cont: the accumulation starts from the bucket sums the previous pass on this workspace left (no reduction happened in between);
// reduce: the buckets are reduced into d_slot now (the last pass of a stream set; always, for the single-pass callers)
int32_t msm_enqueue_acc(c25519_ctx *ctx, const msm_plan &pl, const uint32_t *d_pts, uint32_t *d_slot, hipEvent_t *ring, hipEvent_t wait_acc, bool cont, bool reduce,
                        const uint32_t *d_bad_sticky) {
    const msm_geom &g = pl.g;
    // (k_accumulate's gathers address a record by a 32-bit byte offset from the pass's base: 2^25 records of 128 bytes.  The sorts' entries hold 23- / 24-bit ids.)
    if (pl.n > (1ull << 25)) { ctx->err = "msm: internal error (a pass of more than 2^25 records)"; return -(int32_t)hipErrorInvalidValue; }
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
    static const int coop_reduce = C25519_KNOB("REDUCE_COOP", 1);     // A/B knob: 0 = rounds 2-3's one-lane-per-point reduction (k_reduce_a / k_reduce_b below)
    // Window groups (msm_geom, single-pass calls): k_accumulate is launched once per group on the main stream -- the groups' stretches of the bucket
    // order are separate -- and the second (high-priority) stream reduces group q as soon as ITS accumulation has finished, beside the accumulation
    // of group q + 1: what remains exposed at the end of the call is the reduction of the last group only.
    const int groups = (g.ngroups > 1 && reduce && !cont && coop_reduce) ? g.ngroups : 1;
    // (r5) A call of ONE pass has no other stream set to compete with: its reduction goes on the MAIN stream, straight behind the accumulation, and the
    // read-back behind the reduction -- two cross-stream hand-overs (12 + 20 us of event latency: gpurun_out/r05_timeline_*) less on the critical path
    // of every single-pass call.  The long-bucket kernels stay on the second stream (they run beside the accumulation); the main stream waits for them
    // before the reduction -- long since finished.
    static const int reduce_main = C25519_KNOB("REDUCE_MAIN", 1);       // A/B knob: 0 = always on the second stream (rounds 3-4)
    const bool red_main = reduce_main && ctx->solo && reduce && !cont && groups == 1 && coop_reduce && !wait_acc;
    for (int q = 0; q < groups; q++) {
        const int k0 = groups > 1 ? g.gstart[q] : 0, k1 = groups > 1 ? g.gstart[q + 1] : g.nwin;
        ctx->kname[0] = launch_accumulate(d_pts, pl.sorted, pl.base, pl.perm + (size_t)k0 * g.half, (uint64_t)(k1 - k0) * g.half, pl.n, g, pl.buckets, cont ? 1 : 0, st);
        if (groups > 1) {
            HIPCHK(hipEventRecord(ctx->ev_grp[q], st));
            HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_grp[q], 0));        // (behind the long-bucket kernels the second stream already holds)
            launch_bucket_reduce4(pl.buckets, g, pl.nseg, pl.SW, d_slot, d_bad_sticky ? d_bad_sticky : pl.bad_ws, ctx->aux, k0, k1);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipEventRecord(ctx->ev_acc, st));
    if (ring) HIPCHK(hipEventRecord(ring[1], st));
    // The bucket reduction runs on the SECOND (high-priority) stream, behind the long-bucket kernels it depends on anyway.  On the
    // main stream its few small blocks had normal priority: once the sort of the next pass stopped being late (round 3) the next
    // accumulation -- on the other stream set -- began before they were dispatched, refilled every hole a retiring block left, and
    // k_reduce_b waited 1.1 ms for its 17 wave slots, holding back this stream set's next pass (profiles/r03_msm_2p24_timeline.txt).
    if (red_main) {
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->aux));          // behind the long-bucket kernels
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
        launch_bucket_reduce4(pl.buckets, g, pl.nseg, pl.SW, d_slot, d_bad_sticky ? d_bad_sticky : pl.bad_ws, st);
        HIPCHK(hipGetLastError());
        if (ring) HIPCHK(hipEventRecord(ring[2], st));
        return C25519_OK;
    }
    if (reduce && groups == 1) HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_acc, 0));      // (without a reduction nothing on the second stream needs the accumulated buckets)
    if (groups > 1) {
    } else if (reduce && coop_reduce) {
        launch_bucket_reduce4(pl.buckets, g, pl.nseg, pl.SW, d_slot, d_bad_sticky ? d_bad_sticky : pl.bad_ws, ctx->aux);
        HIPCHK(hipGetLastError());
    } else if (reduce) {
        hipLaunchKernelGGL(k_reduce_a, dim3((unsigned)(g.nwin * pl.nseg)), dim3(64), 0, ctx->aux, pl.buckets, g.half, pl.nseg, red_lb_log2(g.half), pl.SW, d_slot, pl.nseg == 1 ? 1 : 0, d_bad_sticky ? d_bad_sticky : pl.bad_ws);
        if (pl.nseg > 1) hipLaunchKernelGGL(k_reduce_b, dim3((unsigned)g.nwin), dim3(64), 0, ctx->aux, pl.SW, pl.nseg, red_lb_log2(g.half), d_slot);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->ev_join, ctx->aux));
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));     // what follows on the main stream (the next pass of this stream set, the read-back) comes after the long buckets / the reduction
    if (ring) HIPCHK(hipEventRecord(ring[2], st));
    return C25519_OK;
}
// A pass of at most msm_small_max() terms: no sort, no buckets -- small.hip's two launches on the main stream, after whatever the caller
// put on sort_stream (verify_batch: the batch scalars) and after wait_acc.  src_fmt: 0 raw 160-byte points, 1 affine Niels records.
static int32_t msm_small_pass(c25519_ctx *ctx, const uint8_t *d_scalars, const void *d_points, int src_fmt, uint64_t n, const msm_geom &g, uint32_t *d_slot, hipEvent_t *ring,
                              hipStream_t sort_stream, hipEvent_t wait_acc) {
    hipStream_t st = ctx->stream;
    if (sort_stream && sort_stream != st) {
        HIPCHK(hipEventRecord(ctx->ev_sort, sort_stream));
        HIPCHK(hipStreamWaitEvent(st, ctx->ev_sort, 0));
    }
    if (wait_acc) HIPCHK(hipStreamWaitEvent(st, wait_acc, 0));
    if (ring) HIPCHK(hipEventRecord(ring[0], st));
    ctx->kname[0] = "c25519::k_small_cols (tables of multiples by repeated addition, one lane per (window, term))";
    int32_t r = msm_small_enqueue(ctx, d_scalars, d_points, src_fmt, n, g, d_slot, st);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev_acc, st));
    if (ring) { HIPCHK(hipEventRecord(ring[1], st)); HIPCHK(hipEventRecord(ring[2], st)); }
    return C25519_OK;
}
int32_t msm_enqueue(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, const msm_geom &g, uint32_t *d_slot, hipEvent_t *ring,
                    hipStream_t sort_stream, hipEvent_t wait_acc, const mid_run *run) {
    if (n <= msm_small_max() && g.half <= 64) return msm_small_pass(ctx, d_scalars, d_pts, 1, n, g, d_slot, ring, sort_stream, wait_acc);   // (g.half: the layout may belong to larger sibling passes)
    if (ctx->solo && !wait_acc && msm_mid_serves(n, g, true)) {          // (r6) a single pass of 12 288 .. 2^18 terms over prepared records: the mid path (mid.hip), on the main stream
        hipStream_t st = ctx->stream;
        if (!run && sort_stream && sort_stream != st) {              // (what the caller put on the second stream -- verify_batch: the batch scalars -- comes first)
            HIPCHK(hipEventRecord(ctx->ev_sort, sort_stream));
            HIPCHK(hipStreamWaitEvent(st, ctx->ev_sort, 0));
        }
        // (no per-kernel events here either -- each is a ~5 us gap between two kernels of a chain that is all critical path: the ring entry of the caller (a verify_batch pass)
        //  is marked so that the accumulation / reduction phases of this call answer -1 instead of reading events of an older one; its own phases stay)
        if (ring) {
            const int idx = (int)((hipEvent_t(*)[c25519_ctx::RING_EV])ring - ctx->ring);
            if (idx >= 0 && idx < c25519_ctx::RING) ctx->ring_kind[idx] = (uint8_t)(ctx->ring_kind[idx] == 2 ? 3 : 4);      // 3: verify pass without MSM events, 4: MSM pass without them
        }
        return msm_mid_enqueue(ctx, d_scalars, d_pts, 1, n, g, d_slot, 0, n, nullptr, run);
    }
    msm_plan pl;
    int32_t r = msm_enqueue_sort(ctx, d_scalars, n, g, d_slot, sort_stream, pl);
    if (r) return r;
    return msm_enqueue_acc(ctx, pl, d_pts, d_slot, ring, wait_acc);
}

// total = sum_k 2^pos_k col_k by Horner (pippenger.rs:159), host arithmetic over <= 56 points
ge_p3 msm_horner(const uint32_t *cols, const msm_geom &g) {
    // (host51.h: the 5 x 51-bit layout a 64-bit core multiplies natively -- 25 us per fold instead of 74 through the device layout)
    // (AVX-512 IFMA where the host has it -- doublings and additions with the running total in the lanes of five vectors: host51.h hp3_horner)
    hp3 c[MSM_MAX_WIN];
    int shift[MSM_MAX_WIN];
    for (int j = 0; j < g.nwin; j++) {
        const int k = g.nwin - 1 - j;
        c[j] = hp3_from(host_p40(&cols[(size_t)k * 40]));
        shift[j] = j ? g.pos[k + 1] - g.pos[k] : 0;
    }
    return hp3_to(hp3_horner(c, shift, g.nwin));
}
// ---- how a call's results reach the host ----------------------------------------------------------------------------------------
// hipMemcpyAsync of the slots into page-locked memory + hipStreamSynchronize (rounds 1-4, the default).  The round-4 verdict suspected 40 - 60 us of
// copy-engine launch, interrupt and wake-up in there; round 5 built the alternative -- the last kernel WRITES the slots into the context's coherent,
// device-mapped host buffer and releases a sequence word the host polls (k_publish; a long call first blocks on an event recorded at the end of its
// last accumulation, ctx->coarse_wait, so that it does not burn a core for milliseconds) -- and measured it: 111.8 against 114.8 us for a 1-term MSM, level at every other size (profiles/r05_small_call_phases.txt,
// r05_ab_publish.txt).  The end of a call costs ~3 us; what a small call spends is the host's launch work (~25 us), ~65 us of kernels and launch gaps on the
// GPU and the host fold (~28 us).  The k_publish arm for LARGE calls stays behind the PUBLISH knob of the tuning build; the SMALL path's own publication
// (small.hip small_direct, the release default since round 5) ends in wait_published below.
namespace c25519 {
__global__ void __launch_bounds__(1024) k_publish(const u32 *__restrict__ src, u32 *__restrict__ host_dst, u32 words, u32 *__restrict__ host_flag, u32 seq) {
    for (u32 i = threadIdx.x; i < words; i += 1024) host_dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace c25519
static inline void cpu_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}
// Wait for a kernel to release `seq` into the host's sequence word (behind the slots of h_msm).  (r6) Three phases, none bounded by wall-clock alone:
//   1  spin on the word for at most PUBLISH_SPIN_US (2 ms: the calls that publish are over in 0.06 - 0.3 ms once the stream gets to them)
//   2  hipStreamSynchronize: the stream is busy with a caller's earlier work, or the GPU is shared -- block like every other path of the library does, for as
//      long as the STREAM takes; a GPU fault surfaces here as the stream's own error (round 5 returned hipErrorLaunchTimeOut after 60 s of polling whatever
//      the stream was doing: ADVICE r5)
//   3  the stream has drained without error and the word is still not there: a LOST publication.  Nothing of the kind has been observed (profiles/r06_soak_small.txt);
//      the state is recorded in ctx->err (sequence word, expected number, the device's block counter), counted (c25519_ctx_counter), and the caller re-runs
//      the call once through the slot + copy path of rounds 1-4, which needs no publication (C25519_LOST_PUBLICATION is internal: no entry point returns it).
static int32_t wait_published(c25519_ctx *ctx, uint32_t seq) {
    volatile uint32_t *vf = (uint32_t *)ctx->h_msm + (size_t)(C25519_MAX_SLOTS + 1) * C25519_SLOT_U32;
    static const double spin_us = (double)C25519_KNOB("PUBLISH_SPIN_US", 2000);
    const double t0 = wall_us();
    bool seen = false;
    for (uint64_t spins = 1; !(seen = (*vf == seq)); spins++) {
        cpu_relax();
        if ((spins & 0xff) == 0 && wall_us() - t0 > spin_us) break;
    }
    if (!seen) {
        ctx->counters[C25519_CTR_PUBLISH_BLOCKED]++;
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return c25519_fail(ctx, e, "waiting for a call's results");
        // (the release store precedes the end of the kernel; a drained stream means it has been performed -- a short grace period all the same)
        const double t1 = wall_us();
        for (uint64_t spins = 1; !(seen = (*vf == seq)); spins++) {
            cpu_relax();
            if ((spins & 0xff) == 0 && wall_us() - t1 > 200.0) break;
        }
    }
    if (!seen) {
        ctx->counters[C25519_CTR_PUBLISH_LOST]++;
        uint32_t dev_cnt = 0xffffffffu;
        (void)hipMemcpy(&dev_cnt, (uint32_t *)ctx->d_flag + 56, 4, hipMemcpyDeviceToHost);
        char buf[200];
        snprintf(buf, sizeof buf, "a small call's record was not published (sequence word %u, expected %u, device block counter %u, %.0f us waited): re-run through the copy path",
                 (unsigned)*vf, (unsigned)seq, (unsigned)dev_cnt, wall_us() - t0);
        ctx->err = buf;
        return C25519_LOST_PUBLICATION;
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return C25519_OK;
}
static int32_t publish_and_wait(c25519_ctx *ctx, const uint32_t *d_src, uint32_t *h_dst, size_t words) {
    uint32_t *flag = (uint32_t *)ctx->h_msm + (size_t)(C25519_MAX_SLOTS + 1) * C25519_SLOT_U32;      // the word behind the slots (ctx_make_streams)
    uint32_t seq = ++ctx->publish_seq;
    if (seq == 0) seq = ++ctx->publish_seq;
    ctx->host_us[2] = wall_us();                                          // everything is enqueued
    // Measured and NOT adopted (profiles/r05_small_call_phases.txt, r05_ab_publish.txt): 111.8 against 114.8 us for a 1-term MSM, level everywhere else -- the
    // copy engine + hipStreamSynchronize of rounds 1-4 cost ~3 us, not the 40 - 60 the verdict suspected; the default stays the simple path (knob 0).
    static const int publish = C25519_KNOB("PUBLISH", 0);
    if (!publish) {
        ctx->coarse_wait = nullptr;
        HIPCHK(hipMemcpyAsync(h_dst, d_src, words * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->host_us[3] = wall_us();
        return C25519_OK;
    }
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(1024), 0, ctx->stream, d_src, h_dst, (uint32_t)words, flag, seq);
    HIPCHK(hipGetLastError());
    if (ctx->coarse_wait) { (void)hipEventSynchronize(ctx->coarse_wait); ctx->coarse_wait = nullptr; }      // (blocking: the long part of a long call)
    { const int32_t rw = wait_published(ctx, seq); if (rw) return rw; }
    ctx->host_us[3] = wall_us();                                          // the results are on the host
    return C25519_OK;
}
// slots [0, count) -> the host
int32_t slots_collect(c25519_ctx *ctx, int count) {
    return publish_and_wait(ctx, ctx->d_slots, (uint32_t *)ctx->h_msm, (size_t)count * C25519_SLOT_U32);
}
// the context's own record (slot C25519_MAX_SLOTS of d_slots / h_msm): where a call that answers on the host sums its passes
int32_t rec_collect(c25519_ctx *ctx) {
    if (ctx->direct_seq) {                                     // the small path publishes its record itself (small.hip): poll the sequence word
        const uint32_t seq = ctx->direct_seq;
        ctx->direct_seq = 0;
        ctx->host_us[2] = wall_us();
        const int32_t r = wait_published(ctx, seq);
        ctx->host_us[3] = wall_us();
        return r;
    }
    return publish_and_wait(ctx, drec(ctx), (uint32_t *)ctx->h_msm + (size_t)C25519_MAX_SLOTS * C25519_SLOT_U32, C25519_SLOT_U32);
}
void slot_init(uint32_t *d_slot, uint64_t terms, const uint32_t *d_pre, hipStream_t st, int c) {
    hipLaunchKernelGGL(k_slot_init, dim3(1), dim3(256), 0, st, d_slot, (uint32_t)terms, (uint32_t)(terms >> 32), terms ? 1u : 0u, (uint32_t)c, d_pre);
}
static_assert(C25519_PARTIAL_RECORD_BYTES == C25519_SLOT_U32 * 4, "record = slot");

// Fold `count` records (HOST memory) into one point and one set of counters.  Records made with the same number of
// terms share their window layout: their columns are added window by window and the Horner fold (pippenger.rs:159) runs
// once; otherwise every record is folded on its own.  Pure host arithmetic over O(count x windows) points.
int32_t records_fold(const uint8_t *records, uint64_t count, ge_p3 &R, uint32_t flags[8], std::string *err) {
    R = ge_identity();
    for (int j = 0; j < 8; j++) flags[j] = 0;
    bool same = true;
    uint64_t terms0 = 0;
    uint32_t c0 = 0;
    for (uint64_t i = 0; i < count; i++) {
        uint32_t f[16];
        memcpy(f, records + i * C25519_PARTIAL_RECORD_BYTES + (size_t)MSM_MAX_WIN * 160, sizeof f);
        if (f[REC_MAGIC] != REC_MAGIC_VALUE) { if (err) *err = "fold: not a partial-result record (bad magic)"; return -(int32_t)hipErrorInvalidValue; }
        for (int j = 0; j < 8; j++) flags[j] += f[j];
        const uint64_t terms = (uint64_t)f[REC_TERMS_LO] | ((uint64_t)f[REC_TERMS_HI] << 32);
        // (a record without its window width in the header -- the format before the width stopped being a function of the term count -- would be
        //  folded with the wrong window positions: an explicit version error, never a silently different point)
        if (terms != 0 && f[REC_C] == 0) { if (err) *err = "fold: a partial-result record without a window width in its header (made by an older library version)"; return -(int32_t)hipErrorInvalidValue; }
        if (f[REC_C] != 0 && (f[REC_C] < 5 || f[REC_C] > 17)) { if (err) *err = "fold: bad window width in a record header"; return -(int32_t)hipErrorInvalidValue; }
        if (i == 0) { terms0 = terms; c0 = f[REC_C]; } else if (terms != terms0 || f[REC_C] != c0) same = false;
    }
    if (count == 0) return C25519_OK;
    std::vector<uint32_t> cols((size_t)MSM_MAX_WIN * 40);
    if (same) {
        if (terms0 == 0) return C25519_OK;                  // empty shards only
        msm_geom g;
        msm_layout(terms0, g, 0, (int)c0);
        // columns of equal layouts add window by window (host51.h arithmetic), then ONE Horner fold
        std::vector<hp3> hc((size_t)g.nwin);
        for (uint64_t i = 0; i < count; i++) {
            memcpy(cols.data(), records + i * C25519_PARTIAL_RECORD_BYTES, (size_t)g.nwin * 160);      // (records are only 4-byte aligned in general)
            for (int k = 0; k < g.nwin; k++) {
                const hp3 c = hp3_from(host_p40(&cols[(size_t)k * 40]));
                hc[(size_t)k] = i == 0 ? c : hp3_add(hc[(size_t)k], c);
            }
        }
        hp3 total = hp3_identity();
        for (int k = g.nwin - 1; k >= 0; k--) {
            if (k != g.nwin - 1) total = hp3_pow2_fast(total, g.pos[k + 1] - g.pos[k]);      // (AVX-512 IFMA where the host has it: host51.h)
            total = hp3_add(total, hc[(size_t)k]);
        }
        R = hp3_to(total);
        return C25519_OK;
    }
    for (uint64_t i = 0; i < count; i++) {
        const uint32_t *c = (const uint32_t *)(records + i * C25519_PARTIAL_RECORD_BYTES);
        const uint32_t *f = c + (size_t)MSM_MAX_WIN * 40;
        const uint64_t terms = (uint64_t)f[REC_TERMS_LO] | ((uint64_t)f[REC_TERMS_HI] << 32);
        if (terms == 0) continue;
        msm_geom g;
        msm_layout(terms, g, 0, (int)f[REC_C]);
        memcpy(cols.data(), c, (size_t)g.nwin * 160);        // (records are only 4-byte aligned in general)
        R = ge_add(R, msm_horner(cols.data(), g));
    }
    return C25519_OK;
}

// one-shot MSM over prepared points (extra.hip: precomputed tables): enqueue, collect, fold
int32_t msm_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, ge_p3 &R) {
    R = ge_identity();
    // (the sort keeps a 23-bit term index: beyond 2^22 terms the input is cut into passes, each collected and folded on its own --
    //  this entry point serves the dynamic terms of a precomputed MSM, where such sizes are not the common case)
    constexpr uint64_t CORE_PASS = 1ull << 22;
    for (uint64_t lo = 0; lo < n || lo == 0; lo += CORE_PASS) {
        const uint64_t m = std::min(CORE_PASS, n - lo);
        msm_geom g;
        msm_layout(m, g);
        HIPCHK(hipMemsetAsync(dslot(ctx, 0), 0, C25519_SLOT_U32 * 4, ctx->stream));
        if (m) {
            ctx->solo = true;
            int32_t r = msm_enqueue(ctx, d_scalars + lo * 32, m, d_pts + lo * (PTS_BYTES / 4), g, dslot(ctx, 0), nullptr, nullptr);
            if (r) return r;
            if ((r = slots_collect(ctx, 1))) return r;
            if (slot_flags((uint32_t *)hslot(ctx, 0))[0]) { ctx->err = "msm: a scalar has bit 255 set (Scalar invariant #1 violated)"; return -(int32_t)hipErrorInvalidValue; }
            R = ge_add(R, msm_horner(hslot(ctx, 0), g));
        }
        if (n == 0) break;
    }
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
    m.c = pick_window(std::max<uint64_t>(1, ns) * 17, 16);
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
    launch_merged_table(raw, ns, m.c, m.K, (uint8_t *)ctx->tmp_f.p, st);
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
    msm_slice_params(g);
    HIPCHK(hipMemsetAsync(dslot(ctx, 0), 0, C25519_SLOT_U32 * 4, ctx->stream));
    msm_plan pl;
    ctx->solo = true;
    int32_t r = msm_enqueue_sort(ctx, d_scalars, n, g, dslot(ctx, 0), nullptr, pl, &m);
    if (r) return r;
    if ((r = msm_enqueue_acc(ctx, pl, d_table, dslot(ctx, 0), nullptr, nullptr))) return r;
    if ((r = slots_collect(ctx, 1))) return r;
    if (slot_flags((uint32_t *)hslot(ctx, 0))[0]) { ctx->err = "precomp_msm: internal error (top digit out of range)"; return -(int32_t)hipErrorInvalidValue; }
    R = host_p40(hslot(ctx, 0));                          // one window at position 0: the column sum IS the result
    return C25519_OK;
}

// points in any format -> packed affine Niels at d_pts[dst0..]; *d_badcount counts the encodings that do not decode
int32_t prep_points(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_pts, uint64_t dst0, uint32_t *d_badcount, bool expect_affine) {
    return prep_points_on(ctx, d_points, n, in_fmt, d_pts, dst0, d_badcount, ctx->stream, ctx->prefix, expect_affine);
}
// the same on stream st with the prefix-product scratch `pre` (a second normaliser of one context beside the first needs its own: msm_pass_enqueue)
// expect_affine (raw points): the caller's points normally have Z = 1 (VerifyingKey points): k_prep_affine first, the general normaliser only if one did not
int32_t prep_points_on(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_pts, uint64_t dst0, uint32_t *d_badcount, hipStream_t st, devbuf &pre, bool expect_affine) {
    if (n == 0) return C25519_OK;
    if (in_fmt == C25519_FMT_EDWARDS_Y) HIPCHK(launch_prep_compressed(0, d_points, 1, n, d_pts, dst0, d_badcount, false, st));
    else if (in_fmt == C25519_FMT_RISTRETTO) HIPCHK(launch_prep_compressed(1, d_points, 1, n, d_pts, dst0, d_badcount, false, st));
    else if (in_fmt == C25519_FMT_RAW160) {
        int32_t r;
        // points per lane and inversion: 64 when the launch still has >= 2048 waves (the records of the later passes of a
        // multi-pass call; 2^24 terms: 15.8 ms with 16 everywhere, 15.3 with 64), 32 from 2^20 points (profiles/r04_ab_prep_points_per_lane.txt:
        // 2^20 terms 1.10 against 1.18 ms, 2^21 the same either way), 16 below (2^19: 0.76 against 0.81 ms; 8 and 4 buy nothing down to 2^16)
        static const int ch_knob = C25519_KNOB("PREP_CH", 0);             // A/B knob: 4 / 8 / 16 / 32 / 64
        // below 2^18 points the kernel is a latency chain (the lane's prefix products, ONE inversion, the unwinding): 4 points per lane shorten it
        // (2^13 terms 0.48 -> 0.40 ms, 2^16 0.54 -> 0.51)
        const int CH = ch_knob ? ch_knob : (n >= (1ull << 23) ? 64 : n >= (1ull << 20) ? 32 : n >= (1ull << 18) ? 16 : 4);
        constexpr int wpb = 4;
        const unsigned blocks = (unsigned)div_up64((n + CH - 1) / CH, 64 * wpb);
        // the prefix buffer is addressed per wave (CH x 3 x 64 pieces): blocks x wpb waves of them
        r = ctx_reserve(ctx, pre, (size_t)blocks * wpb * CH * 3 * 64 * 16);
        if (r) return r;
        static const int affine_knob = C25519_KNOB("PREP_AFFINE_FIRST", 1);      // A/B knob of the tuning build
        if (expect_affine && affine_knob && n >= 1024) {
            // VerifyingKey points: the speculative one-pass form (k_prep_raw2 SPEC), 8 points per lane from 2^18 points (2048 waves at 2^20), 4 below
            if (n >= (1ull << 18)) {
                const unsigned bl = (unsigned)div_up64((n + 7) / 8, 64 * wpb);
                if ((r = ctx_reserve(ctx, pre, (size_t)bl * wpb * 8 * 3 * 64 * 16))) return r;
                hipLaunchKernelGGL((k_prep_raw2<8, wpb, 0, 1>), dim3(bl), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)pre.p, d_pts, dst0);
            } else {
                const unsigned bl = (unsigned)div_up64((n + 3) / 4, 64 * wpb);
                if ((r = ctx_reserve(ctx, pre, (size_t)bl * wpb * 4 * 3 * 64 * 16))) return r;
                hipLaunchKernelGGL((k_prep_raw2<4, wpb, 0, 1>), dim3(bl), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)pre.p, d_pts, dst0);
            }
            HIPCHK(hipGetLastError());
            return C25519_OK;
        }
        // A/B proxy (profiles/r04_ab_prep_two_waves.txt): what the normaliser's memory system does with TWO waves per compute unit -- the occupancy
        // an LDS-resident inversion tree (prefix products of 16 points per lane kept in LDS: 40 KB per wave) would leave it
        static const int two_waves = C25519_KNOB("PREP_TWO_WAVES", 0);
        static const int prep_nt = C25519_KNOB("PREP_NT", 0);             // A/B knob: 1 = streaming accesses for the normaliser's inputs and prefix scratch (large launches)
        if (two_waves && CH == 16) {
            const unsigned b2 = (unsigned)div_up64((n + CH - 1) / CH, 64 * 2);
            if ((r = ctx_reserve(ctx, pre, (size_t)b2 * 2 * CH * 3 * 64 * 16))) return r;
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_prep_raw2<16, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
            hipLaunchKernelGGL((k_prep_raw2<16, 2>), dim3(b2), dim3(128), 100 * 1024, st, d_points, n, (uint32_t *)pre.p, d_pts, dst0);
        } else
        if (CH == 64 && prep_nt) hipLaunchKernelGGL((k_prep_raw2<64, wpb, 1>), dim3(blocks), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)pre.p, d_pts, dst0);
        else if (CH == 32 && prep_nt) hipLaunchKernelGGL((k_prep_raw2<32, wpb, 1>), dim3(blocks), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)pre.p, d_pts, dst0);
        else if (CH == 64) hipLaunchKernelGGL((k_prep_raw2<64, wpb>), dim3(blocks), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)pre.p, d_pts, dst0);
        else if (CH == 32) hipLaunchKernelGGL((k_prep_raw2<32, wpb>), dim3(blocks), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)pre.p, d_pts, dst0);
        else if (CH == 8) hipLaunchKernelGGL((k_prep_raw2<8, wpb>), dim3(blocks), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)pre.p, d_pts, dst0);
        else if (CH == 4) hipLaunchKernelGGL((k_prep_raw2<4, wpb>), dim3(blocks), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)pre.p, d_pts, dst0);
        else hipLaunchKernelGGL((k_prep_raw2<16, wpb>), dim3(blocks), dim3(64 * wpb), 0, st, d_points, n, (uint32_t *)pre.p, d_pts, dst0);
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
static const uint64_t MSM_PASS = []() -> uint64_t {
    { const long long v = C25519_KNOB_LL("MSM_PASS_TERMS", 0); if (v >= 65536 && v <= (1ll << 22)) return (uint64_t)v; }      // A/B knob
    const int v = C25519_KNOB("MSM_PASS_LOG2", 0); if (!v) return (uint64_t)1750000; return 1ull << (v < 16 ? 16 : (v > 22 ? 22 : v)); }();
static const uint64_t MSM_PASS_MAX = MSM_PASS + MSM_PASS / 2;
static int pass_lanes() { static const int v = [] { int x = C25519_KNOB("PASS_LANES", 2); return x < 1 ? 1 : (x > 4 ? 4 : x); }(); return v; }   // A/B knob: stream sets (2, 3, 4 measure the same within 3 %: the GPU is saturated)

// the peers' streams start after everything already enqueued on the caller's stream (the inputs are complete)
int32_t passes_begin(c25519_ctx *ctx, uint64_t passes, pass_set &ps) {
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
int32_t passes_join(c25519_ctx *ctx, pass_set &ps) {
    for (int l = 1; l < ps.lanes; l++) {
        HIPCHK(hipEventRecord(ps.c[l]->ev_in, ps.c[l]->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ps.c[l]->ev_in, 0));
    }
    return C25519_OK;
}
hipEvent_t *pass_ring(c25519_ctx *owner, c25519_ctx *c, uint8_t kind) {
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
// (A third arm -- the launching pass normalising only its own points, the rest on a third stream behind the first two sorts -- was measured in rounds 5 and 6 and
//  removed: 13.68 against 13.35 - 13.53 ms, with and without streaming accesses; profiles/r05_ab_prep_split.txt, r06_ab_mall.txt.)
struct pts_ahead { uint32_t *pts; uint64_t n, offset; hipEvent_t done; bool launch; };
static int32_t msm_pass_enqueue(c25519_ctx *owner, c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, const msm_geom &g, uint64_t terms, uint32_t *d_slot,
                                hipEvent_t wait_acc, const pts_ahead *ahead = nullptr, hipEvent_t wait_in = nullptr,
                                bool cont = false, bool reduce = true, uint64_t n_carve = 0, uint32_t *d_bad_sticky = nullptr, int parity = -1) {
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
    // A continuing pass on device-resident inputs does not wait for its predecessor on this stream set with the first half of its SORT
    // (k_sweep_local, k_bin_totals: scalars in, sort scratch out -- consumed by kernels that precede it on the second stream); only the
    // second half (k_part2g on) overwrites the lists the predecessor's accumulation reads and waits for it -- or not even that, when the lists
    // exist twice (parity).  Before, the whole sort waited: its 1024-thread blocks found no room beside the OTHER set's accumulation, ended in
    // that kernel's tail, and every second accumulation started ~0.4 ms late (profiles/r04_msm_2p24_last_call_timeline.txt: gaps of
    // 35 / 420 us alternating; with the partition ahead 30 - 190 us, profiles/r04_msm_2p24_sort_ahead_timeline.txt).
    // C25519_SWEEP_EARLY: 0 = round 3's order, 1 (default) = only the partition half (k_sweep_local, k_bin_totals: sort scratch only) runs ahead,
    // 2 = all of it.  profiles/r04_ab_sort_ahead.txt: 13.39 - 13.51 ms against 13.69 - 13.71 on one box, 13.93 - 14.00 against 13.93 - 14.11 on
    // another; 1 and 2 measure the same (the accumulation beside a sort stretches by what the sort no longer costs afterwards), so the
    // default is the one without a second copy of the lists.
    static const int sweep_early = C25519_KNOB("SWEEP_EARLY", 1);
    const bool early = sweep_early && cont && !wait_in && terms > msm_small_max() && parity >= 0;
    hipEvent_t lists_free = nullptr;
    if (parity >= 0) {
        HIPCHK(hipEventRecord(ctx->ev_lists[parity & 1], ctx->stream));                 // the accumulation before this pass (list copy parity ^ 1) is behind this point
        if (early) lists_free = sweep_early >= 2 ? ctx->ev_lists[(parity & 1) ^ 1] : ctx->ev_fork;   // (copy `parity` was last read two passes ago: implied by the stream order, stated anyway)
    }
    if (!early) HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
    if (!cont && !ctx->direct_seq) slot_init(d_slot, terms, nullptr, ctx->stream, g.c);            // (a continuing pass adds its counters to the slot of its stream set; a directly published small pass writes its whole record itself)
    if (terms <= msm_small_max() && g.half <= 64 && g.nwin <= 64 && !cont && reduce && !ahead) {      // (g: a forced width may not be the small path's -- then the bucket pipeline serves)
        // the small path (small.hip): raw points as they are (projective: no normalisation, no inversion); compressed ones through the
        // decompression into records first
        if (in_fmt == C25519_FMT_RAW160) return msm_small_pass(ctx, d_scalars, d_points, 0, n, g, d_slot, ring, nullptr, wait_acc);
        if ((r = prep_points(ctx, d_points, n, in_fmt, d_pts, 0, slot_flags(d_slot) + 1))) return r;
        return msm_small_pass(ctx, d_scalars, d_pts, 1, n, g, d_slot, ring, nullptr, wait_acc);
    }
    msm_plan pl;
    pl.bad_sticky = d_bad_sticky;
    // (normalisation first, then the sort on the second stream: 2.26 against 2.34 ms at 2^21 terms the other way round)
    static const int serial_sort = C25519_KNOB("PROFILE_SERIAL_SORT", 0);     // profiling: the sort only starts after the normaliser, so that its kernels can be timed alone
    // SORT_FIRST (A/B knob): 1 = the sort is enqueued (second, high-priority stream) BEFORE the normaliser, so that its first kernel -- whose
    // 1024-thread blocks of 128 VGPRs need an empty compute unit -- is dispatched first instead of picking up the compute units the normaliser's
    // blocks leave one by one; 2 = the normaliser additionally waits for the partition half of the sort (k_sweep_local, k_bin_totals).
    // Measured and NOT adopted (profiles/r05_ab_window_groups.txt): one box 2.00 -> 1.92 ms per 2^21 terms, another 1.85 -> 1.89, four interleaved
    // repetitions on a third 1.959 (normaliser first) against 1.993; 2^20 level, 2^18 +2 %.  The default stays 0.
    static const int sort_first = C25519_KNOB("SORT_FIRST", 0);
    const bool sorted_first = sort_first && !ahead && !serial_sort && !cont && !wait_in;      // (measured on device-resident single / first passes only)
    if (sorted_first) {
        pl.ev_partition = nullptr;
        if ((r = msm_enqueue_sort(ctx, d_scalars, n, g, d_slot, ctx->aux, pl, nullptr, n_carve, lists_free, sweep_early >= 2 ? parity : -1))) return r;
        if (pl.ev_partition) HIPCHK(hipStreamWaitEvent(ctx->stream, pl.ev_partition, 0));
    }
    if (!ahead) { if ((r = prep_points(ctx, d_points, n, in_fmt, d_pts, 0, slot_flags(d_slot) + 1))) return r; }
    else if (ahead->launch) {
        if ((r = prep_points(ctx, d_points, ahead->n, in_fmt, ahead->pts, 0, slot_flags(d_slot) + 1))) return r;
        HIPCHK(hipEventRecord(ahead->done, ctx->stream));
    } else HIPCHK(hipStreamWaitEvent(ctx->stream, ahead->done, 0));
    if (serial_sort) { HIPCHK(hipEventRecord(ctx->ev_z, ctx->stream)); HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_z, 0)); }
    if (!sorted_first && (r = msm_enqueue_sort(ctx, d_scalars, n, g, d_slot, ctx->aux, pl, nullptr, n_carve, lists_free, sweep_early >= 2 ? parity : -1))) return r;
    // a continuing pass adds onto the bucket sums its predecessor on this stream set left: they must be where it left them
    if (cont && pl.buckets != ctx->cont_buckets) { ctx->err = "msm: internal error (the workspace of a continuing pass moved its buckets)"; return -(int32_t)hipErrorInvalidValue; }
    ctx->cont_buckets = pl.buckets;
    if ((r = msm_enqueue_acc(ctx, pl, d_pts, d_slot, ring, wait_acc, cont, reduce, d_bad_sticky))) return r;
    return C25519_OK;
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
        slot_init(d_record, 0, nullptr, ctx->stream, 0);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        return C25519_OK;
    }
    const uint64_t PT = pass_terms ? pass_terms : MSM_PASS, PTMAX = pass_terms ? pass_terms + pass_terms / 2 : MSM_PASS_MAX;
    const uint64_t passes = n <= PTMAX ? 1 : (n + PT - 1) / PT, per = (n + passes - 1) / passes;
    const size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    msm_geom g;
    pass_set ps;
    int32_t r;
    uint32_t *sticky = (uint32_t *)ctx->d_flag + 40;      // "a scalar has bit 255 set", ORed over the passes of the call
    // (r5) A small call on raw points whose caller reads the record on the host right away (ctx->want_direct, d_record = the context's own record): the small
    // path publishes the record itself -- no sticky-word memset and no k_slot_init before it, no copy launch behind it (small.hip small_direct):
    // 112 -> ~65 us for a 1-term call (profiles/r05_small_call_phases.txt).  A/B knob SMALL_DIRECT = 0: rounds 4's five launches.
    static const int small_direct_knob = C25519_KNOB("SMALL_DIRECT", 1);
    ctx->direct_seq = 0;
    ctx->direct_extra = nullptr;
    // (r6) the mid path (mid.hip: 12 288 .. 2^18 terms in four launches on this stream) publishes its record the same way
    bool mid = false;
    // (a host-pointer call of ONE pass has nothing to overlap its upload with: the inputs go up in one piece and the pass waits for them)
    // (late) ENCODED points -- the decompression's affine records, the 7 M accumulation -- leave the small path earlier than raw ones, like verify_batch's: from
    // verify_small_max() + 1 = 4096 terms, with 12-bit windows up to msm_small_max() (profiles/r06_ab_small_mid_boundary.txt, CompressedEdwardsY, device-resident: 4096 terms
    // 0.269 -> 0.236 ms, 6143 0.287 -> 0.244; 3072: 0.243 against 0.235, left alone)
    const bool enc_mid = in_fmt != C25519_FMT_RAW160 && n > verify_small_max() && n <= msm_small_max();
    if (passes == 1 && (n > msm_small_max() || enc_mid)) { msm_layout(n, g, 0, enc_mid ? 12 : 0); mid = msm_mid_serves(n, g, in_fmt != C25519_FMT_RAW160); }
    if (small_direct_knob && ctx->want_direct && !ctx->no_direct_once && d_record == drec(ctx) && in_fmt == C25519_FMT_RAW160 && ((!fetch && n <= msm_small_max()) || mid)) {
        msm_geom gs;
        msm_layout(n, gs);
        if (mid || (gs.half <= 64 && gs.nwin <= 64)) { ctx->direct_seq = ++ctx->publish_seq; if (!ctx->direct_seq) ctx->direct_seq = ++ctx->publish_seq; }
    }
    ctx->want_direct = false;
    ctx->no_direct_once = false;
    if (ctx->direct_seq && !mid) {
        // (r6) the directly published small call, lean: its two kernels and the bracket of the call -- no phase ring, no fork to the second stream, no pass bookkeeping
        // (eight event records and a cross-stream wait that a 75 us call spent ~8 us of host time on; c25519_last_call_phase_ms answers -1 for such a call)
        msm_geom gs;
        msm_layout(n, gs);
        ctx->last_passes.clear();
        ctx->solo = true;
        ctx->coarse_wait = nullptr;
        ctx->kname[0] = "c25519::k_small_cols (tables of multiples by repeated addition, one lane per (window, term))";
        if ((r = msm_small_enqueue(ctx, d_scalars, d_points, 0, n, gs, d_record, ctx->stream))) { ctx->direct_seq = 0; return r; }
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        return C25519_OK;
    }
    if (mid) {
        ctx->last_passes.clear();
        ctx->solo = true;
        ctx->coarse_wait = nullptr;
        // (no phase ring for the MSM's own mid-size calls: an event record between two kernels is a ~5 us gap on the GPU -- the lean small path got 30 us of GPU span
        //  back from eight of them; c25519_last_call_phase_ms answers -1, c25519_last_kernel_ms still brackets the call)
        hipEvent_t *ring = nullptr;
        if (fetch) {
            hipEvent_t in_ev = nullptr;
            if ((r = (*fetch)(0, n, &in_ev))) { ctx->direct_seq = 0; return r; }
            HIPCHK(hipStreamWaitEvent(ctx->stream, in_ev, 0));
        }
        if (in_fmt == C25519_FMT_RAW160) r = msm_mid_enqueue(ctx, d_scalars, d_points, 0, n, g, d_record, 1, n, ring);
        else {
            // encodings: the slot first (the decompression counts what does not decode into it), then the records, then the pass over them
            ctx->direct_seq = 0;
            slot_init(d_record, n, nullptr, ctx->stream, g.c);
            if ((r = ctx_reserve(ctx, ctx->tmp_e, n * PTS_BYTES + 256))) return r;
            if ((r = prep_points(ctx, d_points, n, in_fmt, (uint32_t *)ctx->tmp_e.p, 0, slot_flags(d_record) + 1))) return r;
            r = msm_mid_enqueue(ctx, d_scalars, ctx->tmp_e.p, 1, n, g, d_record, 0, n, ring);
        }
        if (r) { ctx->direct_seq = 0; return r; }
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        return C25519_OK;
    }
    // (r5: letting the sort start on the second stream without the cross-stream wait when the main stream is idle measured level at 2^14 .. 2^20 terms --
    //  the ~20 us before the first kernels is launch latency, not the event: profiles/r05_ab_midrange_streams.txt; not kept)
    if (!ctx->direct_seq) HIPCHK(hipMemsetAsync(sticky, 0, 4, ctx->stream));
    if ((r = passes_begin(ctx, passes, ps))) return r;
    // One layout for every pass: their column sums add up window by window.  It is derived from the terms of a pass -- or, when stream sets
    // get more than one pass each, from at least 2^21: their passes continue each other's bucket sums, ONE reduction serves all of them, and
    // the wider window (17 bits: 15 additions per term instead of 16) costs what a single pass of 2^21 terms pays for it.  The record header
    // carries the number the layout was derived from (records_fold re-derives it from there).
    const uint64_t layout_terms = (passes > (uint64_t)ps.lanes && per >= (1ull << 20)) ? std::max<uint64_t>(per, 1ull << 21) : per;
    msm_layout(layout_terms, g);
    // A/B knobs (tuning build): window groups of a single-pass call (msm_geom / msm_enqueue_acc), ACC_LAST = content windows of the final group
    // Measured and NOT adopted (profiles/r05_ab_window_groups.txt): with two groups the exposed tail of a 2^21-term call shrinks from 0.26 to 0.18 ms, but
    // the reduction of group 0 takes its issue slots and registers from the second group's accumulation (k_accumulate 1.16 -> 1.17 - 1.26 ms over the two
    // launches, each with its own ramp-down): four interleaved repetitions on one box give 1.959 ms (one group) against 1.965 (two); three and four
    // groups 2.06 / 2.18; at 2^20 and 2^18 terms two groups lose 8 % and 17 %.  The default stays ONE group.
    static const int acc_groups = std::min(2, C25519_KNOB("ACC_GROUPS", 1)), acc_last = 0;      // (three / four groups and an uneven last group lost twice and left the build: profiles/r05_ab_window_groups.txt)
    // (groups exist in the chunk-local sort only: single passes below its lower boundary -- the digit-matrix sort's range -- keep one group)
    static const uint64_t chunk_local_min = (uint64_t)C25519_KNOB("SORT_CHUNK_LOCAL_MIN", 1 << 16);
    if (passes == 1 && n > msm_small_max() && n >= chunk_local_min) msm_set_groups(g, acc_groups, acc_last);
    ctx->solo = passes == 1;               // (overwritten by every call: an early error return leaves nothing behind that a later call would read)
    hipEvent_t prev_acc = nullptr;                         // the accumulation of the previous pass (on the other stream set)
    hipEvent_t prev_acc_last = nullptr;
    // raw points, several passes on two stream sets: pass 1 (the first one on the peer) prepares the records of ALL later
    // passes in one launch beside the sort and the accumulation of pass 0 (pts_ahead; 128 bytes per point stay allocated)
    // (not with a fetch: the later passes' points are not on the device yet)
    static const int prep_ahead = C25519_KNOB("PREP_AHEAD", 1);      // A/B knob: 0 = every pass normalises its own points (on its main stream, ahead of its accumulation)
    const bool ahead = prep_ahead && passes > 1 && ps.lanes > 1 && in_fmt == C25519_FMT_RAW160 && !fetch;
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
        if ((r = msm_pass_enqueue(ctx, c, d_scalars + lo * 32, d_points + lo * psz, m, in_fmt, g, layout_terms, slot, prev_acc, (ahead && p >= 1) ? &ah : nullptr, in_ev,
                                  !first, last, per, sticky, passes > (uint64_t)L ? (int)((p / L) & 1) : -1))) {
            if (ctx->err.empty()) ctx->err = c->err;
            return r;
        }
        prev_acc = L > 1 ? c->ev_acc : nullptr;
        prev_acc_last = c->ev_acc;
    }
    // a long call: the host blocks on the end of the LAST accumulation before it polls for the published results (publish_and_wait)
    ctx->coarse_wait = (n >= (1ull << 17) && prev_acc_last) ? prev_acc_last : nullptr;
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
    ctx->host_us[0] = ctx->host_us[1] = wall_us();
    ctx->want_direct = true;
    int32_t r = msm_record_enqueue(ctx, d_scalars, d_points, n, in_fmt, drec(ctx));
    if (r) { ctx->direct_seq = 0; return r; }
    r = rec_collect(ctx);
    if (r == C25519_LOST_PUBLICATION) {                     // (r6) never observed; see wait_published: once more through the slot + copy path
        ctx->no_direct_once = true;
        if ((r = msm_record_enqueue(ctx, d_scalars, d_points, n, in_fmt, drec(ctx)))) return r;
        r = rec_collect(ctx);
        if (!r) ctx->err.clear();
    }
    if (r) return r;
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
    f[REC_TERMS_LO] = 1; f[REC_PASSES] = 1; f[REC_MAGIC] = REC_MAGIC_VALUE; f[REC_C] = (uint32_t)g.c;
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
    ctx->host_us[4] = wall_us();
    return C25519_OK;
}
EXPORT int32_t c25519_msm_vartime(c25519_ctx *ctx, const uint8_t *scalars, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out) {
    HIPCHK(hipSetDevice(ctx->device));
    if (out_fmt < 0 || out_fmt > 2 || in_fmt < 0 || in_fmt > 2) { ctx->err = "msm: bad format"; return -(int32_t)hipErrorInvalidValue; }
    const size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    int32_t r;
    ctx->host_us[0] = wall_us();
    if (n <= msm_small_max()) {
        // the reference's own benchmark sizes (1 .. 1024 terms) and everything else small.hip serves: the inputs into the page-locked staging buffer (raw points: read
        // there in place by the kernels; encodings: one staged copy up), the kernels, the record published to the host by the last of them
        const void *src[2] = {scalars, points};
        const size_t bytes[2] = {(size_t)n * 32, (size_t)n * psz};
        uint8_t *d[2];
        // raw points on the directly published small path are read in place from the staging buffer (A/B knob ZERO_COPY_MAX: 0 = always upload)
        static const int zero_copy_max = C25519_KNOB("ZERO_COPY_MAX", 4095);
        const bool zc = in_fmt == C25519_FMT_RAW160 && n <= (uint64_t)zero_copy_max && C25519_KNOB("SMALL_DIRECT", 1);
        if ((r = ffi_small_upload(ctx, 2, src, bytes, d, 0, zc))) return r;
        ctx->host_us[1] = wall_us();
        ge_p3 R;
        uint32_t flags[8];
        ctx->want_direct = true;
        r = msm_record_enqueue(ctx, d[0], d[1], n, in_fmt, drec(ctx));
        if (!r) r = rec_collect(ctx);
        else { ctx->direct_seq = 0; (void)hipStreamSynchronize(ctx->stream); }          // (the upload may still be reading the staging buffer)
        if (r == C25519_LOST_PUBLICATION) {                 // (r6) never observed; see wait_published: once more through the slot + copy path (the staged inputs are still there)
            ctx->no_direct_once = true;
            r = msm_record_enqueue(ctx, d[0], d[1], n, in_fmt, drec(ctx));
            if (!r) r = rec_collect(ctx);
            if (!r) ctx->err.clear();
        }
        ffi_small_end(ctx, n * (32 + psz), 0);
        if (r) return r;
        if ((r = records_fold((const uint8_t *)hslot(ctx, C25519_MAX_SLOTS), 1, R, flags, &ctx->err))) return r;
        if ((r = msm_record_status(ctx, flags))) return r;
        host_encode(R, out_fmt, out);
        ctx->host_us[4] = wall_us();
        return C25519_OK;
    }
    ctx->host_us[1] = ctx->host_us[0];
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
    ctx->want_direct = true;                               // (the mid path publishes its record itself; the bucket pipeline ignores the wish)
    r = msm_record_enqueue(ctx, d_s, d_p, n, in_fmt, drec(ctx), &fetch, pass_terms);
    if (!r) r = rec_collect(ctx);
    if (r == C25519_LOST_PUBLICATION) {                     // (never observed; see wait_published: once more through the slot + copy path -- the inputs are on the device)
        ctx->no_direct_once = true;
        r = msm_record_enqueue(ctx, d_s, d_p, n, in_fmt, drec(ctx), nullptr, 0);
        if (!r) r = rec_collect(ctx);
        if (!r) ctx->err.clear();
    }
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
// launchers of the driver's small kernels for the other translation units (verify.hip, small.hip)
void launch_prep_basepoint(uint32_t *pts, uint64_t dst, hipStream_t st) { hipLaunchKernelGGL(k_prep_basepoint, dim3(1), dim3(64), 0, st, pts, dst); }
void launch_record_sum(uint32_t *rec, const uint32_t *slots, int cnt, int nwin, int first, hipStream_t st) { hipLaunchKernelGGL(k_record_sum, dim3(1), dim3(128), 0, st, rec, slots, cnt, nwin, first); }
