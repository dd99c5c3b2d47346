// Small multiscalar multiplications -- up to msm_small_max() = 6143 terms (round 4: 4095; round 5 .. late round 6: 12 287, until the mid path of mid.hip was measured
// against it: msm_internal.h; verify_batch up to 2047 signatures): the reference's own benchmark shapes (benches/dalek_benchmarks.rs:16
// MULTISCALAR_SIZES 1 .. 1024; ed25519_benchmarks.rs:53 verify_batch of 4 .. 256 signatures = 9 .. 513 terms) and everything the reference
// hands to Straus (edwards.rs:1025, below 190 terms).
//
// The bucket pipeline of msm.hip is built for millions of terms: fifteen launches (normalise, four sort kernels, bucket order, accumulate,
// long buckets, two reduction levels, ...) whose fixed cost -- ~0.4 ms, most of it launch gaps and single-wave latency chains, the batched
// inversion of the normaliser alone ~0.1 ms -- is all a 100-term call pays for.  What Straus does on a CPU (straus.rs:159-200: a small table
// of multiples per point, then ONE shared doubling chain) becomes, on a machine with 250 000 lanes and a 0.5 us field multiplication
// latency per lone wave, "no chain at all":
//
//   k_small_cols    a block owns 4 terms.  It builds their tables of multiples {1 .. 2^(c-1)} P in LDS by REPEATED COMPLETE ADDITION in c - 1
//                   rounds (round r: E[2^r + j] = E[2^r] + E[j], j = 1 .. 2^r -- edwards.rs:795 is complete, so the same code doubles), straight
//                   from the raw projective point: no normalisation, no inversion.  Then thread (window k, term i) looks its signed digit
//                   up -- the SAME window layout as the bucket pipeline (msm_layout: c = 5 below 1024 terms, 6 from there; round 4: 7 from 2048), so the
//                   column sums are a partial-result record like any other -- and the four terms of a window are added across the
//                   lanes of a quad.  Chain: c - 1 + 2 additions.
//   k_small_reduce  one block per window: the <= 1024 block partials are added in a shuffle / LDS tree.  Chain: <= 12 additions.
//
// and the Horner fold over the windows stays where it always was (host, msm_horner).  Work is n (2^(c-1) + 2 nwin) additions instead of the
// bucket method's n nwin / 2 -- irrelevant while the GPU is latency-bound, not throughput-bound: up to ~14 000 terms with 6-bit windows (20 KB of tables per
// block, eight blocks per compute unit; the call grows by ~17 ns per term and meets the bucket pipeline's ~0.33 ms there: profiles/r05_ab_small_path_range.txt --
// round 4 stopped at 4095 terms because its 7-bit tables, 40 KB per block, were already losing: 0.181 against 0.156 ms at 4095 terms).  Variable time like the path it
// replaces (digits index the table): vartime_multiscalar_mul and verify_batch only; the constant-time MSM is extra.hip's.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string>
#include "../../include/c25519_hip.h"
#include "devio.h"
#include "ctx.h"
#include "msm_internal.h"
#include "msm_sort.h"

using namespace c25519;
#define HIPCHK(call)                                                \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return c25519_fail(ctx, _e, #call);   \
    } while (0)

namespace c25519 {

constexpr int SMALL_T = 4;                   // terms per block: thread = (window slot, term), the four terms of a window are one quad
constexpr int SMALL_SLOTS = 64;              // window slots per block (>= MSM_MAX_WIN = 56 windows)
constexpr int SMALL_THREADS = SMALL_T * SMALL_SLOTS;

__device__ __forceinline__ void p3_to_lds(u32 *dst, const ge_p3 &p) {
    for (int i = 0; i < 10; i++) { dst[i] = p.X.v[i]; dst[10 + i] = p.Y.v[i]; dst[20 + i] = p.Z.v[i]; dst[30 + i] = p.T.v[i]; }
}
__device__ __forceinline__ ge_p3 p3_from_lds(const u32 *src) {
    ge_p3 p;
    for (int i = 0; i < 10; i++) { p.X.v[i] = src[i]; p.Y.v[i] = src[10 + i]; p.Z.v[i] = src[20 + i]; p.T.v[i] = src[30 + i]; }
    return p;
}
__device__ __forceinline__ ge_p3 p3_quad_xor(const ge_p3 &a, int d) {
    ge_p3 o;
    for (int i = 0; i < 10; i++) {
        o.X.v[i] = __shfl_xor(a.X.v[i], d, 64); o.Y.v[i] = __shfl_xor(a.Y.v[i], d, 64);
        o.Z.v[i] = __shfl_xor(a.Z.v[i], d, 64); o.T.v[i] = __shfl_xor(a.T.v[i], d, 64);
    }
    return o;
}
// (y+x, y-x, 2dxy) of an affine point -> the same point as (2x : 2y : 2 : 2xy): one multiplication by 1/d, no halving
__device__ __forceinline__ ge_p3 p3_from_aniels(const ge_aniels &a) {
    ge_p3 p;
    p.X = fe_carry(fe_sub(a.ypx, a.ymx));
    p.Y = fe_carry(fe_add(a.ypx, a.ymx));
    p.Z = fe_small(2);
    p.T = fe_mul(a.xy2d, fe_d_inv());
    return p;
}

// (Measured and dropped in round 6, profiles/r06_small_call_phases.txt: 5 .. 16 terms as four groups of four in ONE 1024-thread block, the groups' sums added through LDS, so
//  that such a call has no k_small_reduce launch -- 128 VGPRs + 18 spilled words at sixteen waves per compute unit: a 16-term call 124 against 70 us, verify_batch of 2 .. 7
//  signatures 192 against 146 - 150 us.  Four blocks on four compute units and a second launch are the faster form.)
// (r5) DIRECT publication of a small call's record (small_direct.on): the kernels need no cleared slot before them -- the "bit 255" flag travels as one word
// per block (blockflags) instead of an atomicOr into the slot -- and the LAST block to finish writes the 16 header / counter words and then releases `seq` into the
// host's sequence word, with the column sums already written to `cols` in page-locked, coherent host memory: the host polls that word instead of launching a
// copy (msm.hip rec_collect).  done_cnt: a device word that counts the finished blocks of k_small_reduce.  (r6) It is zeroed by k_small_cols -- the kernel that
// ALWAYS precedes k_small_reduce on the stream -- and no longer by the last block of the previous call: a call that died between the two kernels (a fault, a
// context error) used to leave the word non-zero, and every later direct call on that context then never published (round-5 advice).
// extra (verify_batch's small path, may be null): two device words -- keys / R_i that did not decode, left by the decompression kernel ahead on the stream --
// published as the record's counters [2] and [3].
struct small_direct { int on; u32 *blockflags; u32 *done_cnt; u32 *host_flag; u32 seq; u32 terms; u32 c; const u32 *extra; };
__device__ __forceinline__ void small_publish(u32 *cols, const small_direct &dx, u32 bad) {      // one thread, after every column is written and fenced
    u32 *f = cols + MSM_MAX_WIN * 40;
    for (int i = 0; i < 16; i++) f[i] = 0;
    f[0] = bad; f[REC_TERMS_LO] = dx.terms; f[REC_PASSES] = 1; f[REC_MAGIC] = REC_MAGIC_VALUE; f[REC_C] = dx.c;
    if (dx.extra) { f[2] = dx.extra[0]; f[3] = dx.extra[1]; }
    __threadfence_system();
    __hip_atomic_store(dx.host_flag, dx.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// grid: ceil(n / SMALL_T) blocks of SMALL_THREADS threads; dynamic LDS: SMALL_T * half * 160 bytes.
// partial: [block][window] 160-byte sums (or, with a single block, the column sums themselves)
__global__ void __launch_bounds__(SMALL_THREADS) k_small_cols(const uint8_t *__restrict__ scalars, const void *__restrict__ points, int src_fmt, u64 n, msm_geom g,
                                                              u32 *__restrict__ partial, u32 *__restrict__ flags, small_direct dx) {
    extern __shared__ u32 tab[];                          // [SMALL_T][half][40]
    __shared__ u32 sk[SMALL_T * 8];                       // the block's four scalars: read ONCE (r5: every (window, term) thread used to load its term's 32 bytes
    //                                                       itself -- 64 loads of the same words, which matters when the inputs are read in place from host memory)
    // (Measured and dropped, profiles/r05_ab_small_path_range.txt: numbering the threads ROTATED by whole waves with the block index -- on the suspicion that wave 0
    //  of every block, the only one busy in the table phase, shares one SIMD with the wave 0s of its neighbours -- is 5 - 9 % SLOWER from 4096 terms, level below.)
    const int tid = threadIdx.x, ti = tid & (SMALL_T - 1), slot = tid / SMALL_T;
    if (dx.on && blockIdx.x == 0 && tid == SMALL_THREADS - 1 && gridDim.x > 1) *dx.done_cnt = 0;      // (k_small_reduce starts after this kernel has ended)
    if (tid < SMALL_T * 8) {
        const u64 tt = (u64)blockIdx.x * SMALL_T + (u64)(tid >> 3);
        sk[tid] = tt < n ? reinterpret_cast<const u32 *>(scalars)[tt * 8 + (tid & 7)] : 0u;
    }
    const u64 t = (u64)blockIdx.x * SMALL_T + ti;
    const int half = g.half;
    // ---- tables: E[1] = P, then c - 1 rounds of complete additions -----------------------------------------------------------------
    if (tid < SMALL_T) {
        const u64 tt = (u64)blockIdx.x * SMALL_T + tid;
        ge_p3 P = ge_identity();
        if (tt < n) {
            if (src_fmt == 0) {
                // like the normaliser of the bucket pipeline, the small path reads X, Y, Z only: (XZ : YZ : Z^2 : XY) is the same point with a T
                // that is consistent by construction, whatever the caller stored there
                const uint8_t *in = (const uint8_t *)points;
                const feT X = raw160_fe(in, tt, 0), Y = raw160_fe(in, tt, 1), Z = raw160_fe(in, tt, 2);
                P.X = fe_mul(X, Z); P.Y = fe_mul(Y, Z); P.Z = fe_sq(Z); P.T = fe_mul(X, Y);
            } else P = p3_from_aniels(pts_load((const u32 *)points, tt));
        }
        p3_to_lds(tab + ((size_t)tid * half + 0) * 40, P);
    }
    __syncthreads();
    for (int span = 1; span < half; span <<= 1) {         // E[span + 1 + j] = E[span] + E[1 + j]  (entries are stored at index multiple - 1)
        const int pairs = SMALL_T * span;
        if (tid < pairs) {
            const int i = tid / span, j = tid % span;
            const u32 *row = tab + (size_t)i * half * 40;
            const ge_p3 s = ge_add(p3_from_lds(row + (size_t)(span - 1) * 40), p3_from_lds(row + (size_t)j * 40));
            p3_to_lds(tab + ((size_t)i * half + span + j) * 40, s);
        }
        __syncthreads();
    }
    // ---- digits of this thread's (window, term) -----------------------------------------------------------------------------------------
    int d = 0;
    int bad = 0;
    if (slot < g.nwin && t < n) {
        u32 s[9];
        for (int i = 0; i < 8; i++) s[i] = sk[ti * 8 + i];        // (written before the barriers of the table rounds above)
        if ((s[7] >> 31) && slot == 0) { if (dx.on) bad = 1; else atomicOr(flags, 1u); }
        u64 carry = 0;
        for (int i = 0; i < 8; i++) { const u64 v = (u64)s[i] + g.addk[i] + carry; s[i] = (u32)v; carry = v >> 32; }
        s[8] = (u32)carry;
        const int bit = g.pos[slot], wi = bit >> 5, sh = bit & 31;
        u32 lo = 0, hi = 0;
        for (int i = 0; i < 9; i++) { lo = i == wi ? s[i] : lo; hi = i == wi + 1 ? s[i] : hi; }      // (no dynamic register index)
        const u64 two = (u64)lo | ((u64)hi << 32);
        d = digit_of((u32)(two >> sh) & ((1u << g.wid[slot]) - 1u), slot, g);
    }
    ge_p3 Q = ge_identity();
    if (d != 0) {
        Q = p3_from_lds(tab + ((size_t)ti * half + (d > 0 ? d : -d) - 1) * 40);
        if (d < 0) Q = ge_neg(Q);
    }
    // the four terms of a window: lanes 4 slot .. 4 slot + 3
    Q = ge_add(Q, p3_quad_xor(Q, 1));
    Q = ge_add(Q, p3_quad_xor(Q, 2));
    if (ti == 0 && slot < g.nwin) p40_store(partial, (u64)blockIdx.x * g.nwin + slot, Q);
    if (dx.on) {
        if (gridDim.x == 1) __threadfence_system();           // (a single block writes the host's columns itself; a system-scope fence in each of a thousand
        //                                                           blocks of a 4000-term call cost 100 us: first version of this path)
        const int any_bad = __syncthreads_or(bad);
        if (tid == 0) {
            if (gridDim.x == 1) small_publish(partial, dx, (u32)any_bad);
            else dx.blockflags[blockIdx.x] = (u32)any_bad;
        }
    }
}

// one block per window: col_k = sum over the blocks' partials.  TH threads: every thread adds nblocks / TH partials one after the other, then a shuffle tree
// inside each wave and a second one over the waves' sums (round 5; before, thread 0 added the waves' sums one after the other: 4096 terms 0.158 -> 0.151 ms).
// Measured and dropped (profiles/r05_ab_small_path_range.txt): 512 threads above 256 partials -- a shorter chain on paper (12 000 terms: 14 links instead of
// 20), 3 - 10 % SLOWER at every size from 2048 terms.
template <int TH>
__global__ void __launch_bounds__(TH) k_small_reduce(const u32 *__restrict__ partial, int nblocks, int nwin, u32 *__restrict__ cols, small_direct dx) {
    __shared__ u32 stage[(TH / 64) * 40];
    __shared__ int last;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    ge_p3 acc = ge_identity();
    bool any = false;
    for (int b = tid; b < nblocks; b += TH) {
        const ge_p3 v = p40_load(partial, (u64)b * nwin + k);
        acc = any ? ge_add(acc, v) : v;
        any = true;
    }
    // shuffle tree over the lanes that hold something (wave-uniform trip count)
    const int active = nblocks < TH ? nblocks : TH;
    const int in_wave = active - 64 * w < 64 ? active - 64 * w : 64;
    for (int dd = 1; dd < in_wave; dd <<= 1) {
        const ge_p3 o = p3_quad_xor(acc, dd);                 // (lanes beyond `active` hold the identity)
        acc = ge_add(acc, o);
    }
    if (active > 64) {
        if (lane == 0) p3_to_lds(stage + w * 40, acc);
        __syncthreads();
        if (w == 0) {                                         // the waves' sums: a second tree, in wave 0
            const int waves = (active + 63) / 64;
            acc = lane < waves ? p3_from_lds(stage + lane * 40) : ge_identity();
            for (int dd = 1; dd < waves; dd <<= 1) acc = ge_add(acc, p3_quad_xor(acc, dd));
        }
    }
    if (tid == 0) p40_store(cols, k, acc);
    if (dx.on) {
        // the block that finishes LAST publishes: every block fences its column to the system and then counts itself
        if (tid == 0) { __threadfence_system(); last = atomicAdd(dx.done_cnt, 1u) == (u32)nwin - 1u; }
        __syncthreads();
        if (last) {
            int b = 0;
            for (int i = tid; i < nblocks; i += TH) b |= (int)dx.blockflags[i];
            const int any_bad = __syncthreads_or(b);
            if (tid == 0) small_publish(cols, dx, (u32)any_bad);
        }
    }
}

}  // namespace c25519

int32_t msm_small_enqueue(c25519_ctx *ctx, const uint8_t *d_scalars, const void *d_points, int src_fmt, uint64_t n, const msm_geom &g, uint32_t *d_slot, hipStream_t st) {
    if (n == 0 || n > msm_small_max() || g.half > 64 || g.nwin > SMALL_SLOTS) { ctx->err = "msm: internal error (small path outside its range)"; return -(int32_t)hipErrorInvalidValue; }
    const int nblocks = (int)((n + SMALL_T - 1) / SMALL_T);
    const size_t lds = (size_t)SMALL_T * g.half * 160;
    // direct publication (ctx->direct_seq, set by msm_record_enqueue): the record goes to the host's slot, not to d_slot
    small_direct dx = {0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr};
    uint32_t *out = d_slot;
    if (ctx->direct_seq) {
        out = ctx->hd_msm + (size_t)C25519_MAX_SLOTS * C25519_SLOT_U32;
        dx.on = 1; dx.done_cnt = (uint32_t *)ctx->d_flag + 56; dx.host_flag = ctx->hd_msm + (size_t)(C25519_MAX_SLOTS + 1) * C25519_SLOT_U32;
        dx.seq = ctx->direct_seq; dx.terms = (uint32_t)n; dx.c = (uint32_t)g.c; dx.extra = ctx->direct_extra;
        // fault injection (TUNING build only; the release library compiles this to nothing): every FAULT_LOSE_PUBLICATION-th directly published call releases a
        // WRONG sequence number, so that the host's recovery -- wait_published phase 3 and the re-run through the copy path -- is exercised by a test
        static const int lose_every = C25519_KNOB("FAULT_LOSE_PUBLICATION", 0);
        const uint64_t nth = ++ctx->counters[C25519_CTR_PUBLISH_DIRECT];
        if (lose_every > 0 && nth % (uint64_t)lose_every == 0) dx.seq ^= 0x40000000u;
    }
    uint32_t *partial = out;                                        // a single block writes the column sums themselves
    if (nblocks > 1) {
        int32_t r = ctx_reserve(ctx, ctx->tmp_d, (size_t)nblocks * g.nwin * 160 + (size_t)nblocks * 4 + 512);
        if (r) return r;
        partial = (uint32_t *)ctx->tmp_d.p;
        dx.blockflags = partial + (size_t)nblocks * g.nwin * 40 + 16;
    }
    hipLaunchKernelGGL(k_small_cols, dim3(nblocks), dim3(SMALL_THREADS), lds, st, d_scalars, d_points, src_fmt, n, g, partial, slot_flags(d_slot), dx);
    if (nblocks > 1) hipLaunchKernelGGL(k_small_reduce<256>, dim3(g.nwin), dim3(256), 0, st, partial, nblocks, g.nwin, out, dx);
    HIPCHK(hipGetLastError());
    return C25519_OK;
}
