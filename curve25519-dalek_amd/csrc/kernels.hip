// HIP kernels for gfx950 (MI355X): one curve point / scalar / ladder per lane.
// Launchers at the bottom are the only symbols the host side (capi.hip) uses.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#define C25519_CHAIN 1   // chained-carry fe_mul / fe_sq (fe26.h): +4 % on the comb and the ladder
#include "devio.h"
#include "kernels.h"
#include "fe26x.h"
#include "knobs.h"
#ifndef C25519_PREPC_ATTR
#define C25519_PREPC_ATTR
#endif

namespace c25519 {

// ================================================================================================
// K2  fixed-base batch: s*B with a per-window table of affine Niels multiples staged in LDS.
//     Algorithm of EdwardsBasepointTable::mul_base (edwards.rs:1192-1209) with the window count
//     re-derived for the GPU: one table per window position (no doublings at all), signed
//     radix-2^W digits (scalar.rs:1093-1150 recentering), W = 6 -> 43 mixed additions per scalar
//     against the reference's 64 additions + 4 doublings at radix 16.
//     Table layout (global and LDS): [NWIN][HALF+1] entries x 6 uint4; entry j of window i is
//     j * 2^(W*i) * B as canonical (y+x, y-x, 2dxy) 3 x 32 bytes; entry 0 is the identity.
// ================================================================================================
//     CT = true is the constant-time form for SECRET scalars: the reference's LookupTable::select (window.rs:54-76)
//     -- every entry of the window's table is read (one LDS address for the whole wave: a broadcast) and the wanted one is
//     kept with v_cndmask, so neither an address nor a branch depends on the scalar; the recoding and the conditional
//     negation are already branch-free.  Cost: (2^(W-1) + 1) x 24 selects per window beside the 7 M addition.
// OUT: 0 = P32 scratch record (X,Y,Z) for the batched compressor, 1 = raw 160-byte point, 2 = P40 (tight limbs)
template <int W, int BS, int OUT, bool CT>
__global__ void __launch_bounds__(BS) k_mul_base(const uint8_t *__restrict__ scalars, u64 n,
                                                 const uint4 *__restrict__ gtab, u32 *__restrict__ scratch,
                                                 uint8_t *__restrict__ out_raw) {
    constexpr int NWIN = (256 + W - 1) / W, HALF = 1 << (W - 1), ENT = HALF + 1;
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < NWIN * ENT * 6; i += BS) lds[i] = gtab[i];
    __syncthreads();
    for (u64 idx = (u64)blockIdx.x * BS + threadIdx.x; idx < n; idx += (u64)gridDim.x * BS) {
        u32 s[8];
        load8(scalars, idx, s);
        ge_p3 P = ge_identity();
        u32 carry = 0;
        const uint4 *wtab = lds;
#pragma unroll 1
        for (int win = 0; win < NWIN; win++) {
            u32 d = (s[0] & (2u * HALF - 1u)) + carry;
#pragma unroll
            for (int i = 0; i < 7; i++) s[i] = (s[i] >> W) | (s[i + 1] << (32 - W));
            s[7] >>= W;
            // recentre to [-HALF, HALF) except in the top window (scalar.rs:1136-1147)
            bool neg = (win != NWIN - 1) && (d >= (u32)HALF);
            u32 mag = neg ? 2u * HALF - d : d;
            carry = neg ? 1u : 0u;
            u32 tw[24];
            if (CT) {
                // window.rs:54-76 on the GPU: every entry of the window is READ (wave-uniform LDS addresses) and the wanted
                // one kept by selects.  The reads are pinned: without the asm LLVM sinks them under `hit` -- an exec-masked
                // read behind s_cbranch_execz, i.e. a branch on whether any lane of the wave has this digit (found in round
                // 2; tests/test_ct_isa.py asserts the instruction stream: 6 ds_read_b128 + 24 v_cndmask per entry, no
                // exec-mask branch).  (The compare and the selects as one asm block with the condition in VCC: 3.04 ms
                // against 2.81 for this form; unrolling the entry loop by 2 / 4: 2.84 / 2.83.)
#pragma unroll
                for (int i = 0; i < 24; i++) tw[i] = 0;
                tw[0] = 1; tw[8] = 1;                                   // entry 0 is the identity (y+x = y-x = 1, 2dxy = 0): not scanned
#pragma unroll 1
                for (int ent = 1; ent < ENT; ent++) {
                    const uint4 *e = wtab + ent * 6;                    // wave-uniform address
                    const bool hit = (u32)ent == mag;
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        uint4 v = e[i];
                        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
                        tw[4 * i] = hit ? v.x : tw[4 * i]; tw[4 * i + 1] = hit ? v.y : tw[4 * i + 1];
                        tw[4 * i + 2] = hit ? v.z : tw[4 * i + 2]; tw[4 * i + 3] = hit ? v.w : tw[4 * i + 3];
                    }
                }
            } else {
                const uint4 *e = wtab + mag * 6;
                uint4 q0 = e[0], q1 = e[1], q2 = e[2], q3 = e[3], q4 = e[4], q5 = e[5];
                u32 t2[24] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w,
                              q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w};
#pragma unroll
                for (int i = 0; i < 24; i++) tw[i] = t2[i];
            }
#if defined(__HIP_DEVICE_COMPILE__) && defined(C25519_CT_LOCKSTEP)
            // A/B arm: the seven products of the addition in lockstep (fe26x.h, the form k_accumulate uses), the sign by operand swap
            if (CT) P = ge_madd_signed_p3_lockstep(P, aniels_from_words(tw), neg);
            else { aniels_words_cneg(tw, neg); P = ge_p1p1_to_p3(ge_madd(P, aniels_from_words(tw))); }
#else
            aniels_words_cneg(tw, neg);
            P = ge_p1p1_to_p3(ge_madd(P, aniels_from_words(tw)));
#endif
            wtab += ENT * 6;
        }
        if (OUT == 1) raw160_store(out_raw, idx, P);
        else if (OUT == 2) {
            uint4 *q = reinterpret_cast<uint4 *>(scratch) + 10 * idx;
            u32 t[40];
            for (int i = 0; i < 10; i++) { t[i] = P.X.v[i]; t[10 + i] = P.Y.v[i]; t[20 + i] = P.Z.v[i]; t[30 + i] = P.T.v[i]; }
            for (int i = 0; i < 10; i++) q[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
        } else p32_store(scratch, idx, P.X, P.Y, P.Z);
    }
}

// ---- constant-time fixed base for SMALL batches: a scalar's windows split between two threads of the block -----------------
// The 85 KB table allows one block per compute unit, so a batch of n <= 2^17 scalars in 1024-thread blocks uses n / 1024 of the
// 256 compute units (2^16 scalars: 64 of them, 0.78 ms), and smaller blocks leave every SIMD with a single wave that issues at
// half rate.  Here thread t < BS/2 of a block does windows 0 .. 25 of scalar t and thread t + BS/2 windows 26 .. 51 of the SAME
// scalar (after running the recoding's carry chain over the lower half: a handful of integer operations), so 2^16 scalars fill
// all 256 compute units with two waves per SIMD; the two partial points meet through LDS (the table's space, no longer
// needed) and one complete addition.  The part is uniform per wave, so the scan reads stay wave-uniform LDS addresses and the
// kernel is as constant-time as k_mul_base<5, CT> (tests/test_ct_isa.py covers both).
template <int BS, int OUT>
__global__ void __launch_bounds__(BS) k_mul_base_ct_split(const uint8_t *__restrict__ scalars, u64 n, const uint4 *__restrict__ gtab, u32 *__restrict__ scratch,
                                                          uint8_t *__restrict__ out_raw) {
    constexpr int W = C25519_CT_W, NWIN = (256 + W - 1) / W, HALF = 1 << (W - 1), ENT = HALF + 1, PER = BS / 2, WLO = NWIN / 2;
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < NWIN * ENT * 6; i += BS) lds[i] = gtab[i];
    __syncthreads();
    const int part = threadIdx.x >= PER ? 1 : 0, j = threadIdx.x - part * PER;        // wave-uniform: PER is a multiple of 64
    const int win0 = part ? WLO : 0, win1 = part ? NWIN : WLO;
#pragma unroll 1
    for (u64 base = (u64)blockIdx.x * PER; base < n; base += (u64)gridDim.x * PER) {   // block-uniform trip count (barriers inside)
        const u64 idx = base + j;
        const bool valid = idx < n;
        u32 s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (valid) load8(scalars, idx, s);
        u32 carry = 0;
#pragma unroll 1
        for (int win = 0; win < win0; win++) {               // the recoding's carry into this thread's first window (scalar.rs:1136-1147)
            const u32 d = (s[0] & (2u * HALF - 1u)) + carry;
#pragma unroll
            for (int i = 0; i < 7; i++) s[i] = (s[i] >> W) | (s[i + 1] << (32 - W));
            s[7] >>= W;
            carry = d >= (u32)HALF ? 1u : 0u;
        }
        ge_p3 P = ge_identity();
        const uint4 *wtab = lds + win0 * ENT * 6;
#pragma unroll 1
        for (int win = win0; win < win1; win++) {
            u32 d = (s[0] & (2u * HALF - 1u)) + carry;
#pragma unroll
            for (int i = 0; i < 7; i++) s[i] = (s[i] >> W) | (s[i + 1] << (32 - W));
            s[7] >>= W;
            const bool neg = (win != NWIN - 1) && (d >= (u32)HALF);
            const u32 mag = neg ? 2u * HALF - d : d;
            carry = neg ? 1u : 0u;
            u32 tw[24];
#pragma unroll
            for (int i = 0; i < 24; i++) tw[i] = 0;
            tw[0] = 1; tw[8] = 1;                           // entry 0 is the identity: not scanned
#pragma unroll 1
            for (int ent = 1; ent < ENT; ent++) {           // window.rs:54-76: every entry read (wave-uniform address), one kept by selects
                const uint4 *e = wtab + ent * 6;
                const bool hit = (u32)ent == mag;
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    uint4 v = e[i];
                    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));      // pinned: LLVM would sink the reads under `hit` (kernels.hip k_mul_base)
                    tw[4 * i] = hit ? v.x : tw[4 * i]; tw[4 * i + 1] = hit ? v.y : tw[4 * i + 1];
                    tw[4 * i + 2] = hit ? v.z : tw[4 * i + 2]; tw[4 * i + 3] = hit ? v.w : tw[4 * i + 3];
                }
            }
            aniels_words_cneg(tw, neg);
            P = ge_p1p1_to_p3(ge_madd(P, aniels_from_words(tw)));
            wtab += ENT * 6;
        }
        // the upper halves go through LDS (the table's space: every wave is past its last table read after the barrier)
        __syncthreads();
        uint4 *xch = lds + (size_t)j * 10;
        if (part) {
            u32 t[40];
            for (int i = 0; i < 10; i++) { t[i] = P.X.v[i]; t[10 + i] = P.Y.v[i]; t[20 + i] = P.Z.v[i]; t[30 + i] = P.T.v[i]; }
            for (int i = 0; i < 10; i++) xch[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
        }
        __syncthreads();
        if (!part) {
            u32 t[40];
            for (int i = 0; i < 10; i++) { const uint4 v = xch[i]; t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w; }
            ge_p3 Q;
            for (int i = 0; i < 10; i++) { Q.X.v[i] = t[i]; Q.Y.v[i] = t[10 + i]; Q.Z.v[i] = t[20 + i]; Q.T.v[i] = t[30 + i]; }
            P = ge_add(P, Q);
            if (valid) {
                if (OUT == 1) raw160_store(out_raw, idx, P);
                else if (OUT == 2) {
                    uint4 *q = reinterpret_cast<uint4 *>(scratch) + 10 * idx;
                    u32 o[40];
                    for (int i = 0; i < 10; i++) { o[i] = P.X.v[i]; o[10 + i] = P.Y.v[i]; o[20 + i] = P.Z.v[i]; o[30 + i] = P.T.v[i]; }
                    for (int i = 0; i < 10; i++) q[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
                } else p32_store(scratch, idx, P.X, P.Y, P.Z);
            }
        }
        // a further iteration needs the table again: restore it (only batches that outgrow the grid take this path)
        __syncthreads();
        if (base + (u64)gridDim.x * PER < n) {
            for (int i = threadIdx.x; i < NWIN * ENT * 6; i += BS) lds[i] = gtab[i];
            __syncthreads();
        }
    }
    // secrets do not stay in LDS: the exchanged partial points are wiped
    for (int i = threadIdx.x; i < PER * 10; i += BS) lds[i] = make_uint4(0, 0, 0, 0);
}

// ---- constant-time fixed base by CROSS-LANE FETCH (round 5): no scan, no selects ------------------------------------------
// The full-window scan above is at its issue bound with 434 v_cndmask per window beside the 757 multiplier instructions of the addition
// (a select costs a whole 4-cycle slot on gfx950: profiles/r04_ab_fixed_base_ct.txt).  What the reference's LookupTable::select
// (window.rs:54-76) has to guarantee is that no ADDRESS and no BRANCH depends on the digit.  Here the digit never reaches an address at all:
//   * group of 2^W lanes (W = 5: the two 32-lane halves of a wave; W = 6: the wave): lane j of its group reads, from LDS, the table entry of the
//     SIGNED digit value dv = j - 2^(W-1) of the current window -- |dv| selects the entry, and for dv < 0 the lane takes (y-x, y+x, p - 2dxy),
//     i.e. the negated point (curve_models.rs:501-512), all decided by its LANE INDEX: six ds_read_b128 at lane-id addresses per window;
//   * every lane then pulls the 24 words of ITS digit from lane (group base + d + 2^(W-1)) with ds_bpermute_b32 -- a register-to-register
//     crossbar transfer through the LDS unit that reads no memory.  The digit is the permute's lane selector, nothing else.
// No select on the digit, no conditional negation (the sign is part of the selector), 24 permutes + 6 reads instead of 96 reads + 434 selects.
// Whether the permute's duration depends on the selector pattern was measured before adopting it (c25519_microbench 50 .. 67, tools/probes.py,
// profiles/r05_instruction_rates.txt and, with the many-to-one patterns, profiles/r06_instruction_rates.txt: all-equal, identity, random within the group,
// pairs 32 lanes apart, k lanes on one source for k = 2 .. 32, random over the wave -- a MEASURED property of gfx950, not an architectural guarantee: see the
// constant-time paragraph of include/c25519_hip.h); with W = 5 a lane's sources stay inside its own 32-lane half by construction.
// EXEC must be all ones at the permutes (an inactive source lane would return zeros): the scalar loop is block-uniform, lanes past the end
// of the batch run on a zero scalar and are masked at the store.
// LDS table: [window][part 0..2 = y+x, y-x, 2dxy][16-byte half 0..1][entry 0..2^(W-1)]: consecutive lanes read consecutive 16-byte pieces.
template <int W>
struct ctp_lane {
    static constexpr int GS = 1 << W, HALF = GS / 2, ENT = HALF + 1, WSTRIDE = 6 * ENT;
    u32 a0, a1, b0, b1, c0, c1;      // uint4 offsets of this lane's six pieces inside a window's table
    u32 group;                       // first lane of this lane's group
    bool neg;                        // this lane serves a negative digit value
    __device__ __forceinline__ void init(u32 lane) {
        const int j = (int)(lane & (GS - 1)), dv = j - HALF;
        neg = dv < 0;
        const u32 mag = (u32)(neg ? -dv : dv);
        const u32 pa = neg ? 1u : 0u, pb = neg ? 0u : 1u;
        a0 = (pa * 2 + 0) * ENT + mag; a1 = (pa * 2 + 1) * ENT + mag;
        b0 = (pb * 2 + 0) * ENT + mag; b1 = (pb * 2 + 1) * ENT + mag;
        c0 = (2 * 2 + 0) * ENT + mag; c1 = (2 * 2 + 1) * ENT + mag;
        group = lane & ~(u32)(GS - 1);
    }
};
// stage the table of gtab ([window][entry] x 6 uint4, the layout of build_window_table) into the LDS layout above
template <int W, int BS>
__device__ __forceinline__ void ctp_stage(uint4 *lds, const uint4 *__restrict__ gtab) {
    constexpr int NWIN = (256 + W - 1) / W, ENT = (1 << (W - 1)) + 1;
    for (int i = threadIdx.x; i < NWIN * ENT * 6; i += BS) {
        const int win = i / (ENT * 6), r = i - win * (ENT * 6), e = r / 6, q = r - e * 6;
        lds[win * (ENT * 6) + q * ENT + e] = gtab[i];
    }
}
// the affine Niels point of signed digit dsel - 2^(W-1) of the window whose table starts at wtab (dsel in [0, 2^W))
template <int W>
__device__ __forceinline__ ge_aniels ctp_fetch(const uint4 *wtab, const ctp_lane<W> &L, u32 dsel) {
    uint4 v[6] = {wtab[L.a0], wtab[L.a1], wtab[L.b0], wtab[L.b1], wtab[L.c0], wtab[L.c1]};
    {   // p - 2dxy for the lanes that serve a negative digit value (a choice by lane index)
        const u32 c[8] = {v[4].x, v[4].y, v[4].z, v[4].w, v[5].x, v[5].y, v[5].z, v[5].w};
        u32 m[8];
        u64 borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u64 pi = (i == 0) ? 0xffffffedull : (i == 7 ? 0x7fffffffull : 0xffffffffull);
            const u64 d = pi - (u64)c[i] - borrow;
            borrow = (d >> 63) & 1;
            m[i] = L.neg ? (u32)d : c[i];
        }
        v[4] = make_uint4(m[0], m[1], m[2], m[3]); v[5] = make_uint4(m[4], m[5], m[6], m[7]);
    }
    const int sel = (int)((L.group + dsel) << 2);        // byte address of the source lane's register
    u32 tw[24];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        tw[4 * i + 0] = (u32)__builtin_amdgcn_ds_bpermute(sel, (int)v[i].x);
        tw[4 * i + 1] = (u32)__builtin_amdgcn_ds_bpermute(sel, (int)v[i].y);
        tw[4 * i + 2] = (u32)__builtin_amdgcn_ds_bpermute(sel, (int)v[i].z);
        tw[4 * i + 3] = (u32)__builtin_amdgcn_ds_bpermute(sel, (int)v[i].w);
    }
    return aniels_from_words(tw);
}
// ---- the same with the table kept as LIMBS in LDS (C25519_CT_LIMBS, the default): an entry is 3 x 10 tight limbs, each field element padded to three
// 16-byte pieces (144 bytes per entry, 52 x 17 x 144 = 127 KB) -- nine ds_read_b128 and THIRTY permutes per window, but no unpacking of three 255-bit
// values into limbs per window (~70 VALU instructions) and the negation is limb-wise (2p - x: ten subtractions, no borrow chain).
// LDS layout: [window][part 0..2][piece 0..2][entry].
template <int W>
struct ctp_lane_l {
    static constexpr int GS = 1 << W, HALF = GS / 2, ENT = HALF + 1, WSTRIDE = 9 * ENT;
    u32 a0, b0, c0;                  // uint4 offsets of the first piece of this lane's three field elements (pieces are ENT apart)
    u32 group;
    bool neg;
    __device__ __forceinline__ void init(u32 lane) {
        const int j = (int)(lane & (GS - 1)), dv = j - HALF;
        neg = dv < 0;
        const u32 mag = (u32)(neg ? -dv : dv);
        a0 = (neg ? 3u : 0u) * ENT + mag; b0 = (neg ? 0u : 3u) * ENT + mag; c0 = 6u * ENT + mag;
        group = lane & ~(u32)(GS - 1);
    }
};
template <int W, int BS>
__device__ __forceinline__ void ctp_stage_l(uint4 *lds, const uint4 *__restrict__ gtab) {
    constexpr int NWIN = (256 + W - 1) / W, ENT = (1 << (W - 1)) + 1;
    for (int i = threadIdx.x; i < NWIN * ENT * 3; i += BS) {             // one field element per trip: words -> limbs once per block
        const int win = i / (ENT * 3), r = i - win * (ENT * 3), e = r / 3, part = r - e * 3;
        const uint4 lo = gtab[(win * ENT + e) * 6 + 2 * part], hi = gtab[(win * ENT + e) * 6 + 2 * part + 1];
        const feT f = fe_from_q(lo, hi);
        uint4 *dst = lds + win * (ENT * 9) + (part * 3) * ENT + e;
        dst[0] = make_uint4(f.v[0], f.v[1], f.v[2], f.v[3]); dst[ENT] = make_uint4(f.v[4], f.v[5], f.v[6], f.v[7]); dst[2 * ENT] = make_uint4(f.v[8], f.v[9], 0u, 0u);
    }
}
template <int W>
__device__ __forceinline__ ge_aniels ctp_fetch_l(const uint4 *wtab, const ctp_lane_l<W> &L, u32 dsel) {
    constexpr int ENT = (1 << (W - 1)) + 1;
    const uint4 a0 = wtab[L.a0], a1 = wtab[L.a0 + ENT], a2 = wtab[L.a0 + 2 * ENT];
    const uint4 b0 = wtab[L.b0], b1 = wtab[L.b0 + ENT], b2 = wtab[L.b0 + 2 * ENT];
    const uint4 c0 = wtab[L.c0], c1 = wtab[L.c0 + ENT], c2 = wtab[L.c0 + 2 * ENT];
    u32 own[30] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y,
                   c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y};
    // 2p - 2dxy, limb-wise, for the lanes that serve a negative digit value (a choice by lane index); tight in, below 2p's limbs out: a legal `g` operand
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const u32 m = (i == 0 ? 0x7ffffdau : (i & 1) ? 0x3fffffeu : 0x7fffffeu) - own[20 + i];
        own[20 + i] = L.neg ? m : own[20 + i];
    }
    const int sel = (int)((L.group + dsel) << 2);
    ge_aniels A;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        A.ypx.v[i] = (u32)__builtin_amdgcn_ds_bpermute(sel, (int)own[i]);
        A.ymx.v[i] = (u32)__builtin_amdgcn_ds_bpermute(sel, (int)own[10 + i]);
        A.xy2d.v[i] = (u32)__builtin_amdgcn_ds_bpermute(sel, (int)own[20 + i]);
    }
    return A;
}
#ifndef C25519_CT_LIMBS
#define C25519_CT_LIMBS 1
#endif
// SPLIT = false: one lane per scalar, all windows (large batches).  SPLIT = true: two lanes per scalar, windows [0, NWIN/2) and [NWIN/2, NWIN)
// (small batches: every compute unit gets a block; the partial sums meet through the table's LDS space -- cf. k_mul_base_ct_split)
template <int W, int BS, int OUT, bool SPLIT, bool LIMBS>
__global__ void __launch_bounds__(BS) k_mul_base_ctp(const uint8_t *__restrict__ scalars, u64 n, const uint4 *__restrict__ gtab, u32 *__restrict__ scratch,
                                                     uint8_t *__restrict__ out_raw) {
    constexpr int NWIN = (256 + W - 1) / W, HALF = 1 << (W - 1), ENT = HALF + 1, PER = SPLIT ? BS / 2 : BS, WLO = NWIN / 2, EQ = LIMBS ? 9 : 6;
    extern __shared__ uint4 lds[];
    if (LIMBS) ctp_stage_l<W, BS>(lds, gtab); else ctp_stage<W, BS>(lds, gtab);
    __syncthreads();
    ctp_lane<W> L;
    ctp_lane_l<W> LL;
    L.init(threadIdx.x & 63u); LL.init(threadIdx.x & 63u);
    // (wave-uniform: PER is a multiple of 64 -- said to the compiler with readfirstlane, so that the window loop below is a SCALAR loop:
    //  a loop on a per-lane trip count would run under an exec mask, and the permutes need all 64 lanes)
    const int part = SPLIT ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >= (unsigned)PER)) : 0, j = (int)threadIdx.x - part * PER;
    const int win0 = part ? WLO : 0, win1 = (SPLIT && !part) ? WLO : NWIN;
#pragma unroll 1
    for (u64 base = (u64)blockIdx.x * PER; base < n; base += (u64)gridDim.x * PER) {    // block-uniform trip count (permutes and barriers inside)
        const u64 idx = base + (u64)j;
        const bool valid = idx < n;
        u32 s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (valid) load8(scalars, idx, s);
        u32 carry = 0;
#pragma unroll 1
        for (int win = 0; win < win0; win++) {               // the recoding's carry into this thread's first window (scalar.rs:1136-1147)
            const u32 d = (s[0] & (2u * HALF - 1u)) + carry;
#pragma unroll
            for (int i = 0; i < 7; i++) s[i] = (s[i] >> W) | (s[i + 1] << (32 - W));
            s[7] >>= W;
            carry = d >= (u32)HALF ? 1u : 0u;
        }
        ge_p3 P = ge_identity();
        const uint4 *wtab = lds + win0 * (ENT * EQ);
#pragma unroll 1
        for (int win = win0; win < win1; win++) {
            const u32 d = (s[0] & (2u * HALF - 1u)) + carry;                 // 0 .. 2^W
#pragma unroll
            for (int i = 0; i < 7; i++) s[i] = (s[i] >> W) | (s[i + 1] << (32 - W));
            s[7] >>= W;
            // recentre to [-HALF, HALF) except in the top window (scalar.rs:1136-1147): digit value d - 2^W when d >= HALF; the selector is
            // (digit value + HALF) in [0, 2^W) -- i.e. d + HALF, or d - HALF after a carry out
            const bool hi = (win != NWIN - 1) && (d >= (u32)HALF);
            carry = hi ? 1u : 0u;
            const u32 dsel = hi ? d - (u32)HALF : d + (u32)HALF;
            if (LIMBS) P = ge_p1p1_to_p3(ge_madd(P, ctp_fetch_l<W>(wtab, LL, dsel)));
            else P = ge_p1p1_to_p3(ge_madd(P, ctp_fetch<W>(wtab, L, dsel)));
            wtab += ENT * EQ;
        }
        if (SPLIT) {
            // the upper halves go through LDS (the table's space: every wave is past its last table read after the barrier)
            __syncthreads();
            uint4 *xch = lds + (size_t)j * 10;
            if (part) {
                u32 t[40];
                for (int i = 0; i < 10; i++) { t[i] = P.X.v[i]; t[10 + i] = P.Y.v[i]; t[20 + i] = P.Z.v[i]; t[30 + i] = P.T.v[i]; }
                for (int i = 0; i < 10; i++) xch[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
            }
            __syncthreads();
            if (!part) {
                u32 t[40];
                for (int i = 0; i < 10; i++) { const uint4 v = xch[i]; t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w; }
                ge_p3 Q;
                for (int i = 0; i < 10; i++) { Q.X.v[i] = t[i]; Q.Y.v[i] = t[10 + i]; Q.Z.v[i] = t[20 + i]; Q.T.v[i] = t[30 + i]; }
                P = ge_add(P, Q);
            }
        }
        if (valid && !part) {
            if (OUT == 1) raw160_store(out_raw, idx, P);
            else if (OUT == 2) {
                uint4 *q = reinterpret_cast<uint4 *>(scratch) + 10 * idx;
                u32 o[40];
                for (int i = 0; i < 10; i++) { o[i] = P.X.v[i]; o[10 + i] = P.Y.v[i]; o[20 + i] = P.Z.v[i]; o[30 + i] = P.T.v[i]; }
                for (int i = 0; i < 10; i++) q[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
            } else p32_store(scratch, idx, P.X, P.Y, P.Z);
        }
        if (SPLIT) {
            // a further iteration needs the table again: restore it (only batches that outgrow the grid take this path)
            __syncthreads();
            if (base + (u64)gridDim.x * PER < n) { if (LIMBS) ctp_stage_l<W, BS>(lds, gtab); else ctp_stage<W, BS>(lds, gtab); __syncthreads(); }
        }
    }
    // secrets do not stay in LDS: the exchanged partial points are wiped
    if (SPLIT) for (int i = threadIdx.x; i < PER * 10; i += BS) lds[i] = make_uint4(0, 0, 0, 0);
}

// ================================================================================================
// K2c fixed-base batch, signed multi-table comb (context flags = 9; bootstraps the wide tables): 31 mixed additions + 4 doublings per
//     scalar instead of 43 additions.
//     With s' = s | 1 and c = (s' + 2^270 - 1)/2 = (s >> 1) + 2^269, s' = sum_{i<270} (2 c_i - 1) 2^i
//     (every digit is +-1).  Bits are arranged in T = 9 teeth x D = 30 columns (bit i = tooth*30 + col);
//     column j contributes 2^j * sum_tau (+-1) 2^(30 tau) B, a 9-bit sign pattern looked up in a table
//     of 2^8 entries (top tooth positive; the opposite pattern is the negated entry).  V = 6 tables
//     hold the patterns pre-multiplied by 2^(5m), so columns j = 5m + r share one Horner step:
//         acc = 2*acc + sum_m T_m[pattern(5m + r)],  r = 4..0.
//     Finally subtract B when s was even.  Table: [6][256] entries x 96 B = 147 456 B of LDS, plus the
//     entry for B.  Same affine-Niels entry format and mixed addition as k_mul_base.
// ================================================================================================
constexpr int COMB_T = 9, COMB_V = 6, COMB_E = 5, COMB_ENT = 256;
constexpr int COMB_TABLE_Q = (COMB_V * COMB_ENT + 1) * 6;   // uint4 count (last entry = B)

template <int BS, int OUT>
__global__ void __launch_bounds__(BS) k_mul_base_comb(const uint8_t *__restrict__ scalars, u64 n, const uint4 *__restrict__ gtab,
                                                      u32 *__restrict__ scratch, uint8_t *__restrict__ out_raw) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < COMB_TABLE_Q; i += BS) lds[i] = gtab[i];
    __syncthreads();
    for (u64 idx = (u64)blockIdx.x * BS + threadIdx.x; idx < n; idx += (u64)gridDim.x * BS) {
        u32 s[8];
        load8(scalars, idx, s);
        const bool even = (s[0] & 1u) == 0;
        // c = (s >> 1) + 2^269 as nine 30-bit tooth registers R[tau] = bits [30 tau, 30 tau + 30) of c
        u32 c[9];
#pragma unroll
        for (int i = 0; i < 7; i++) c[i] = (s[i] >> 1) | (s[i + 1] << 31);
        c[7] = s[7] >> 1; c[8] = 1u << 13;                       // bit 269 = word 8, bit 13
        u32 R[COMB_T];
#pragma unroll
        for (int t = 0; t < COMB_T; t++) {
            const int bit = 30 * t, wi = bit >> 5, sh = bit & 31;
            u64 two = (u64)c[wi] | ((u64)(wi + 1 < 9 ? c[wi + 1] : 0u) << 32);
            R[t] = (u32)(two >> sh) & 0x3fffffffu;
        }
        ge_p3 P = ge_identity();
#pragma unroll 1
        for (int r = COMB_E - 1; r >= 0; r--) {
            if (r != COMB_E - 1) P = ge_dbl_p3(P);
#pragma unroll 1
            for (int m = 0; m < COMB_V; m++) {
                const int j = COMB_E * m + r;
                u32 pat = 0;
#pragma unroll
                for (int t = 0; t < COMB_T; t++) pat |= ((R[t] >> j) & 1u) << t;
                const bool neg = (pat >> (COMB_T - 1)) == 0;     // top tooth digit -1: use the opposite pattern, negated
                const u32 e_idx = (neg ? ~pat : pat) & (COMB_ENT - 1);
                const uint4 *e = lds + ((u32)m * COMB_ENT + e_idx) * 6;
                uint4 q0 = e[0], q1 = e[1], q2 = e[2], q3 = e[3], q4 = e[4], q5 = e[5];
                u32 tw[24] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w,
                              q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w};
                aniels_words_cneg(tw, neg);
                P = ge_p1p1_to_p3(ge_madd(P, aniels_from_words(tw)));
            }
        }
        {   // s even: s = s' - 1  ->  subtract B (add the identity otherwise)
            const uint4 *e = lds + (COMB_V * COMB_ENT) * 6;
            uint4 q0 = e[0], q1 = e[1], q2 = e[2], q3 = e[3], q4 = e[4], q5 = e[5];
            u32 tw[24] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w,
                          q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w};
            aniels_words_cneg(tw, true);
            ge_aniels A = aniels_from_words(tw);
            A.ypx = fe_select(fe_one(), A.ypx, even); A.ymx = fe_select(fe_one(), A.ymx, even); A.xy2d = fe_select(fe_zero(), A.xy2d, even);
            P = ge_p1p1_to_p3(ge_madd(P, A));
        }
        if (OUT == 1) raw160_store(out_raw, idx, P);
        else if (OUT == 2) {
            uint4 *q = reinterpret_cast<uint4 *>(scratch) + 10 * idx;
            u32 t[40];
            for (int i = 0; i < 10; i++) { t[i] = P.X.v[i]; t[10 + i] = P.Y.v[i]; t[20 + i] = P.Z.v[i]; t[30 + i] = P.T.v[i]; }
            for (int i = 0; i < 10; i++) q[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
        } else p32_store(scratch, idx, P.X, P.Y, P.Z);
    }
}

// ================================================================================================
// K1w fixed base with WIDE signed windows: s = sum_j d_j 2^(C j), d_j in (-2^(C-1), 2^(C-1)], and one table of
//     e * 2^(C j) * B per digit position (the reference's EdwardsBasepointTable structure, edwards.rs:1131-1141,
//     with radix 2^C instead of 16): ceil(256 / C) mixed additions, no doublings.  The table does not fit LDS --
//     it lives in HBM and is served by L2 (C <= 12: <= 5.8 MB) or the 256 MB MALL (C = 16: 71 MB), so each
//     addition costs one 128-byte gather -- exactly one cache line: an entry is (y+x, y-x, 2dxy) as 3 x 10 LIMBS
//     + 8 bytes of padding, ready for ge_madd_signed_p3 without unpacking -- prefetched one window ahead.  The top
//     window is unsigned (it absorbs the last carry and bit 255) and has its own length.
//     Layout: windows 0 .. nw-2: 2^(C-1)+1 entries each (entry 0 = identity), then 2^rem + 1 entries.
// ================================================================================================
template <int OUT>
__global__ void __launch_bounds__(256) k_mul_base_wide(const uint8_t *__restrict__ scalars, u64 n, const u32 *__restrict__ tab, int C, int nw,
                                                       u32 *__restrict__ scratch, uint8_t *__restrict__ out_raw) {
    const u64 idx = (u64)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    u32 s[8];
    load8(scalars, idx, s);
    const u32 HALF = 1u << (C - 1), MASK = (HALF << 1) - 1u, ENT = HALF + 1;
    // The table entry of window j+1 is fetched while the addition of window j runs.  Holding it in registers would
    // cost 30 VGPRs for the whole iteration; instead each wave DMAs it straight into its own 8 x 1 KiB LDS slots
    // (global_load_lds_dwordx4: per-lane global address, LDS destination = wave base + 16 * lane) and reads it back
    // at the top of the next iteration.  Every lane only ever touches its own slot, so no barrier is involved:
    // the wave's own vmcnt(0) orders DMA -> ds_read, and lgkmcnt(0) orders ds_read -> the next DMA into the slot.
    // (The wave-cooperative form of k_accumulate<4> -- eight lanes per entry, 8 cache lines per instruction instead of 64 --
    //  was tried here: 0.563 against 0.533 ms.  Nothing shares the texture path with this kernel, and the eight
    //  ds_bpermute + address computations per window are not free.)
    __shared__ uint4 stage[8 * 256];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;
    // (r6, last) the wave's slot from a SCALAR wave index, the entry by a 32-bit byte offset from the scalar table base, the eight pieces by the instruction's own offset field
    // (which also moves the LDS destination: the slot base compensates) -- one v_lshl_add_u32 per window where the 64-bit addresses took eleven vector instructions and
    // eight v_readfirstlane (accum.hip has the same change; the table is at most 17 x 32 769 entries of 128 bytes: 71 MB)
    uint4 *wave_slot = stage + (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 64u;
    uint4 *my_slot = stage + threadIdx.x;
    // the scalar is a shift register: the current window is always its low C bits
    u32 d = s[0] & MASK;
    bool neg = d > HALF;
    u32 carry = neg ? 1u : 0u;
    u32 mag = neg ? (MASK + 1u - d) : d;
    u32 e = mag << 7;
#define C25519_STAGE_PIECE(boff, i)                                                                                      \
    __builtin_amdgcn_global_load_lds((gbl_void *)(reinterpret_cast<const char *>(tab) + (boff)), (lds_void *)(reinterpret_cast<char *>(wave_slot + (i) * 256) - 16 * (i)), 16, 16 * (i), 0)
#define C25519_STAGE_ENTRY(boff)                                                                                         \
    C25519_STAGE_PIECE(boff, 0); C25519_STAGE_PIECE(boff, 1); C25519_STAGE_PIECE(boff, 2); C25519_STAGE_PIECE(boff, 3);  \
    C25519_STAGE_PIECE(boff, 4); C25519_STAGE_PIECE(boff, 5); C25519_STAGE_PIECE(boff, 6); C25519_STAGE_PIECE(boff, 7)
    C25519_STAGE_ENTRY(e);
    ge_p3 P = ge_identity();
#pragma unroll 1
    for (int j = 0; j < nw; j++) {
        const bool cur_neg = neg;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // entry j has landed in the slot
        uint4 q[8];
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = my_slot[i * 256];
        u32 tw[32] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w,
                      q[4].x, q[4].y, q[4].z, q[4].w, q[5].x, q[5].y, q[5].z, q[5].w, q[6].x, q[6].y, q[6].z, q[6].w, q[7].x, q[7].y, q[7].z, q[7].w};
#pragma unroll
        for (int i = 0; i < 30; i++) asm volatile("" : "+v"(tw[i]));   // the reads are complete (lgkmcnt) before ...
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j + 1 < nw) {                                           // ... the slot is refilled with entry j+1
#pragma unroll
            for (int i = 0; i < 7; i++) s[i] = __funnelshift_r(s[i], s[i + 1], C);
            s[7] >>= C;
            d = (s[0] & MASK) + carry;
            const bool top = (j + 2 == nw);
            neg = !top && d > HALF;
            carry = neg ? 1u : 0u;
            mag = neg ? (MASK + 1u - d) : d;
            e = ((u32)(j + 1) * ENT + mag) << 7;
            C25519_STAGE_ENTRY(e);
        }
        ge_aniels A;                                                // the table holds limbs: no unpacking
#pragma unroll
        for (int i = 0; i < 10; i++) { A.ypx.v[i] = tw[i]; A.ymx.v[i] = tw[10 + i]; A.xy2d.v[i] = tw[20 + i]; }
        // (the digit's sign applied lazily to the running point -- ge26.h ge_madd_lazy_p3, as k_accumulate does -- measured level here: 0.578 against 0.580 of the
        //  multiplier peak, gpurun call 68; every window's addition waits for the previous one, and the negation sits on that chain)
        if (j == 0) P = ge_from_aniels_signed(A, cur_neg);         // identity + Q: 1 M instead of 7 M (ge26.h)
        else P = ge_madd_signed_p3(P, A, cur_neg);
        ge_pin(P);
    }
#undef C25519_STAGE_ENTRY
#undef C25519_STAGE_PIECE
    if (OUT == 1) raw160_store(out_raw, idx, P);
    else if (OUT == 2) {
        uint4 *q = reinterpret_cast<uint4 *>(scratch) + 10 * idx;
        u32 t[40];
        for (int i = 0; i < 10; i++) { t[i] = P.X.v[i]; t[10 + i] = P.Y.v[i]; t[20 + i] = P.Z.v[i]; t[30 + i] = P.T.v[i]; }
        for (int i = 0; i < 10; i++) q[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
    } else p32_store(scratch, idx, P.X, P.Y, P.Z);
}

// ================================================================================================
// K4  X25519 batch: constant-time cswap ladder, montgomery.rs:183-211 + :430-468, then U/W
//     (as_affine :409; invert(0) = 0 so low-order inputs give the all-zero output).
//     The scalar is kept as a 256-bit shift register so no register is indexed dynamically.
// ================================================================================================
#ifdef C25519_X25519_WAVES       // A/B arm (tools/build_variant.sh): a register budget for this many waves per SIMD (default: the compiler's 137 VGPRs = 3 waves)
#define C25519_X25519_ATTR __attribute__((amdgpu_waves_per_eu(C25519_X25519_WAVES, C25519_X25519_WAVES)))
#else
#define C25519_X25519_ATTR
#endif
__global__ void __launch_bounds__(256) C25519_X25519_ATTR k_x25519(const uint8_t *__restrict__ ks, const uint8_t *__restrict__ us, u64 n,
                                                u32 *__restrict__ scratch) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    u32 s[8], uw[8];
    load8(ks, idx, s);
    load8(us, idx, uw);
    s[0] &= 0xfffffff8u; s[7] &= 0x7fffffffu; s[7] |= 0x40000000u;   // clamp_integer, scalar.rs:1407
    feT au = fe_from_words(uw);
    mont_pp x0, x1;
    x0.U = fe_one(); x0.W = fe_zero(); x1.U = au; x1.W = fe_one();
#ifdef C25519_X25519_SCALAR_LDS
    // A/B arm: the clamped scalar waits in LDS (word w of thread t at [w][t]: conflict-free) and ONE word at a time is a shift register -- seven
    // registers and seven of the eight funnel shifts of a ladder step less.  The LDS address is the loop counter: public.
    __shared__ u32 sk[8 * 256];
#pragma unroll
    for (int w = 0; w < 8; w++) sk[w * 256 + threadIdx.x] = s[w];
    u32 prev = 0, word = 0;
#pragma unroll 1
    for (int i = 0; i < 255; i++) {
        const int bit = 254 - i;                               // bits 254 .. 0 of the clamped scalar (bit 255 is clear)
        if ((bit & 31) == 31 || i == 0) word = sk[(bit >> 5) * 256 + threadIdx.x] << (31 - (bit & 31));
        const u32 cur = word >> 31;
        word <<= 1;
        const u32 sw = prev ^ cur;
        fe_cswap(x0.U, x1.U, sw); fe_cswap(x0.W, x1.W, sw);
        mont_diff_add_and_double(x0, x1, au);
        prev = cur;
    }
#pragma unroll
    for (int w = 0; w < 8; w++) sk[w * 256 + threadIdx.x] = 0;        // the secret does not stay in LDS
#else
    // bit 254 -> position 255
#pragma unroll
    for (int i = 7; i > 0; i--) s[i] = (s[i] << 1) | (s[i - 1] >> 31);
    s[0] <<= 1;
    u32 prev = 0;
#pragma unroll 1
    for (int i = 0; i < 255; i++) {
        u32 cur = s[7] >> 31;
#pragma unroll
        for (int k = 7; k > 0; k--) s[k] = (s[k] << 1) | (s[k - 1] >> 31);
        s[0] <<= 1;
        u32 sw = prev ^ cur;
        fe_cswap(x0.U, x1.U, sw); fe_cswap(x0.W, x1.W, sw);
        mont_diff_add_and_double(x0, x1, au);
        prev = cur;
    }
#endif
    fe_cswap(x0.U, x1.U, prev); fe_cswap(x0.W, x1.W, prev);
    p32_store(scratch, idx, x0.U, x0.U, x0.W);     // (U : W); the division is batched in k_ratio_p32
}

// ================================================================================================
// K5  decompression batch (edwards.rs:211-258): one pow_p58 per lane; validity byte per point.
// ================================================================================================
__global__ void __launch_bounds__(256) k_decompress_edwards(const uint8_t *__restrict__ in, u64 n, uint8_t *__restrict__ out_raw,
                                                            uint8_t *__restrict__ ok, u32 *__restrict__ any_bad) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    u32 w[8];
    load8(in, idx, w);
    ge_p3 P;
    bool good = ge_decompress(P, w);
    raw160_store(out_raw, idx, P);
    ok[idx] = good ? 1 : 0;
    if (!good) atomicOr(any_bad, 1u);
}


// Ristretto variants (ristretto.rs:266-345, :500-533): one inverse square root per lane each.
__global__ void __launch_bounds__(256) k_decompress_ristretto(const uint8_t *__restrict__ in, u64 n, uint8_t *__restrict__ out_raw,
                                                              uint8_t *__restrict__ ok, u32 *__restrict__ any_bad) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    u32 w[8];
    load8(in, idx, w);
    ge_p3 P;
    bool good = ris_decompress(P, w);
    raw160_store(out_raw, idx, P);
    ok[idx] = good ? 1 : 0;
    if (!good) atomicOr(any_bad, 1u);
}
// ================================================================================================
// MSM / verify_batch input preparation for compressed points (msm.hip prep_points)
// ================================================================================================
// compressed (Edwards y / Ristretto) -> packed affine Niels at pts[dst0 + i]; bad encodings counted
template <int FMT>
__global__ void __launch_bounds__(256) C25519_PREPC_ATTR k_prep_compressed(const uint8_t *__restrict__ in, u64 stride_items, u64 n, u32 *__restrict__ pts,
                                                         u64 dst0, u32 *__restrict__ bad_count) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 w[8];
    load8(in, i * stride_items, w);     // stride 1 for point arrays, 2 to pick R out of 64-byte signatures
    ge_p3 P;
    bool ok = (FMT == 0) ? ge_decompress(P, w) : ris_decompress(P, w);
    pts_store(pts, dst0 + i, P.X, P.Y);
    if (!ok) atomicAdd(bad_count, 1u);
}

// verify_batch, small batches: the keys A_i (n x 32 bytes -> records n + 1 ..) and the R_i (first half of every 64-byte signature -> records 1 ..)
// in ONE launch.  A decompression is a chain of ~280 dependent field operations, ~70 us for a lone wave however few points there are: two
// launches one after the other are two of those chains.  bad_count[0] counts the keys, bad_count[1] the R_i that do not decode.
__global__ void __launch_bounds__(256) k_prep_compressed_keys_and_r(const uint8_t *__restrict__ pks, const uint8_t *__restrict__ sigs, u64 n, u32 *__restrict__ pts,
                                                                    u32 *__restrict__ bad_count) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n) return;
    const bool is_r = i >= n;
    const u64 j = is_r ? i - n : i;
    u32 w[8];
    load8(is_r ? sigs : pks, is_r ? 2 * j : j, w);
    ge_p3 P;
    const bool ok = ge_decompress(P, w);
    pts_store(pts, (is_r ? 1 : n + 1) + j, P.X, P.Y);
    if (!ok) atomicAdd(bad_count + (is_r ? 1 : 0), 1u);
}

// (k_prep_small_verify -- the same for at most 128 signatures in ONE block -- lives in finish.hip since round 6: a lone wave's 252-squaring chain wants the ten-column
//  products of that translation unit, not the chained ones of this)

// ================================================================================================
// launchers
// ================================================================================================
static inline unsigned div_up(u64 a, u64 b) { return (unsigned)((a + b - 1) / b); }
hipError_t launch_prep_compressed_keys_and_r(const uint8_t *pks, const uint8_t *sigs, uint64_t n, uint32_t *pts, uint32_t *bad_count, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prep_compressed_keys_and_r, dim3(div_up(2 * n, 256)), dim3(256), 0, st, pks, sigs, n, pts, bad_count);
    return hipGetLastError();
}

template <int W, int BS, bool CT = false>
static hipError_t launch_mul_base_w(const uint8_t *scalars, u64 n, const uint32_t *tab, uint32_t *scratch, uint8_t *out_raw,
                                    int num_cus, hipStream_t st, bool p40 = false) {
    constexpr int NWIN = (256 + W - 1) / W, ENT = (1 << (W - 1)) + 1;
    size_t lds_bytes = (size_t)NWIN * ENT * 96;
    unsigned grid = div_up(n, BS);
    unsigned maxgrid = (unsigned)num_cus * (lds_bytes > 80 * 1024 ? 1u : 2u);
    if (grid > maxgrid) grid = maxgrid;
    if (p40) {
        auto kfn = k_mul_base<W, BS, 2, CT>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(BS), lds_bytes, st, scalars, n, reinterpret_cast<const uint4 *>(tab), scratch, out_raw);
    } else if (out_raw) {
        auto kfn = k_mul_base<W, BS, 1, CT>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(BS), lds_bytes, st, scalars, n, reinterpret_cast<const uint4 *>(tab), scratch, out_raw);
    } else {
        auto kfn = k_mul_base<W, BS, 0, CT>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(BS), lds_bytes, st, scalars, n, reinterpret_cast<const uint4 *>(tab), scratch, out_raw);
    }
    return hipGetLastError();
}

template <int BS, int OUT>
static hipError_t launch_comb_bs(const uint8_t *scalars, u64 n, const uint32_t *tab, uint32_t *scratch, uint8_t *out_raw, int num_cus, hipStream_t st);
template <int OUT>
static hipError_t launch_comb(const uint8_t *scalars, u64 n, const uint32_t *tab, uint32_t *scratch, uint8_t *out_raw, int num_cus, hipStream_t st) {
    return launch_comb_bs<1024, OUT>(scalars, n, tab, scratch, out_raw, num_cus, st);     // (768 / 512-thread blocks measured slower in round 1)
}
template <int BS, int OUT>
static hipError_t launch_comb_bs(const uint8_t *scalars, u64 n, const uint32_t *tab, uint32_t *scratch, uint8_t *out_raw, int num_cus, hipStream_t st) {
    size_t lds_bytes = (size_t)COMB_TABLE_Q * 16;
    unsigned grid = div_up(n, BS);
    if (grid > (unsigned)num_cus) grid = (unsigned)num_cus;
    auto kfn = k_mul_base_comb<BS, OUT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(BS), lds_bytes, st, scalars, n, reinterpret_cast<const uint4 *>(tab), scratch, out_raw);
    return hipGetLastError();
}

hipError_t launch_mul_base(int w, const uint8_t *scalars, u64 n, const uint32_t *tab, uint32_t *scratch, uint8_t *out_raw,
                           int num_cus, hipStream_t st) {
    if (n == 0) return hipSuccess;
    switch (w) {
    case 9: return out_raw ? launch_comb<1>(scalars, n, tab, scratch, out_raw, num_cus, st) : launch_comb<0>(scalars, n, tab, scratch, out_raw, num_cus, st);
    case 4: return launch_mul_base_w<4, 512>(scalars, n, tab, scratch, out_raw, num_cus, st);
    case 5: return launch_mul_base_w<5, 512>(scalars, n, tab, scratch, out_raw, num_cus, st);
    case 6: return launch_mul_base_w<6, 1024>(scalars, n, tab, scratch, out_raw, num_cus, st);
    default: break;
    }
    if (w >= 10 && w <= 20) {
        const int nw = (256 + w - 1) / w;
        if (out_raw) hipLaunchKernelGGL(k_mul_base_wide<1>, dim3(div_up(n, 256)), dim3(256), 0, st, scalars, n, tab, w, nw, scratch, out_raw);
        else hipLaunchKernelGGL(k_mul_base_wide<0>, dim3(div_up(n, 256)), dim3(256), 0, st, scalars, n, tab, w, nw, scratch, out_raw);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

// constant-time fixed base (secret scalars): radix-2^5 LDS tables with the full-window scan.  out_raw / scratch as launch_mul_base;
// p40: write P40 records to scratch instead
template <int BS>
static hipError_t launch_ct_split(const uint8_t *scalars, u64 n, const uint32_t *tab_ct, uint32_t *scratch, uint8_t *out_raw, int num_cus, hipStream_t st, bool p40) {
    constexpr int NWIN = (256 + C25519_CT_W - 1) / C25519_CT_W, ENT = (1 << (C25519_CT_W - 1)) + 1;
    const size_t lds_bytes = (size_t)NWIN * ENT * 96;      // (the exchange of BS/2 x 160 bytes reuses it: 80 KB <= 85 KB at BS = 1024)
    unsigned grid = div_up(n, BS / 2);
    if (grid > (unsigned)num_cus) grid = (unsigned)num_cus;
#define C25519_CT_SPLIT_LAUNCH(OUTV)                                                                                                   \
    {                                                                                                                                  \
        auto kfn = k_mul_base_ct_split<BS, OUTV>;                                                                                      \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        if (e != hipSuccess) return e;                                                                                                 \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(BS), lds_bytes, st, scalars, n, reinterpret_cast<const uint4 *>(tab_ct), scratch, out_raw); \
    }
    if (p40) C25519_CT_SPLIT_LAUNCH(2) else if (out_raw) C25519_CT_SPLIT_LAUNCH(1) else C25519_CT_SPLIT_LAUNCH(0)
#undef C25519_CT_SPLIT_LAUNCH
    return hipGetLastError();
}
// the cross-lane-fetch form (k_mul_base_ctp).  The limb tables serve the P32 output (the batched compressor: mul_base, keygen, sign -- 122 VGPRs, no
// scratch); the raw-point and P40 outputs keep the packed tables (with 30 more live words their 1024-thread kernels would spill 9 - 14 registers
// into the window loop)
template <int W, int BS, bool SPLIT>
static hipError_t launch_ctp(const uint8_t *scalars, u64 n, const uint32_t *tab_ct, uint32_t *scratch, uint8_t *out_raw, int num_cus, hipStream_t st, bool p40) {
    constexpr int NWIN = (256 + W - 1) / W, ENT = (1 << (W - 1)) + 1;
    const bool limbs = C25519_CT_LIMBS && !p40 && !out_raw;
    const size_t lds_bytes = (size_t)NWIN * ENT * (limbs ? 144 : 96);      // (SPLIT: the exchange of BS/2 x 160 bytes reuses it: 80 KB at BS = 1024)
    unsigned grid = div_up(n, SPLIT ? BS / 2 : BS);
    if (grid > (unsigned)num_cus) grid = (unsigned)num_cus;
#define C25519_CTP_LAUNCH(OUTV, LIMBSV)                                                                                                \
    {                                                                                                                                  \
        auto kfn = k_mul_base_ctp<W, BS, OUTV, SPLIT, LIMBSV>;                                                                         \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        if (e != hipSuccess) return e;                                                                                                 \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(BS), lds_bytes, st, scalars, n, reinterpret_cast<const uint4 *>(tab_ct), scratch, out_raw); \
    }
    if (p40) C25519_CTP_LAUNCH(2, false) else if (out_raw) C25519_CTP_LAUNCH(1, false) else if (limbs) C25519_CTP_LAUNCH(0, true) else C25519_CTP_LAUNCH(0, false)
#undef C25519_CTP_LAUNCH
    return hipGetLastError();
}
// A/B knob (tuning build only): 0 = the full-window scan of rounds 2-4 (k_mul_base<5, CT>, k_mul_base_ct_split), 1 = the cross-lane fetch
static int ct_fetch() { static const int v = C25519_KNOB("CT_FETCH", 1); return v; }
// the kernel launch_mul_base_ct picks for n scalars (what the timing records attribute phase 0 to)
const char *mul_base_ct_kernel_name(u64 n, int num_cus) {
    const u64 per_cu = (n + (u64)num_cus - 1) / (u64)(num_cus > 0 ? num_cus : 1);
    if (ct_fetch()) {
        if (per_cu <= 128) return "c25519::k_mul_base_ctp<5, 256, OUT, split> (constant-time cross-lane fetch, two lanes per scalar)";
        if (per_cu <= 256) return "c25519::k_mul_base_ctp<5, 512, OUT, split> (constant-time cross-lane fetch, two lanes per scalar)";
        if (per_cu <= 512) return "c25519::k_mul_base_ctp<5, 1024, OUT, split> (constant-time cross-lane fetch, two lanes per scalar)";
        return "c25519::k_mul_base_ctp<5, 1024, OUT> (constant-time cross-lane fetch, radix-2^5 tables in LDS)";
    }
    if (per_cu <= 128) return "c25519::k_mul_base_ct_split<256, OUT> (constant-time scan, two lanes per scalar)";
    if (per_cu <= 256) return "c25519::k_mul_base_ct_split<512, OUT> (constant-time scan, two lanes per scalar)";
    if (per_cu <= 512) return "c25519::k_mul_base_ct_split<1024, OUT> (constant-time scan, two lanes per scalar)";
    return "c25519::k_mul_base<5, 1024, OUT, true> (constant-time scan, radix-2^5 tables in LDS)";
}
hipError_t launch_mul_base_ct(const uint8_t *scalars, u64 n, const uint32_t *tab_ct, uint32_t *scratch, uint8_t *out_raw, int num_cus, hipStream_t st, bool p40) {
    if (n == 0) return hipSuccess;
    // small batches: two threads per scalar, one block per compute unit; the block grows with the batch
    const u64 per_cu = (n + (u64)num_cus - 1) / (u64)num_cus;
#ifndef C25519_CT_BS
#define C25519_CT_BS 1024
#endif
    if (ct_fetch()) {
        if (per_cu <= 128) return launch_ctp<C25519_CT_W, 256, true>(scalars, n, tab_ct, scratch, out_raw, num_cus, st, p40);
        if (per_cu <= 256) return launch_ctp<C25519_CT_W, 512, true>(scalars, n, tab_ct, scratch, out_raw, num_cus, st, p40);
        if (per_cu <= 512) return launch_ctp<C25519_CT_W, 1024, true>(scalars, n, tab_ct, scratch, out_raw, num_cus, st, p40);
        return launch_ctp<C25519_CT_W, C25519_CT_BS, false>(scalars, n, tab_ct, scratch, out_raw, num_cus, st, p40);
    }
    if (per_cu <= 128) return launch_ct_split<256>(scalars, n, tab_ct, scratch, out_raw, num_cus, st, p40);
    if (per_cu <= 256) return launch_ct_split<512>(scalars, n, tab_ct, scratch, out_raw, num_cus, st, p40);
    if (per_cu <= 512) return launch_ct_split<1024>(scalars, n, tab_ct, scratch, out_raw, num_cus, st, p40);
    return launch_mul_base_w<C25519_CT_W, C25519_CT_BS, true>(scalars, n, tab_ct, scratch, out_raw, num_cus, st, p40);
}

hipError_t launch_mul_base_p40(int w, const uint8_t *scalars, u64 n, const uint32_t *tab, uint32_t *out40, int num_cus, hipStream_t st) {
    if (n == 0) return hipSuccess;
    switch (w) {
    case 9: return launch_comb<2>(scalars, n, tab, out40, nullptr, num_cus, st);
    case 4: return launch_mul_base_w<4, 512>(scalars, n, tab, out40, nullptr, num_cus, st, true);
    case 5: return launch_mul_base_w<5, 512>(scalars, n, tab, out40, nullptr, num_cus, st, true);
    case 6: return launch_mul_base_w<6, 1024>(scalars, n, tab, out40, nullptr, num_cus, st, true);
    default: break;
    }
    if (w >= 10 && w <= 20) {
        hipLaunchKernelGGL(k_mul_base_wide<2>, dim3(div_up(n, 256)), dim3(256), 0, st, scalars, n, tab, w, (256 + w - 1) / w, out40, (uint8_t *)nullptr);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}


hipError_t launch_x25519(const uint8_t *k, const uint8_t *u, u64 n, uint32_t *scratch, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_x25519, dim3(div_up(n, 256)), dim3(256), 0, st, k, u, n, scratch);
    return hipGetLastError();
}

hipError_t launch_decompress_edwards(const uint8_t *in, u64 n, uint8_t *out_raw, uint8_t *ok, uint32_t *any_bad, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_decompress_edwards, dim3(div_up(n, 256)), dim3(256), 0, st, in, n, out_raw, ok, any_bad);
    return hipGetLastError();
}


hipError_t launch_decompress_ristretto(const uint8_t *in, u64 n, uint8_t *out_raw, uint8_t *ok, uint32_t *any_bad, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_decompress_ristretto, dim3(div_up(n, 256)), dim3(256), 0, st, in, n, out_raw, ok, any_bad);
    return hipGetLastError();
}



// shared: the launch runs BESIDE a latency-bound chain on another stream (verify_batch: hash tree and sort while R is
// decompressed).  The kernel fits three waves per SIMD (164 VGPRs), and three of them leave the SIMD's register file
// no room for a wave of anything else: the chain then waits for the whole decompression (round 2: k_ztree_first 1025 us
// instead of 380, verify_batch 3.38 against 3.08 ms).  A 64 KB LDS reservation caps it at two blocks per CU.
hipError_t launch_prep_compressed(int fmt, const uint8_t *in, uint64_t stride_items, uint64_t n, uint32_t *pts, uint64_t dst0, uint32_t *bad_count, bool shared, hipStream_t st) {
    if (n == 0) return hipSuccess;
    // (shared: the call runs beside verify_batch's hash chain.  Rounds 2-3 reserved 64 KB of LDS there -- two blocks per compute unit,
    //  so that the chain's kernels found room beside 164-VGPR waves: 3.08 against 3.38 ms then; with the chain shortened in round 3
    //  (k_hram 307 -> 178 us, level 0 of the z-tree 333 -> 231) the cap costs more than it buys: 2.73 - 2.74 ms with it, 2.69 without)
    (void)shared;
    const unsigned lds = 0u;
    if (fmt == 0) hipLaunchKernelGGL(k_prep_compressed<0>, dim3(div_up(n, 256)), dim3(256), lds, st, in, stride_items, n, pts, dst0, bad_count);
    else hipLaunchKernelGGL(k_prep_compressed<1>, dim3(div_up(n, 256)), dim3(256), lds, st, in, stride_items, n, pts, dst0, bad_count);
    return hipGetLastError();
}

// clamp_integer (scalar.rs:1407) over a batch: the X25519 secret -> the integer the basepoint is multiplied by
__global__ void __launch_bounds__(256) k_clamp(const uint8_t *__restrict__ in, u64 n, uint8_t *__restrict__ out) {
    const u64 idx = (u64)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    u32 s[8];
    load8(in, idx, s);
    s[0] &= 0xfffffff8u; s[7] &= 0x7fffffffu; s[7] |= 0x40000000u;
    store8(out, idx, s);
}
hipError_t launch_clamp(const uint8_t *in, uint64_t n, uint8_t *out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_clamp, dim3(div_up(n, 256)), dim3(256), 0, st, in, n, out);
    return hipGetLastError();
}



}  // namespace c25519
