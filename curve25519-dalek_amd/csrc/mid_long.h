// Over-long gather lists of the mid path (mid.hip), shared with the accumulation of prepared records (accum.hip k_accumulate_long): the record forms, the
// complete addition of the shuffle trees, and the work loop of one wave per list segment.  Device code of translation units that define C25519_CHAIN 1 and
// include fe26x.h first.
#pragma once

namespace c25519 {

constexpr int MID_REC_Q = 10;                 // a projective Niels record: 4 x 10 tight limbs = 160 bytes = ten 16-byte pieces
struct mid_item { u32 gid, lo, hi, first, lb, nseg, pad0, pad1; };      // one segment of an over-long list: bucket, entries [lo, hi), the bucket's first item, its index, its segments
// entries per segment of an over-long list: one wave, four additions per lane, then a shuffle tree of six -- the tree and the final sum over the segments are the chain
// (2 x 6 complete additions, ~45 us); with the bucket pipeline's 1024 entries per segment the sixteen additions per lane in front of them made it 93 us for the ~n / 2
// entries of verify_batch's carry bucket at 2^14 signatures (profiles/r06_timeline_mid_first.txt)
constexpr u32 MID_LONG_SEG = 256;

// ---- accumulate ----------------------------------------------------------------------------------------------------------------------------------------------------
// P + (neg ? -Q : Q) for a projective Niels record Q = (Y+X, Y-X, Z, 2dT): curve_models.rs:411-429 + :365-373 (8 M), the sign folded into operand selection exactly
// as in ge_madd_signed_p3 (ge26.h) -- -Q swaps the first two entries and negates TT, which only decides which of ZZ2 + TT / ZZ2 - TT plays Z and which plays T of the
// completed point.  The two groups of four independent products are issued in lockstep (fe26x.h).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ ge_p3 ge_add_cached_signed_p3_lockstep(const ge_p3 &p, const feT &qypx, const feT &qymx, const feT &qz, const feT &qt2d, bool neg) {
    // (the first four products as two lockstep pairs: as one group of four the kernel needs 184 registers -- two waves per SIMD, or 17 spilled words at three)
    const lanemask nm = lane_mask(neg);
    feT r4[4];
    {
        feW f2[2]; feL g2[2]; feT r2[2];
        f2[0] = fe_add(p.Y, p.X); f2[1] = fe_sub(p.Y, p.X);
#pragma unroll
        for (int i = 0; i < 10; i++) { g2[0].v[i] = sel_u32(qypx.v[i], qymx.v[i], nm); g2[1].v[i] = sel_u32(qymx.v[i], qypx.v[i], nm); }
        fe_mul_chain_n<2>(r2, f2, g2);
        r4[0] = r2[0]; r4[1] = r2[1];
    }
    {
        feW f2[2]; feL g2[2]; feT r2[2];
        f2[0] = p.T; f2[1] = p.Z; g2[0] = qt2d; g2[1] = qz;
        fe_mul_chain_n<2>(r2, f2, g2);
        r4[2] = r2[0]; r4[3] = r2[1];
    }
    const feT &PP = r4[0], &MM = r4[1], &TT = r4[2], &ZZ = r4[3];
    const feL ZZ2 = fe_twice(ZZ);
    const feL X = fe_sub(PP, MM), Y = fe_add(PP, MM);
    const feL zp = fe_add_lt(ZZ2, TT);
    const feW zm = fe_sub_w(ZZ2, TT);
    feW h4[4]; feL k4[4]; feT o4[4];
#pragma unroll
    for (int i = 0; i < 10; i++) { h4[0].v[i] = sel_u32(zm.v[i], zp.v[i], nm); h4[1].v[i] = sel_u32(zp.v[i], zm.v[i], nm); }
    h4[2] = zm; h4[3] = X;
    k4[0] = X; k4[1] = Y; k4[2] = zp; k4[3] = Y;
    fe_mul_chain_n<4>(o4, h4, k4);
    ge_p3 r;
    r.X = o4[0]; r.Y = o4[1]; r.Z = o4[2]; r.T = o4[3];
    return r;
}
// ... and with the sign LAZILY on the accumulator (fe26x.h ge_madd_lazy_p3_lockstep has the reasoning): `flip` = all ones in the lanes whose stored point changes sides first
__device__ __forceinline__ ge_p3 ge_add_cached_lazy_p3_lockstep(const ge_p3 &p, const feT &qypx, const feT &qymx, const feT &qz, const feT &qt2d, u32 flip) {
    const feL Xs = fe_cond_neg(p.X, flip), Ts = fe_cond_neg(p.T, flip);
    feT r4[4];
    {
        feW f2[2]; feL g2[2]; feT r2[2];
        f2[0] = fe_add_w(feL(p.Y), Xs); f2[1] = fe_sub_w(feL(p.Y), Xs);
        g2[0] = qypx; g2[1] = qymx;
        fe_mul_chain_n<2>(r2, f2, g2);
        r4[0] = r2[0]; r4[1] = r2[1];
    }
    {
        feW f2[2]; feL g2[2]; feT r2[2];
        f2[0] = feW(Ts); f2[1] = p.Z; g2[0] = qt2d; g2[1] = qz;
        fe_mul_chain_n<2>(r2, f2, g2);
        r4[2] = r2[0]; r4[3] = r2[1];
    }
    const feT &PP = r4[0], &MM = r4[1], &TT = r4[2], &ZZ = r4[3];
    const feL ZZ2 = fe_twice(ZZ);
    const feL X = fe_sub(PP, MM), Y = fe_add(PP, MM);
    const feL zp = fe_add_lt(ZZ2, TT);
    const feW zm = fe_sub_w(ZZ2, TT);
    feW h4[4]; feL k4[4]; feT o4[4];
    h4[0] = zm; h4[1] = feW(zp); h4[2] = zm; h4[3] = feW(X);
    k4[0] = X; k4[1] = Y; k4[2] = zp; k4[3] = Y;
    fe_mul_chain_n<4>(o4, h4, k4);
    ge_p3 r;
    r.X = o4[0]; r.Y = o4[1]; r.Z = o4[2]; r.T = o4[3];
    return r;
}
#define mid_madd ge_madd_signed_p3_lockstep
#else           // (the host pass of hipcc only parses the kernels: fe26x.h is device code)
C25519_HD ge_p3 ge_add_cached_signed_p3_lockstep(const ge_p3 &p, const feT &qypx, const feT &qymx, const feT &qz, const feT &qt2d, bool neg) {
    ge_cached q; q.YpX = qypx; q.YmX = qymx; q.Z = qz; q.T2d = qt2d;
    return ge_p1p1_to_p3(ge_add_cached(p, ge_cached_cneg(q, neg)));
}
#define mid_madd ge_madd_signed_p3
C25519_HD ge_p3 ge_add_cached_lazy_p3_lockstep(const ge_p3 &p, const feT &qypx, const feT &qymx, const feT &qz, const feT &qt2d, u32 flip) {
    ge_p3 a = p; a.X = fe_carry(feW(fe_cond_neg(p.X, flip))); a.T = fe_carry(feW(fe_cond_neg(p.T, flip)));
    ge_cached q; q.YpX = qypx; q.YmX = qymx; q.Z = qz; q.T2d = qt2d;
    return ge_p1p1_to_p3(ge_add_cached(a, q));
}
#define ge_madd_lazy_p3_lockstep ge_madd_lazy_p3
#endif
// FMT 0: projective Niels records of 160 bytes (k_mid_front); FMT 1: affine Niels records of 128 bytes (devio.h pts_*: decompressed inputs)
template <int FMT> struct mid_rec;
template <> struct mid_rec<0> {
    uint4 q[MID_REC_Q];
    __device__ __forceinline__ void load(const u32 *recs, u32 idx) {
        const uint4 *src = reinterpret_cast<const uint4 *>(recs) + (u64)MID_REC_Q * idx;
#pragma unroll
        for (int i = 0; i < MID_REC_Q; i++) q[i] = src[i];
    }
    __device__ __forceinline__ ge_p3 add_to(const ge_p3 &acc, bool neg) const {
        const u32 w[40] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w, q[4].x, q[4].y, q[4].z, q[4].w,
                           q[5].x, q[5].y, q[5].z, q[5].w, q[6].x, q[6].y, q[6].z, q[6].w, q[7].x, q[7].y, q[7].z, q[7].w, q[8].x, q[8].y, q[8].z, q[8].w, q[9].x, q[9].y, q[9].z, q[9].w};
        feT a, b, z, t;
#pragma unroll
        for (int i = 0; i < 10; i++) { a.v[i] = w[i]; b.v[i] = w[10 + i]; z.v[i] = w[20 + i]; t.v[i] = w[30 + i]; }
        return ge_add_cached_signed_p3_lockstep(acc, a, b, z, t, neg);
    }
    __device__ __forceinline__ ge_p3 add_to_lazy(const ge_p3 &acc, u32 flip) const {
        const u32 w[40] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w, q[4].x, q[4].y, q[4].z, q[4].w,
                           q[5].x, q[5].y, q[5].z, q[5].w, q[6].x, q[6].y, q[6].z, q[6].w, q[7].x, q[7].y, q[7].z, q[7].w, q[8].x, q[8].y, q[8].z, q[8].w, q[9].x, q[9].y, q[9].z, q[9].w};
        feT a, b, z, t;
#pragma unroll
        for (int i = 0; i < 10; i++) { a.v[i] = w[i]; b.v[i] = w[10 + i]; z.v[i] = w[20 + i]; t.v[i] = w[30 + i]; }
        return ge_add_cached_lazy_p3_lockstep(acc, a, b, z, t, flip);
    }
};
template <> struct mid_rec<1> {
    uint4 q[PTS_Q];
    __device__ __forceinline__ void load(const u32 *recs, u32 idx) {
        const uint4 *src = reinterpret_cast<const uint4 *>(recs) + (u64)PTS_Q * idx;
#pragma unroll
        for (int i = 0; i < PTS_Q; i++) q[i] = src[i];
    }
    __device__ __forceinline__ ge_p3 add_to(const ge_p3 &acc, bool neg) const { return mid_madd(acc, pts_from_q(q), neg); }
    __device__ __forceinline__ ge_p3 add_to_lazy(const ge_p3 &acc, u32 flip) const { return ge_madd_lazy_p3_lockstep(acc, pts_from_q(q), flip); }
};
// a + b (complete addition, edwards.rs:795-800) with the ten-column products (fe26x.h fe_mul_cols_g): the shuffle tree of k_mid_long is a chain of complete additions
// on a lone wave, where a product with ten independent column sums issues a multiply-add every ~6 cycles and the chained form of this translation unit one every ~12
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ ge_p3 mid_add_cols(const ge_p3 &p, const ge_p3 &q) {
    const feT T2d = fe_mul_cols_g(q.T, fe_d2());
    const feT PP = fe_mul_cols_g(fe_add(p.Y, p.X), fe_add(q.Y, q.X)), MM = fe_mul_cols_g(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    const feT TT = fe_mul_cols_g(p.T, T2d), ZZ = fe_mul_cols_g(p.Z, q.Z);
    const feL ZZ2 = fe_twice(ZZ);
    const feL X = fe_sub(PP, MM), Y = fe_add(PP, MM), Z = fe_add_lt(ZZ2, TT);
    const feW T = fe_sub_w(ZZ2, TT);
    ge_p3 r;
    r.X = fe_mul_cols_g(T, X); r.Y = fe_mul_cols_g(feW(Z), Y); r.Z = fe_mul_cols_g(T, Z); r.T = fe_mul_cols_g(feW(X), Y);
    return r;
}
#else
C25519_HD ge_p3 mid_add_cols(const ge_p3 &p, const ge_p3 &q) { return ge_add(p, q); }
#endif
__device__ __forceinline__ ge_p3 mid_wave_sum(ge_p3 acc) {      // complete additions across the 64 lanes; lane 0 ends with the total
#pragma unroll 1
    for (int off = 32; off > 0; off >>= 1) {
        ge_p3 o;
        for (int i = 0; i < 10; i++) {
            o.X.v[i] = __shfl_down(acc.X.v[i], off, 64); o.Y.v[i] = __shfl_down(acc.Y.v[i], off, 64);
            o.Z.v[i] = __shfl_down(acc.Z.v[i], off, 64); o.T.v[i] = __shfl_down(acc.T.v[i], off, 64);
        }
        acc = mid_add_cols(acc, o);
    }
    return acc;
}
// over-long lists (more than long_cap entries: skewed digits -- verify_batch's carry digit puts ~n/2 terms into ONE bucket, equal scalars do it in every window): one wave
// per segment of MID_LONG_SEG entries; the wave that completes a bucket's last segment adds the segment sums and writes the bucket (the bucket lanes skip those buckets).
// wv / nwv: this wave's index among the nwv waves that share the work list (four per block of 256 threads)
template <int FMT>
__device__ __forceinline__ void mid_long_body(const u32 *__restrict__ recs, const u32 *__restrict__ sorted, u64 n, const msm_geom &g, u32 *__restrict__ buckets, u32 max_items,
                                              const mid_item *__restrict__ items, const u32 *__restrict__ counters, u32 *__restrict__ seg_sums, u32 *__restrict__ long_done, u32 wv, u32 nwv) {
    const u32 nitems = counters[0] < max_items ? counters[0] : max_items;
    const u32 lane = threadIdx.x & 63u;
#pragma unroll 1
    for (u32 item = wv; item < nitems; item += nwv) {
        const mid_item it = items[item];
        const u32 *list = sorted + (u64)(it.gid / (u32)g.half) * n;
        ge_p3 acc = ge_identity();
#pragma unroll 1
        for (u32 i = it.lo + lane; i < it.hi; i += 64) {
            const u32 e = list[i];
            mid_rec<FMT> r;
            r.load(recs, e & 0x7fffffffu);
            acc = r.add_to(acc, (e >> 31) != 0);
        }
        acc = mid_wave_sum(acc);
        u32 last = 0;
        if (lane == 0) {
            p40_store(seg_sums, item, acc);
            __threadfence();
            last = atomicAdd(&long_done[it.lb], 1u) == it.nseg - 1u ? 1u : 0u;
        }
        last = (u32)__shfl((int)last, 0, 64);
        if (last) {
            __threadfence();                                // the other segments' sums, written by other waves (other compute units)
            ge_p3 tot = ge_identity();
            bool any = false;
#pragma unroll 1
            for (u32 sg = lane; sg < it.nseg && it.first + sg < max_items; sg += 64) {
                const ge_p3 v = p40_load(seg_sums, it.first + sg);
                tot = any ? mid_add_cols(tot, v) : v;
                any = true;
            }
            if (it.nseg > 1) tot = mid_wave_sum(tot);
            if (lane == 0) p40_store(buckets, it.gid, tot);
        }
    }
}

}  // namespace c25519
