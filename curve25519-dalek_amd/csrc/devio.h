// Device-side I/O helpers shared by the kernels: how the wire formats and scratch records of DESIGN.md §3
// are moved between HBM and registers (16-byte accesses wherever the layout allows).
#pragma once
#include <hip/hip_runtime.h>
#include "ge26.h"

namespace c25519 {

// ------------------------------------------------------------------------------------------------
// 32-byte items: two 16-byte loads/stores per lane; consecutive lanes touch consecutive 32-byte
// items, so every 128-byte line a wave touches is fully used.
__device__ __forceinline__ void load8(const uint8_t *base, u64 idx, u32 w[8]) {
    const uint4 *q = reinterpret_cast<const uint4 *>(base) + 2 * idx;
    uint4 a = q[0], b = q[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void store8(uint8_t *base, u64 idx, const u32 w[8]) {
    uint4 *q = reinterpret_cast<uint4 *>(base) + 2 * idx;
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ feT fe_from_q(const uint4 &a, const uint4 &b) {
    u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fe_from_words(w);
}

// Scratch point record "P32": 32 u32 per point = X[10] Y[10] Z[10] pad[2] (tight limbs), 128 bytes.
__device__ __forceinline__ void p32_store(u32 *scratch, u64 idx, const feT &X, const feT &Y, const feT &Z) {
    uint4 *q = reinterpret_cast<uint4 *>(scratch) + 8 * idx;
    q[0] = make_uint4(X.v[0], X.v[1], X.v[2], X.v[3]);
    q[1] = make_uint4(X.v[4], X.v[5], X.v[6], X.v[7]);
    q[2] = make_uint4(X.v[8], X.v[9], Y.v[0], Y.v[1]);
    q[3] = make_uint4(Y.v[2], Y.v[3], Y.v[4], Y.v[5]);
    q[4] = make_uint4(Y.v[6], Y.v[7], Y.v[8], Y.v[9]);
    q[5] = make_uint4(Z.v[0], Z.v[1], Z.v[2], Z.v[3]);
    q[6] = make_uint4(Z.v[4], Z.v[5], Z.v[6], Z.v[7]);
    q[7] = make_uint4(Z.v[8], Z.v[9], 0u, 0u);
}
__device__ __forceinline__ void p32_load_xy(const u32 *scratch, u64 idx, feT &X, feT &Y) {
    const uint4 *q = reinterpret_cast<const uint4 *>(scratch) + 8 * idx;
    uint4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4];
    X.v[0] = a.x; X.v[1] = a.y; X.v[2] = a.z; X.v[3] = a.w; X.v[4] = b.x; X.v[5] = b.y; X.v[6] = b.z; X.v[7] = b.w;
    X.v[8] = c.x; X.v[9] = c.y; Y.v[0] = c.z; Y.v[1] = c.w;
    Y.v[2] = d.x; Y.v[3] = d.y; Y.v[4] = d.z; Y.v[5] = d.w; Y.v[6] = e.x; Y.v[7] = e.y; Y.v[8] = e.z; Y.v[9] = e.w;
}
__device__ __forceinline__ feT p32_load_z(const u32 *scratch, u64 idx) {
    const uint4 *q = reinterpret_cast<const uint4 *>(scratch) + 8 * idx;
    uint4 f = q[5], g = q[6], h = q[7];
    feT Z;
    Z.v[0] = f.x; Z.v[1] = f.y; Z.v[2] = f.z; Z.v[3] = f.w; Z.v[4] = g.x; Z.v[5] = g.y; Z.v[6] = g.z; Z.v[7] = g.w;
    Z.v[8] = h.x; Z.v[9] = h.y;
    return Z;
}
// 10 tight limbs <-> a 48-byte slot (prefix products of the batched inversion)
__device__ __forceinline__ void fe48_store(u32 *base, u64 idx, const feT &a) {
    uint4 *q = reinterpret_cast<uint4 *>(base) + 3 * idx;
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
    q[2] = make_uint4(a.v[8], a.v[9], 0u, 0u);
}
__device__ __forceinline__ feT fe48_load(const u32 *base, u64 idx) {
    const uint4 *q = reinterpret_cast<const uint4 *>(base) + 3 * idx;
    uint4 a = q[0], b = q[1], c = q[2];
    feT r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = c.x; r.v[9] = c.y;
    return r;
}

// raw 160-byte EdwardsPoint (fmt 2): {X,Y,Z,T} x 5 x u64 radix-2^51, limbs < 2^52
// EXACT for every u64 limb value: the element is sum_i l_i 2^(51 i) mod p (u64/field.rs:43-52), whatever the magnitudes -- the
// reference's own operations keep limbs below 2^52 (its debug_assert!s, field.rs:162-166), but a caller that hands over sums it
// never reduced is still given the value its limbs denote, not a truncation: bits 51..63 of limb i are a 13-bit carry into limb
// i+1 (x19 from the top limb), taken before the 26/25 split.
C25519_HD feT fe_from_limbs51(const u64 l[5]) {
    feW t;
    u32 hi[5];
    for (int i = 0; i < 5; i++) { t.v[2 * i] = (u32)l[i] & M26; t.v[2 * i + 1] = (u32)(l[i] >> 26) & M25; hi[i] = (u32)(l[i] >> 32) >> 19; }
    t.v[0] += 19u * hi[4];
    for (int i = 1; i < 5; i++) t.v[2 * i] += hi[i - 1];
    return fe_carry(t);
}
__device__ __forceinline__ void fe_to_limbs51(const feT &a, u64 l[5]) {
    u32 c[10];
    fe_canonical_limbs(a, c);
    for (int i = 0; i < 5; i++) l[i] = (u64)c[2 * i] | ((u64)c[2 * i + 1] << 26);
}
__device__ __forceinline__ void raw160_store(uint8_t *out, u64 idx, const ge_p3 &p) {
    u64 l[20];
    fe_to_limbs51(p.X, l); fe_to_limbs51(p.Y, l + 5); fe_to_limbs51(p.Z, l + 10); fe_to_limbs51(p.T, l + 15);
    ulonglong2 *q = reinterpret_cast<ulonglong2 *>(out) + 10 * idx;
    for (int i = 0; i < 10; i++) q[i] = make_ulonglong2(l[2 * i], l[2 * i + 1]);
}
__device__ __forceinline__ ge_p3 raw160_load(const uint8_t *in, u64 idx) {
    const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(in) + 10 * idx;
    u64 l[20];
    for (int i = 0; i < 10; i++) { ulonglong2 v = q[i]; l[2 * i] = v.x; l[2 * i + 1] = v.y; }
    ge_p3 p;
    p.X = fe_from_limbs51(l); p.Y = fe_from_limbs51(l + 5); p.Z = fe_from_limbs51(l + 10); p.T = fe_from_limbs51(l + 15);
    return p;
}

// ---- affine Niels point as stored for gathers: (y+x, y-x, 2dxy) as 3 x 10 tight LIMBS + 8 bytes of padding = 128 B,
//      i.e. exactly one cache line per point and nothing to unpack; the sign of a signed digit is applied by
//      ge_madd_signed_p3 (operand swap), not to the stored data --------------------------------------------------------
constexpr int PTS_BYTES = 128, PTS_Q = PTS_BYTES / 16;
__device__ __forceinline__ void pts_store(u32 *pts, u64 idx, const feT &x, const feT &y) {
    feT a = fe_carry(fe_add(y, x)), b = fe_carry(fe_sub(y, x)), c = fe_mul(fe_mul(x, y), fe_d2());
    u32 t[32];
    for (int i = 0; i < 10; i++) { t[i] = a.v[i]; t[10 + i] = b.v[i]; t[20 + i] = c.v[i]; }
    t[30] = 0; t[31] = 0;
    uint4 *q = reinterpret_cast<uint4 *>(pts) + PTS_Q * idx;
    for (int i = 0; i < PTS_Q; i++) q[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
}
// the same record as eight 16-byte pieces in registers (k_prep_raw2 stores them through LDS)
__device__ __forceinline__ void pts_pieces(const feT &x, const feT &y, uint4 q[PTS_Q]) {
    feT a = fe_carry(fe_add(y, x)), b = fe_carry(fe_sub(y, x)), c = fe_mul(fe_mul(x, y), fe_d2());
    u32 t[32];
    for (int i = 0; i < 10; i++) { t[i] = a.v[i]; t[10 + i] = b.v[i]; t[20 + i] = c.v[i]; }
    t[30] = 0; t[31] = 0;
    for (int i = 0; i < PTS_Q; i++) q[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
}
__device__ __forceinline__ ge_aniels pts_from_q(const uint4 q[PTS_Q]) {
    u32 t[32] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w,
                 q[4].x, q[4].y, q[4].z, q[4].w, q[5].x, q[5].y, q[5].z, q[5].w, q[6].x, q[6].y, q[6].z, q[6].w, q[7].x, q[7].y, q[7].z, q[7].w};
    ge_aniels A;
    for (int i = 0; i < 10; i++) { A.ypx.v[i] = t[i]; A.ymx.v[i] = t[10 + i]; A.xy2d.v[i] = t[20 + i]; }
    return A;
}
__device__ __forceinline__ ge_aniels pts_load(const u32 *pts, u64 idx) {
    const uint4 *src = reinterpret_cast<const uint4 *>(pts) + PTS_Q * idx;
    uint4 q[PTS_Q];
    for (int i = 0; i < PTS_Q; i++) q[i] = src[i];
    return pts_from_q(q);
}
// extended point as 40 u32 tight limbs (bucket sums, partial results)
__device__ __forceinline__ void p40_store(u32 *base, u64 idx, const ge_p3 &p) {
    uint4 *q = reinterpret_cast<uint4 *>(base) + 10 * idx;
    u32 t[40];
    for (int i = 0; i < 10; i++) { t[i] = p.X.v[i]; t[10 + i] = p.Y.v[i]; t[20 + i] = p.Z.v[i]; t[30 + i] = p.T.v[i]; }
    for (int i = 0; i < 10; i++) q[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
}
__device__ __forceinline__ ge_p3 p40_load(const u32 *base, u64 idx) {
    const uint4 *q = reinterpret_cast<const uint4 *>(base) + 10 * idx;
    u32 t[40];
    for (int i = 0; i < 10; i++) { uint4 v = q[i]; t[4 * i] = v.x; t[4 * i + 1] = v.y; t[4 * i + 2] = v.z; t[4 * i + 3] = v.w; }
    ge_p3 p;
    for (int i = 0; i < 10; i++) { p.X.v[i] = t[i]; p.Y.v[i] = t[10 + i]; p.Z.v[i] = t[20 + i]; p.T.v[i] = t[30 + i]; }
    return p;
}


// one field of a raw 160-byte point (40 bytes, only 8-byte aligned)
__device__ __forceinline__ feT raw160_fe(const uint8_t *in, u64 idx, int which) {
    const u64 *p = reinterpret_cast<const u64 *>(in + idx * 160 + which * 40);
    u64 l[5] = {p[0], p[1], p[2], p[3], p[4]};
    return fe_from_limbs51(l);
}

// Z of a raw point stored as exactly (1, 0, 0, 0, 0): what CompressedEdwardsY::decompress leaves (edwards.rs:256)
__device__ __forceinline__ bool raw160_z_is_one(const uint8_t *in, u64 idx) {
    const u64 *p = reinterpret_cast<const u64 *>(in + idx * 160 + 80);
    return (p[0] == 1) & ((p[1] | p[2] | p[3] | p[4]) == 0);
}

}  // namespace c25519
