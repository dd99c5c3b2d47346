// Per-item (not batched-equation) paths that sit either side of the MSM on the reference's hot path:
//
//   c25519_mul_batch            out[i] = s_i * P_i          variable_base::mul (backend/serial/scalar_mul/
//                                                           variable_base.rs:11-47; edwards.rs:890-911)
//   ed25519_verify_each         per-signature verify / verify_strict, one status byte per signature:
//                               R' = [s]B - [k]A, compress(R') == R bytes  (verifying.rs:203-214,
//                               :359-382, RCompute::finish :549-556 over vartime_double_base.rs:23-72)
//   ed25519_keygen_batch / ed25519_sign_batch   consumers of the fixed-base kernel
//                               (verifying.rs:97-101, signing.rs:878-905; RFC 8032 5.1.5 / 5.1.6)
//
// GPU schedule of the double-base computation: the reference interleaves NAF(5) of k on a per-call
// table of A with NAF(8) of s on a static table of B inside one 256-step doubling chain.  One chain per
// lane with per-lane NAF patterns would diverge on every step, so the two halves are split:
// [s]B comes from the LDS fixed-base table kernel (43 mixed additions, no doublings, k_mul_base) and
// [k](-A) from a regular radix-16 fixed-window ladder (63 x 4 doublings + 64 additions, the schedule of
// variable_base.rs) whose per-lane table of 8 multiples lives in an L2-resident scratch array laid out
// [entry][quad][lane] so that a wave's reads stay within at most 9 contiguous 1-KiB rows.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <string.h>
#include <vector>
#include "../../include/c25519_hip.h"
#include "devio.h"
#include "sc_sha.h"
#include "sc28.h"
#include "kernels.h"
#include "ctx.h"
#include "ffi.h"

using namespace c25519;
#define EXPORT extern "C" __attribute__((visibility("default")))
#define HIPCHK(call)                                                \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return c25519_fail(ctx, _e, #call);   \
    } while (0)

namespace c25519 {

// per-lane table entry j of lane L: quad q at  tab[((j * 10 + q) * stride + L)]   (uint4 units)
__device__ __forceinline__ void tab_store(uint4 *tab, u64 stride, u64 lane, int j, const ge_cached &c) {
    u32 t[40];
    for (int i = 0; i < 10; i++) { t[i] = c.YpX.v[i]; t[10 + i] = c.YmX.v[i]; t[20 + i] = c.Z.v[i]; t[30 + i] = c.T2d.v[i]; }
    for (int q = 0; q < 10; q++) tab[((u64)(j * 10 + q)) * stride + lane] = make_uint4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
}
__device__ __forceinline__ ge_cached tab_load(const uint4 *tab, u64 stride, u64 lane, u32 j) {
    u32 t[40];
    for (int q = 0; q < 10; q++) { uint4 v = tab[((u64)(j * 10 + q)) * stride + lane]; t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w; }
    ge_cached c;
    for (int i = 0; i < 10; i++) { c.YpX.v[i] = t[i]; c.YmX.v[i] = t[10 + i]; c.Z.v[i] = t[20 + i]; c.T2d.v[i] = t[30 + i]; }
    return c;
}
// ================================================================================================
// variable-base scalar multiplication, radix 16 (variable_base.rs:11-47), one (scalar, point) per lane
//   IN_FMT 0: CompressedEdwardsY, 2: raw 160-byte;  NEGATE: multiply -P (verify: [k](-A))
//   ok[i] = 0 if the point does not decompress (result unspecified)
// ================================================================================================
//   CT: constant-time table access for secret scalars -- all nine entries of the lane's table are read and the wanted
//   one kept with selects (LookupTable::select, window.rs:54-76), so no address depends on the scalar
template <int IN_FMT, bool NEGATE, bool CT>
__global__ void __launch_bounds__(256) k_var_base(const uint8_t *__restrict__ scalars, const uint8_t *__restrict__ points, u64 n,
                                                  uint4 *__restrict__ tab, u32 *__restrict__ out40, uint8_t *__restrict__ ok) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    ge_p3 P;
    bool good = true;
    if (IN_FMT == 0) { u32 w[8]; load8(points, idx, w); good = ge_decompress(P, w); }
    else P = raw160_load(points, idx);
    if (NEGATE) P = ge_neg(P);
    if (ok) ok[idx] = good ? 1 : 0;
    // table: entry 0 = identity, entry j = j*P (j = 1..8), window.rs:97-104
    {
        ge_cached id;
        id.YpX = fe_one(); id.YmX = fe_one(); id.Z = fe_one(); id.T2d = fe_zero();
        tab_store(tab, stride, idx, 0, id);
        ge_cached c1 = ge_p3_to_cached(P);
        tab_store(tab, stride, idx, 1, c1);
        ge_p3 acc = P;
#pragma unroll 1
        for (int j = 2; j <= 8; j++) {
            acc = ge_p1p1_to_p3(ge_add_cached(acc, c1));
            tab_store(tab, stride, idx, j, ge_p3_to_cached(acc));
        }
    }
    // digits: nibble_i(s') - 8 with s' = s + 0x0888...8 (top nibble left unsigned), scalar.rs:1019-1051
    u32 s[8];
    load8(scalars, idx, s);
    {
        u64 carry = 0;
        for (int i = 0; i < 8; i++) { u64 v = (u64)s[i] + (i == 7 ? 0x08888888u : 0x88888888u) + carry; s[i] = (u32)v; carry = v >> 32; }
    }
    ge_p3 acc = ge_identity();
#pragma unroll 1
    for (int i = 63; i >= 0; i--) {
        u32 nib = s[7] >> 28;
#pragma unroll
        for (int k = 7; k > 0; k--) s[k] = (s[k] << 4) | (s[k - 1] >> 28);
        s[0] <<= 4;
        int d = (i == 63) ? (int)nib : (int)nib - 8;
        if (i != 63) acc = ge_mul_by_pow_2(acc, 4);
        bool neg = d < 0;
        u32 mag = (u32)(neg ? -d : d);
        ge_cached c;
        if (CT) {
            c = tab_load(tab, stride, idx, 0);
#pragma unroll 1
            for (u32 ent = 1; ent <= 8; ent++) {
                const ge_cached t = tab_load(tab, stride, idx, ent);
                const bool hit = ent == mag;
                c.YpX = fe_select(c.YpX, t.YpX, hit); c.YmX = fe_select(c.YmX, t.YmX, hit);
                c.Z = fe_select(c.Z, t.Z, hit); c.T2d = fe_select(c.T2d, t.T2d, hit);
            }
        } else {
            c = tab_load(tab, stride, idx, mag);
        }
        acc = ge_p1p1_to_p3(ge_add_cached(acc, ge_cached_cneg(c, neg)));
    }
    p40_store(out40, idx, acc);
}

// P40 -> raw160 / P32 (for the batched compressor)
__global__ void __launch_bounds__(256) k_p40_to_raw(const u32 *__restrict__ in40, const u32 *__restrict__ b40, u64 n, uint8_t *__restrict__ out_raw) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    ge_p3 p = b40 ? ge_add(p40_load(in40, idx), p40_load(b40, idx)) : p40_load(in40, idx);
    u64 l[20];
    const feT *f[4] = {&p.X, &p.Y, &p.Z, &p.T};
    for (int c = 0; c < 4; c++) { u32 cl[10]; fe_canonical_limbs(*f[c], cl); for (int i = 0; i < 5; i++) l[5 * c + i] = (u64)cl[2 * i] | ((u64)cl[2 * i + 1] << 26); }
    ulonglong2 *q = reinterpret_cast<ulonglong2 *>(out_raw) + 10 * idx;
    for (int i = 0; i < 10; i++) q[i] = make_ulonglong2(l[2 * i], l[2 * i + 1]);
}
// sum of two P40 arrays -> P32 scratch (X,Y,Z) for the batched compressor
__global__ void __launch_bounds__(256) k_p40_add_to_p32(const u32 *__restrict__ a40, const u32 *__restrict__ b40, u64 n, u32 *__restrict__ scratch) {
    u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    ge_p3 r = b40 ? ge_add(p40_load(a40, idx), p40_load(b40, idx)) : p40_load(a40, idx);
    uint4 *q = reinterpret_cast<uint4 *>(scratch) + 8 * idx;
    q[0] = make_uint4(r.X.v[0], r.X.v[1], r.X.v[2], r.X.v[3]); q[1] = make_uint4(r.X.v[4], r.X.v[5], r.X.v[6], r.X.v[7]);
    q[2] = make_uint4(r.X.v[8], r.X.v[9], r.Y.v[0], r.Y.v[1]); q[3] = make_uint4(r.Y.v[2], r.Y.v[3], r.Y.v[4], r.Y.v[5]);
    q[4] = make_uint4(r.Y.v[6], r.Y.v[7], r.Y.v[8], r.Y.v[9]); q[5] = make_uint4(r.Z.v[0], r.Z.v[1], r.Z.v[2], r.Z.v[3]);
    q[6] = make_uint4(r.Z.v[4], r.Z.v[5], r.Z.v[6], r.Z.v[7]); q[7] = make_uint4(r.Z.v[8], r.Z.v[9], 0u, 0u);
}

// ================================================================================================
// per-signature verification glue
// ================================================================================================
// k_i = SHA-512(R||A||M) mod l from the 64-byte digests (scalar.rs:248); s canonical flag
__global__ void __launch_bounds__(256) k_hram_reduce(const uint8_t *__restrict__ hram, const uint8_t *__restrict__ sigs, u64 n,
                                                     uint8_t *__restrict__ kscal, uint8_t *__restrict__ sscal, uint8_t *__restrict__ s_ok) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 *hw = reinterpret_cast<const u32 *>(hram) + 16 * i;
    u32 h16[16];
    for (int j = 0; j < 16; j++) h16[j] = hw[j];
    u32 kw[8];
    sc28_to_words(sc28_from_wide(h16), kw);
    store8(kscal, i, kw);
    u32 s[8];
    load8(sigs, 2 * i + 1, s);
    bool canon = sc28_words_canonical(s);
    s_ok[i] = canon ? 1 : 0;
    if (!canon) { for (int j = 0; j < 8; j++) s[j] = 0; }   // keep the fixed-base kernel's precondition (< 2^255)
    store8(sscal, i, s);
}
// strict mode: small-order checks on R and A (verifying.rs:371-374, edwards.rs:1405)
__global__ void __launch_bounds__(256) k_strict_checks(const uint8_t *__restrict__ sigs, const uint8_t *__restrict__ pks, u64 n, uint8_t *__restrict__ strict_bad) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 r[8], a[8];
    load8(sigs, 2 * i, r);
    load8(pks, i, a);
    ge_p3 R, A;
    bool okR = ge_decompress(R, r), okA = ge_decompress(A, a);
    bool bad = !okR;
    bad |= ge_is_identity(ge_mul_by_pow_2(R, 3));
    bad |= okA && ge_is_identity(ge_mul_by_pow_2(A, 3));
    strict_bad[i] = bad ? 1 : 0;
}
// status_i from the flags and compress(R') == R bytes   (verifying.rs:211, :377-381)
__global__ void __launch_bounds__(256) k_verdict(const uint8_t *__restrict__ sigs, const uint8_t *__restrict__ rcheck, const uint8_t *__restrict__ a_ok,
                                                 const uint8_t *__restrict__ s_ok, const uint8_t *__restrict__ strict_bad, u64 n, uint8_t *__restrict__ status) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 r[8], c[8];
    load8(sigs, 2 * i, r);
    load8(rcheck, i, c);
    u32 d = 0;
    for (int j = 0; j < 8; j++) d |= r[j] ^ c[j];
    uint8_t st = C25519_OK;
    if (!a_ok[i]) st = C25519_NONE;
    else if (!s_ok[i]) st = C25519_SCALAR_FORMAT;
    else if ((strict_bad && strict_bad[i]) || d != 0) st = C25519_VERIFY;
    status[i] = st;
}

// ================================================================================================
// batched key generation and signing (RFC 8032 5.1.5 / 5.1.6; signing.rs:878-905, verifying.rs:97-101)
// ================================================================================================
// h = SHA-512(seed): a = clamp(h[0..32]) -> scal_a;  prefix = h[32..64]
__global__ void __launch_bounds__(256) k_expand_seed(const uint8_t *__restrict__ seeds, u64 n, uint8_t *__restrict__ scal_a, uint8_t *__restrict__ prefix) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 sd[8];
    load8(seeds, i, sd);
    u64 hs[8], w[16];
    sha512_init(hs);
    for (int j = 0; j < 4; j++) w[j] = bswap64((u64)sd[2 * j] | ((u64)sd[2 * j + 1] << 32));
    w[4] = 0x8000000000000000ull;
    for (int j = 5; j < 15; j++) w[j] = 0;
    w[15] = 32 * 8;
    sha512_compress(hs, w);
    u32 d[16];
    sha512_digest_words(hs, d);
    d[0] &= 0xfffffff8u; d[7] &= 0x7fffffffu; d[7] |= 0x40000000u;   // clamp_integer, scalar.rs:1407
    store8(scal_a, i, d);
    store8(prefix, i, d + 8);
}
// r_i = SHA-512(prefix_i || M_i) mod l
__global__ void __launch_bounds__(256) k_sign_nonce(const uint8_t *__restrict__ prefix, const uint8_t *__restrict__ msgs, const u64 *__restrict__ msg_off, u64 msgs_len, u64 n,
                                                    uint8_t *__restrict__ rscal) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 p[8];
    load8(prefix, i, p);
    sha512_stream st;
    st.init();
    for (int j = 0; j < 4; j++) st.w[j] = bswap64((u64)p[2 * j] | ((u64)p[2 * j + 1] << 32));
    st.fill = 32; st.total = 32;
    const u64 o0 = msg_off[i], o1 = msg_off[i + 1];
    const u64 len = (o0 <= o1 && o1 <= msgs_len) ? o1 - o0 : 0;        // bad offsets: flagged by k_hram, nothing read out of bounds
    const uint8_t *m = msgs + o0;
    st.put_bytes(m, len);
    st.finish();
    u32 d[16], r[8];
    sha512_digest_words(st.h, d);
    sc28_to_words(sc28_from_wide(d), r);
    store8(rscal, i, r);
}
// Ed25519ph: r_i = SHA-512(dom2 || prefix_i || PH(M_i)) mod l (signing.rs:952-960); ph: n x 64 bytes
__global__ void __launch_bounds__(256) k_sign_nonce_dom(const uint8_t *__restrict__ dom, u32 dom_len, const uint8_t *__restrict__ prefix, const uint8_t *__restrict__ ph, u64 n,
                                                        uint8_t *__restrict__ rscal) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sha512_stream st;
    st.init();
    st.put_bytes(dom, dom_len);
    st.put_bytes(prefix + 32 * i, 32);
    st.put_bytes(ph + 64 * i, 64);
    st.finish();
    u32 d[16], r[8];
    sha512_digest_words(st.h, d);
    sc28_to_words(sc28_from_wide(d), r);
    store8(rscal, i, r);
}
// s_i = k_i * a_i + r_i mod l;  sig_i = R_i || s_i      (k_i = H(R||A||M) mod l given as 64-byte digests)
__global__ void __launch_bounds__(256) k_sign_finish(const uint8_t *__restrict__ hram, const uint8_t *__restrict__ scal_a, const uint8_t *__restrict__ rscal,
                                                     const uint8_t *__restrict__ Renc, u64 n, uint8_t *__restrict__ sigs) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 *hw = reinterpret_cast<const u32 *>(hram) + 16 * i;
    u32 h16[16];
    for (int j = 0; j < 16; j++) h16[j] = hw[j];
    u32 a[8], r[8], R[8], s[8];
    load8(scal_a, i, a); load8(rscal, i, r); load8(Renc, i, R);
    // S = r + k a mod l (signing.rs:899): k reduced, a the clamped integer as it is (k a < 2^508: sc28_mul reduces the full product)
    const sc28 sv = sc28_add(sc28_mul(sc28_from_wide(h16).v, sc28_from_words(a).v), sc28_from_words(r));
    sc28_to_words(sv, s);
    store8(sigs, 2 * i, R);
    store8(sigs, 2 * i + 1, s);
}
// signatures with R filled in only (so k_hram can hash R || A || M): copy R into sig slots
__global__ void __launch_bounds__(256) k_place_R(const uint8_t *__restrict__ Renc, u64 n, uint8_t *__restrict__ sigs) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 R[8];
    load8(Renc, i, R);
    store8(sigs, 2 * i, R);
}

}  // namespace c25519

static inline unsigned dup(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

// ---- variable base ------------------------------------------------------------------------------------
// ct: the scalars are secret (constant-address table scan); false for public scalars (per-signature verification)
static int32_t var_base_launch(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, bool negate, bool ct, uint32_t *out40, uint8_t *d_ok) {
    const unsigned grid = dup(n, 256);
    const uint64_t stride = (uint64_t)grid * 256;
    int32_t r = ctx_reserve(ctx, ctx->tmp_d, stride * 9 * 160);
    if (r) return r;
    uint4 *tab = (uint4 *)ctx->tmp_d.p;
    hipStream_t st = ctx->stream;
#define VB(F, N, C) hipLaunchKernelGGL((k_var_base<F, N, C>), dim3(grid), dim3(256), 0, st, d_scalars, d_points, n, tab, out40, d_ok)
    if (in_fmt == C25519_FMT_EDWARDS_Y) {
        if (negate) { if (ct) VB(0, true, true); else VB(0, true, false); }
        else { if (ct) VB(0, false, true); else VB(0, false, false); }
    } else if (in_fmt == C25519_FMT_RAW160) {
        if (negate) { if (ct) VB(2, true, true); else VB(2, true, false); }
        else { if (ct) VB(2, false, true); else VB(2, false, false); }
    } else { ctx->err = "mul_batch: in_fmt must be 0 or 2"; return -(int32_t)hipErrorInvalidValue; }
#undef VB
    HIPCHK(hipGetLastError());
    if (ct) HIPCHK(hipMemsetAsync(tab, 0, stride * 9 * 160, st));     // the per-lane tables are multiples of a possibly secret point path: wipe
    return C25519_OK;
}

int32_t mul_batch_impl(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, int out_fmt, uint8_t *d_out, uint8_t *d_ok, bool ct) {
    HIPCHK(hipSetDevice(ctx->device));
    if (out_fmt != C25519_FMT_EDWARDS_Y && out_fmt != C25519_FMT_RAW160) { ctx->err = "mul_batch: out_fmt must be 0 or 2"; return -(int32_t)hipErrorInvalidValue; }
    if (n == 0) return C25519_OK;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_e, n * 160 + n + 256))) return r;
    uint32_t *p40 = (uint32_t *)ctx->tmp_e.p;
    uint8_t *okbuf = d_ok ? d_ok : (uint8_t *)ctx->tmp_e.p + n * 160;
    hipEvent_t *ring = ctx_ring_item(ctx);
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(hipEventRecord(ring[0], ctx->stream));
    if ((r = var_base_launch(ctx, d_scalars, d_points, n, in_fmt, false, ct, p40, okbuf))) return r;
    HIPCHK(hipEventRecord(ring[1], ctx->stream));
    if (out_fmt == C25519_FMT_RAW160) {
        hipLaunchKernelGGL(k_p40_to_raw, dim3(dup(n, 256)), dim3(256), 0, ctx->stream, p40, (const uint32_t *)nullptr, n, d_out);
    } else {
        if ((r = ctx_reserve(ctx, ctx->scratch, n * 128)) || (r = ctx_reserve(ctx, ctx->prefix, n * 48))) return r;
        stream_wipe wipe(ctx->stream);                    // (declared before the launches: also wiped if one of them fails)
        if (ct) { wipe.add(ctx->scratch.p, n * 128); wipe.add(ctx->prefix.p, n * 48); }
        hipLaunchKernelGGL(k_p40_add_to_p32, dim3(dup(n, 256)), dim3(256), 0, ctx->stream, p40, (const uint32_t *)nullptr, n, (uint32_t *)ctx->scratch.p);
        HIPCHK(launch_compress_p32((const uint32_t *)ctx->scratch.p, (uint32_t *)ctx->prefix.p, n, d_out, ctx->stream));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ring[2], ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    return C25519_OK;
}
EXPORT int32_t c25519_mul_batch_dev(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, int out_fmt, uint8_t *d_out, uint8_t *d_ok) {
    return mul_batch_impl(ctx, d_scalars, d_points, n, in_fmt, out_fmt, d_out, d_ok, !(ctx->flags & C25519_FLAG_VARTIME_TABLES));
}
// ---- order checks (edwards.rs:1405-1437) --------------------------------------------------------------------------------------
namespace c25519 {
// flags[i] = decodes | small_order << 1 (8 P = O); raw points always decode
template <int FMT>
__global__ void __launch_bounds__(256) k_small_order(const uint8_t *__restrict__ in, u64 n, uint8_t *__restrict__ flags) {
    const u64 idx = (u64)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    ge_p3 P;
    bool ok = true;
    if (FMT == 0) { u32 w[8]; load8(in, idx, w); ok = ge_decompress(P, w); }
    else P = raw160_load(in, idx);
    const bool small = ge_is_identity(ge_mul_by_pow_2(P, 3));
    flags[idx] = ok ? (uint8_t)(1u | (small ? 2u : 0u)) : (uint8_t)0;
}
// the group order l as a 32-byte scalar for every lane (edwards.rs:1436 BASEPOINT_ORDER)
__global__ void __launch_bounds__(256) k_fill_order(u64 n, uint8_t *__restrict__ out) {
    const u64 idx = (u64)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    u32 l[8];
    sc28_l_words(l);
    store8(out, idx, l);
}
// flags[i] |= 4 where the point decodes (ok[i]) and enc[i] is the identity's encoding (y = 1, sign 0)
__global__ void __launch_bounds__(256) k_flag_identity_enc(const uint8_t *__restrict__ enc, const uint8_t *__restrict__ ok, u64 n, uint8_t *__restrict__ flags, int fresh) {
    const u64 idx = (u64)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    u32 w[8];
    load8(enc, idx, w);
    const bool id = w[0] == 1u && (w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0u;
    const uint8_t prev = fresh ? (uint8_t)(ok[idx] ? 1 : 0) : flags[idx];
    flags[idx] = ok[idx] ? (uint8_t)(prev | (id ? 4u : 0u)) : (uint8_t)0;
}
}  // namespace c25519
EXPORT int32_t c25519_point_order_checks_batch_dev(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, int which, uint8_t *d_flags) {
    HIPCHK(hipSetDevice(ctx->device));
    if (in_fmt != C25519_FMT_EDWARDS_Y && in_fmt != C25519_FMT_RAW160) { ctx->err = "point_order_checks: in_fmt must be 0 or 2"; return -(int32_t)hipErrorInvalidValue; }
    if (!(which & (C25519_POINT_SMALL_ORDER | C25519_POINT_TORSION_FREE))) { ctx->err = "point_order_checks: nothing to check"; return -(int32_t)hipErrorInvalidValue; }
    if (n == 0) return C25519_OK;
    const bool small = (which & C25519_POINT_SMALL_ORDER) != 0;
    if (small) {
        if (in_fmt == C25519_FMT_EDWARDS_Y) hipLaunchKernelGGL(k_small_order<0>, dim3(dup(n, 256)), dim3(256), 0, ctx->stream, d_points, n, d_flags);
        else hipLaunchKernelGGL(k_small_order<2>, dim3(dup(n, 256)), dim3(256), 0, ctx->stream, d_points, n, d_flags);
        HIPCHK(hipGetLastError());
    }
    if (which & C25519_POINT_TORSION_FREE) {
        int32_t r;
        if ((r = ctx_reserve(ctx, ctx->tmp_c2, n * 65 + 64))) return r;
        uint8_t *l = (uint8_t *)ctx->tmp_c2.p, *enc = l + n * 32, *ok = enc + n * 32;
        hipLaunchKernelGGL(k_fill_order, dim3(dup(n, 256)), dim3(256), 0, ctx->stream, n, l);
        if ((r = mul_batch_impl(ctx, l, d_points, n, in_fmt, C25519_FMT_EDWARDS_Y, enc, ok, false))) return r;     // public points, public scalar: the fast tables
        hipLaunchKernelGGL(k_flag_identity_enc, dim3(dup(n, 256)), dim3(256), 0, ctx->stream, enc, ok, n, d_flags, small ? 0 : 1);
        HIPCHK(hipGetLastError());
    }
    return C25519_OK;
}
EXPORT int32_t c25519_point_order_checks_batch(c25519_ctx *ctx, const uint8_t *points, uint64_t n, int in_fmt, int which, uint8_t *flags) {
    HIPCHK(hipSetDevice(ctx->device));
    if (in_fmt != C25519_FMT_EDWARDS_Y && in_fmt != C25519_FMT_RAW160) { ctx->err = "point_order_checks: in_fmt must be 0 or 2"; return -(int32_t)hipErrorInvalidValue; }
    const size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, n * psz + 16)) || (r = ctx_reserve(ctx, ctx->tmp_b, n + 16))) return r;
    uint8_t *dp = (uint8_t *)ctx->tmp_a.p, *dfl = (uint8_t *)ctx->tmp_b.p;
    const ffi_in in = {points, dp, psz};
    const ffi_out o = {flags, dfl, 1};
    return ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 16), &in, 1, &o, 1, [&](uint64_t lo, uint64_t m) -> int32_t {
        return c25519_point_order_checks_batch_dev(ctx, dp + lo * psz, m, in_fmt, which, dfl + lo);
    });
}
static int32_t mul_batch_host(c25519_ctx *ctx, const uint8_t *scalars, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out, uint8_t *ok, bool clamp) {
    HIPCHK(hipSetDevice(ctx->device));
    const size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32, osz = out_fmt == C25519_FMT_RAW160 ? 160 : 32;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, n * 32 + 16)) || (r = ctx_reserve(ctx, ctx->tmp_b, n * psz + 16)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * osz + n + 16))) return r;
    uint8_t *ds = (uint8_t *)ctx->tmp_a.p, *dp = (uint8_t *)ctx->tmp_b.p, *dout = (uint8_t *)ctx->tmp_c.p, *dok = dout + n * osz;
    const bool secret = ctx_secret_default(ctx);
    stream_wipe wipe(ctx->stream);
    if (secret) { wipe.add(ds, n * 32); wipe.add(dout, n * osz); }        // staged secret scalars and the products (e.g. shared secrets)
    const ffi_in in[2] = {{scalars, ds, 32}, {points, dp, psz}};
    const ffi_out o[2] = {{out, dout, osz}, {ok, dok, 1}};
    return ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 16), in, 2, o, 2, [&](uint64_t lo, uint64_t m) -> int32_t {
        if (clamp) HIPCHK(launch_clamp(ds + lo * 32, m, ds + lo * 32, ctx->stream));
        return mul_batch_impl(ctx, ds + lo * 32, dp + lo * psz, m, in_fmt, out_fmt, dout + lo * osz, dok + lo, secret);
    });
}
EXPORT int32_t c25519_mul_batch(c25519_ctx *ctx, const uint8_t *scalars, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out, uint8_t *ok) {
    return mul_batch_host(ctx, scalars, points, n, in_fmt, out_fmt, out, ok, false);
}
// EdwardsPoint::mul_clamped (edwards.rs:932-946): out[i] = clamp_integer(bytes[i]) * points[i], the scalar NOT reduced mod l
EXPORT int32_t c25519_mul_clamped_batch_dev(c25519_ctx *ctx, const uint8_t *d_bytes, const uint8_t *d_points, uint64_t n, int in_fmt, int out_fmt, uint8_t *d_out, uint8_t *d_ok) {
    HIPCHK(hipSetDevice(ctx->device));
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_c2, n * 32 + 16))) return r;
    HIPCHK(launch_clamp(d_bytes, n, (uint8_t *)ctx->tmp_c2.p, ctx->stream));
    r = mul_batch_impl(ctx, (const uint8_t *)ctx->tmp_c2.p, d_points, n, in_fmt, out_fmt, d_out, d_ok, ctx_secret_default(ctx));
    if (n) hipMemsetAsync(ctx->tmp_c2.p, 0, n * 32, ctx->stream);      // the clamped secrets, on every path
    return r;
}
EXPORT int32_t c25519_mul_clamped_batch(c25519_ctx *ctx, const uint8_t *bytes, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out, uint8_t *ok) {
    return mul_batch_host(ctx, bytes, points, n, in_fmt, out_fmt, out, ok, true);
}

// ---- double base: out[i] = a[i] * A[i] + b[i] * B ----------------------------------------------------------
EXPORT int32_t c25519_double_base_batch_dev(c25519_ctx *ctx, const uint8_t *d_a, const uint8_t *d_A, const uint8_t *d_b, uint64_t n, int in_fmt, int out_fmt,
                                            uint8_t *d_out, uint8_t *d_ok) {
    HIPCHK(hipSetDevice(ctx->device));
    if (out_fmt != C25519_FMT_EDWARDS_Y && out_fmt != C25519_FMT_RAW160) { ctx->err = "double_base: out_fmt must be 0 or 2"; return -(int32_t)hipErrorInvalidValue; }
    if (n == 0) return C25519_OK;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_e, 2 * n * 160 + n + 256))) return r;
    uint32_t *P40 = (uint32_t *)ctx->tmp_e.p, *Q40 = P40 + n * 40;
    uint8_t *okbuf = d_ok ? d_ok : (uint8_t *)ctx->tmp_e.p + 2 * n * 160;
    hipStream_t st = ctx->stream;
    hipEvent_t *ring = ctx_ring_item(ctx);
    HIPCHK(hipEventRecord(ctx->ev0, st));
    HIPCHK(hipEventRecord(ring[0], st));
    if ((r = var_base_launch(ctx, d_a, d_A, n, in_fmt, false, false, P40, okbuf))) return r;          // a * A (vartime by contract)
    HIPCHK(hipEventRecord(ring[1], st));
    HIPCHK(launch_mul_base_p40(ctx->w, d_b, n, ctx->d_table, Q40, ctx->num_cus, st));                 // b * B (public)
    if (out_fmt == C25519_FMT_RAW160) {
        hipLaunchKernelGGL(k_p40_to_raw, dim3(dup(n, 256)), dim3(256), 0, st, P40, Q40, n, d_out);
    } else {
        if ((r = ctx_reserve(ctx, ctx->scratch, n * 128)) || (r = ctx_reserve(ctx, ctx->prefix, n * 48))) return r;
        hipLaunchKernelGGL(k_p40_add_to_p32, dim3(dup(n, 256)), dim3(256), 0, st, P40, Q40, n, (uint32_t *)ctx->scratch.p);
        HIPCHK(launch_compress_p32((const uint32_t *)ctx->scratch.p, (uint32_t *)ctx->prefix.p, n, d_out, st));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ring[2], st));
    HIPCHK(hipEventRecord(ctx->ev1, st));
    return C25519_OK;
}
EXPORT int32_t c25519_double_base_batch(c25519_ctx *ctx, const uint8_t *a, const uint8_t *A, const uint8_t *b, uint64_t n, int in_fmt, int out_fmt,
                                        uint8_t *out, uint8_t *ok) {
    HIPCHK(hipSetDevice(ctx->device));
    const size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32, osz = out_fmt == C25519_FMT_RAW160 ? 160 : 32;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, 2 * n * 32 + 16)) || (r = ctx_reserve(ctx, ctx->tmp_b, n * psz + 16)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * osz + n + 16))) return r;
    uint8_t *da = (uint8_t *)ctx->tmp_a.p, *db = da + n * 32, *dA = (uint8_t *)ctx->tmp_b.p, *dout = (uint8_t *)ctx->tmp_c.p, *dok = dout + n * osz;
    const ffi_in in[3] = {{a, da, 32}, {b, db, 32}, {A, dA, psz}};
    const ffi_out o[2] = {{out, dout, osz}, {ok, dok, 1}};
    return ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 16), in, 3, o, 2, [&](uint64_t lo, uint64_t m) -> int32_t {
        return c25519_double_base_batch_dev(ctx, da + lo * 32, dA + lo * psz, db + lo * 32, m, in_fmt, out_fmt, dout + lo * osz, dok + lo);
    });
}

// ---- per-signature verify ---------------------------------------------------------------------------------
// dom2 of Ed25519ph (RFC 8032 5.1: "SigEd25519 no Ed25519 collisions" || 0x01 || len(ctx) || ctx) into the context's small device buffer; the host
// copy lives in the context until the next call, so the asynchronous upload never reads a dead frame
static int32_t dom2_upload(c25519_ctx *ctx, const uint8_t *context, uint32_t context_len, const uint8_t **d_dom, uint32_t *dom_len) {
    if (context_len > 255) { ctx->err = "prehashed: the context must not be longer than 255 octets"; return C25519_PREHASHED_CONTEXT_LENGTH; }    // signing.rs:931-933
    if (context_len && !context) { ctx->err = "prehashed: null context"; return -(int32_t)hipErrorInvalidValue; }
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->dom, 512))) return r;
    ctx->h_dom.assign(34 + context_len, 0);
    memcpy(ctx->h_dom.data(), "SigEd25519 no Ed25519 collisions", 32);
    ctx->h_dom[32] = 1; ctx->h_dom[33] = (uint8_t)context_len;
    if (context_len) memcpy(ctx->h_dom.data() + 34, context, context_len);
    HIPCHK(hipMemcpyAsync(ctx->dom.p, ctx->h_dom.data(), ctx->h_dom.size(), hipMemcpyHostToDevice, ctx->stream));
    *d_dom = (const uint8_t *)ctx->dom.p; *dom_len = (uint32_t)ctx->h_dom.size();
    return C25519_OK;
}
// d_dom == nullptr: plain Ed25519 (messages through d_msg_off); otherwise Ed25519ph: d_msgs = n x 64 bytes of prehashes, d_msg_off unused
static int32_t verify_each_impl(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, int strict, uint8_t *d_status, const uint8_t *d_dom, uint32_t dom_len) {
    if (n == 0) return C25519_OK;
    hipStream_t st = ctx->stream;
    int32_t r;
    // tmp_f: hram 64n | k 32n | s 32n | s_ok n | a_ok n | strict n | rcheck 32n | P40 [k](-A) 160n | Q40 [s]B 160n
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t oH = carve(n * 64), oK = carve(n * 32), oS = carve(n * 32), oSo = carve(n), oAo = carve(n), oSt = carve(n), oRc = carve(n * 32), oP = carve(n * 160), oQ = carve(n * 160);
    if ((r = ctx_reserve(ctx, ctx->tmp_f, off)) || (r = ctx_reserve(ctx, ctx->scratch, n * 128)) || (r = ctx_reserve(ctx, ctx->prefix, n * 48))) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_f.p;
    uint8_t *hram = ws + oH, *kscal = ws + oK, *sscal = ws + oS, *s_ok = ws + oSo, *a_ok = ws + oAo, *sbad = ws + oSt, *rcheck = ws + oRc;
    uint32_t *P40 = (uint32_t *)(ws + oP), *Q40 = (uint32_t *)(ws + oQ);
    hipEvent_t *ring = ctx_ring_item(ctx);
    HIPCHK(hipEventRecord(ctx->ev0, st));
    HIPCHK(hipMemsetAsync(ctx->d_flag, 0, 16, st));
    if (d_dom) HIPCHK(launch_hram_dom(d_dom, dom_len, d_msgs, nullptr, n * 64, 64, d_sigs, d_pks, n, hram, (uint32_t *)ctx->d_flag, st));
    else HIPCHK(launch_hram(d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, hram, (uint32_t *)ctx->d_flag, st));
    hipLaunchKernelGGL(k_hram_reduce, dim3(dup(n, 256)), dim3(256), 0, st, hram, d_sigs, n, kscal, sscal, s_ok);
    HIPCHK(hipEventRecord(ring[0], st));
    if ((r = var_base_launch(ctx, kscal, d_pks, n, C25519_FMT_EDWARDS_Y, true, false, P40, a_ok))) return r;    // [k](-A), k public
    HIPCHK(hipEventRecord(ring[1], st));
    HIPCHK(launch_mul_base_p40(ctx->w, sscal, n, ctx->d_table, Q40, ctx->num_cus, st));                      // [s]B
    hipLaunchKernelGGL(k_p40_add_to_p32, dim3(dup(n, 256)), dim3(256), 0, st, P40, Q40, n, (uint32_t *)ctx->scratch.p);
    HIPCHK(launch_compress_p32((const uint32_t *)ctx->scratch.p, (uint32_t *)ctx->prefix.p, n, rcheck, st));
    if (strict) hipLaunchKernelGGL(k_strict_checks, dim3(dup(n, 256)), dim3(256), 0, st, d_sigs, d_pks, n, sbad);
    hipLaunchKernelGGL(k_verdict, dim3(dup(n, 256)), dim3(256), 0, st, d_sigs, rcheck, a_ok, s_ok, strict ? sbad : (const uint8_t *)nullptr, n, d_status);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ring[2], st));
    HIPCHK(hipEventRecord(ctx->ev1, st));
    return C25519_OK;
}
EXPORT int32_t ed25519_verify_each_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                       const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, int strict, uint8_t *d_status) {
    HIPCHK(hipSetDevice(ctx->device));
    return verify_each_impl(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, strict, d_status, nullptr, 0);
}
// Ed25519ph / Ed25519ctx, per signature (verifying.rs:230-257 raw_verify_prehashed / :284 verify_prehashed; strict: :424-461): d_prehashes = n x 64
// bytes (SHA-512 of each message: the reference takes the digest state and finalises it), context = HOST pointer, at most 255 bytes, one for the batch
EXPORT int32_t ed25519_verify_each_prehashed_dev(c25519_ctx *ctx, const uint8_t *d_prehashes, const uint8_t *context, uint32_t context_len,
                                                 const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, int strict, uint8_t *d_status) {
    HIPCHK(hipSetDevice(ctx->device));
    const uint8_t *d_dom = nullptr; uint32_t dom_len = 0;
    int32_t r = dom2_upload(ctx, context, context_len, &d_dom, &dom_len);
    if (r) return r;
    return verify_each_impl(ctx, d_prehashes, nullptr, n * 64, d_sigs, d_pks, n, strict, d_status, d_dom, dom_len);
}
EXPORT int32_t ed25519_verify_each_prehashed(c25519_ctx *ctx, const uint8_t *prehashes, const uint8_t *context, uint32_t context_len, const uint8_t *sigs, const uint8_t *pks,
                                             uint64_t n, int strict, uint8_t *status) {
    HIPCHK(hipSetDevice(ctx->device));
    if (context_len > 255) { ctx->err = "prehashed: the context must not be longer than 255 octets"; return C25519_PREHASHED_CONTEXT_LENGTH; }
    if (n == 0) return C25519_OK;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, n * 64 + 64)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 64 + n * 32 + n + 64))) return r;
    uint8_t *dph = (uint8_t *)ctx->tmp_a.p, *dsig = (uint8_t *)ctx->tmp_c.p, *dpk = dsig + n * 64, *dst = dpk + n * 32;
    const ffi_in in[3] = {{prehashes, dph, 64}, {sigs, dsig, 64}, {pks, dpk, 32}};
    const ffi_out o = {status, dst, 1};
    return ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 16), in, 3, &o, 1, [&](uint64_t lo, uint64_t m) -> int32_t {
        return ed25519_verify_each_prehashed_dev(ctx, dph + lo * 64, context, context_len, dsig + lo * 64, dpk + lo * 32, m, strict, dst + lo);
    });
}
EXPORT int32_t ed25519_verify_each(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks,
                                   uint64_t n, int strict, uint8_t *status) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;
    for (uint64_t i = 0; i < n; i++) if (msg_off[i] > msg_off[i + 1]) { ctx->err = "verify_each: msg_off is not monotone"; return -(int32_t)hipErrorInvalidValue; }
    const uint64_t mlen = msg_off[n];
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, mlen + 64)) || (r = ctx_reserve(ctx, ctx->tmp_b, (n + 1) * 8)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 64 + n * 32 + n + 64))) return r;
    uint8_t *dmsg = (uint8_t *)ctx->tmp_a.p, *dsig = (uint8_t *)ctx->tmp_c.p, *dpk = dsig + n * 64, *dst = dpk + n * 32;
    uint64_t *doff = (uint64_t *)ctx->tmp_b.p;
    // the message blob and the offsets go up whole (the kernels index the blob through the offsets); signatures and keys in chunks
    if ((r = ffi_begin(ctx))) return r;
    ffi_guard guard(ctx);                                 // an early exit below still drains the copy stream
    if (mlen) HIPCHK(hipMemcpyAsync(dmsg, msgs, mlen, hipMemcpyHostToDevice, ctx->s_h2d));
    HIPCHK(hipMemcpyAsync(doff, msg_off, (n + 1) * 8, hipMemcpyHostToDevice, ctx->s_h2d));
    const ffi_in in[2] = {{sigs, dsig, 64}, {pks, dpk, 32}};
    const ffi_out o = {status, dst, 1};
    guard.dismiss();                                      // ffi_pipeline calls ffi_end on every path
    return ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 16), in, 2, &o, 1, [&](uint64_t lo, uint64_t m) -> int32_t {
        return ed25519_verify_each_dev(ctx, dmsg, doff + lo, mlen, dsig + lo * 64, dpk + lo * 32, m, strict, dst + lo);
    }, true, mlen + (n + 1) * 8);
}

// ---- key generation / signing -------------------------------------------------------------------------------
EXPORT int32_t ed25519_keygen_batch_dev(c25519_ctx *ctx, const uint8_t *d_seeds, uint64_t n, uint8_t *d_pks) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_f, n * 64 + 256))) return r;
    uint8_t *a = (uint8_t *)ctx->tmp_f.p, *prefix = a + n * 32;
    hipLaunchKernelGGL(k_expand_seed, dim3(dup(n, 256)), dim3(256), 0, ctx->stream, d_seeds, n, a, prefix);
    r = mul_base_impl(ctx, a, n, C25519_FMT_EDWARDS_Y, d_pks, ctx_secret_default(ctx));       // secret scalar: constant-time tables
    hipError_t e = hipMemsetAsync(a, 0, n * 64, ctx->stream);   // wipe the expanded secrets, on every path
    if (r) return r;
    HIPCHK(e);
    HIPCHK(hipGetLastError());
    return C25519_OK;
}
// d_dom != nullptr: Ed25519ph (signing.rs:917-976) -- d_msgs = n x 64 bytes of prehashes, dom2 in front of both hashes
static int32_t sign_body(c25519_ctx *ctx, const uint8_t *d_seeds, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len, uint64_t n,
                         uint8_t *d_pks, uint8_t *d_sigs, uint8_t *a, uint8_t *prefix, uint8_t *rscal, uint8_t *Renc, uint8_t *hram,
                         const uint8_t *d_dom = nullptr, uint32_t dom_len = 0) {
    hipStream_t st = ctx->stream;
    const bool secret = ctx_secret_default(ctx);
    int32_t r;
    // A = a*B and R = r*B (the nonce is as secret as the key) in ONE fixed-base launch over the 2n scalars a || r: the nonce
    // r = H(prefix || M) does not depend on A, and a batch of 2^16 signatures is two launches' worth of latency otherwise
    // (2 x (0.21 + 0.10) ms of a 0.83 ms call, profiles/r03_sign_keygen_2p16.txt).  rscal = a + 32 n, Renc = AR + 32 n.
    hipLaunchKernelGGL(k_expand_seed, dim3(dup(n, 256)), dim3(256), 0, st, d_seeds, n, a, prefix);
    if (d_dom) hipLaunchKernelGGL(k_sign_nonce_dom, dim3(dup(n, 256)), dim3(256), 0, st, d_dom, dom_len, prefix, d_msgs, n, rscal);
    else hipLaunchKernelGGL(k_sign_nonce, dim3(dup(n, 256)), dim3(256), 0, st, prefix, d_msgs, d_msg_off, msgs_len, n, rscal);
    uint8_t *AR = Renc - n * 32;
    if ((r = mul_base_impl(ctx, a, 2 * n, C25519_FMT_EDWARDS_Y, AR, secret))) return r;
    HIPCHK(hipMemcpyAsync(d_pks, AR, n * 32, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_place_R, dim3(dup(n, 256)), dim3(256), 0, st, Renc, n, d_sigs);
    HIPCHK(hipMemsetAsync(ctx->d_flag, 0, 16, st));
    if (d_dom) HIPCHK(launch_hram_dom(d_dom, dom_len, d_msgs, nullptr, n * 64, 64, d_sigs, d_pks, n, hram, (uint32_t *)ctx->d_flag, st));      // k = H(dom2||R||A||PH(M))
    else HIPCHK(launch_hram(d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, hram, (uint32_t *)ctx->d_flag, st));          // k = H(R||A||M)
    hipLaunchKernelGGL(k_sign_finish, dim3(dup(n, 256)), dim3(256), 0, st, hram, a, rscal, Renc, n, d_sigs);
    HIPCHK(hipGetLastError());
    return C25519_OK;
}
static int32_t sign_batch_dev_impl(c25519_ctx *ctx, const uint8_t *d_seeds, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                   uint64_t n, uint8_t *d_pks, uint8_t *d_sigs, const uint8_t *d_dom, uint32_t dom_len) {
    if (n == 0) return C25519_OK;
    int32_t r;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    // a || r contiguous (one fixed-base launch over both, sign_body), then the prefixes; then A || R encodings, then the hashes
    size_t oA = carve(2 * n * 32), oPre = carve(n * 32), oAR = carve(2 * n * 32), oH = carve(n * 64);
    if ((r = ctx_reserve(ctx, ctx->tmp_f, off))) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_f.p;
    r = sign_body(ctx, d_seeds, d_msgs, d_msg_off, msgs_len, n, d_pks, d_sigs, ws + oA, ws + oPre, ws + oA + n * 32, ws + oAR + n * 32, ws + oH, d_dom, dom_len);
    hipError_t e = hipMemsetAsync(ws, 0, oAR, ctx->stream);     // wipe secret scalars / nonces / prefixes on EVERY exit path
    if (r) return r;
    HIPCHK(e);
    uint32_t fl[4] = {0, 0, 0, 0};                              // bad message offsets (k_hram) -> error, like verify_batch
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(fl, ctx->d_flag, 16, hipMemcpyDeviceToHost));      // (blocking: no copy is ever pending into this frame)
    if (fl[1]) { ctx->err = "sign_batch: msg_off is not monotone or runs past msgs_len"; return -(int32_t)hipErrorInvalidValue; }
    return C25519_OK;
}
EXPORT int32_t ed25519_sign_batch_dev(c25519_ctx *ctx, const uint8_t *d_seeds, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                      uint64_t n, uint8_t *d_pks, uint8_t *d_sigs) {
    HIPCHK(hipSetDevice(ctx->device));
    return sign_batch_dev_impl(ctx, d_seeds, d_msgs, d_msg_off, msgs_len, n, d_pks, d_sigs, nullptr, 0);
}
// Ed25519ph signing (signing.rs:312 sign_prehashed / :917 raw_sign_prehashed): d_prehashes = n x 64 bytes, context = HOST pointer (<= 255 bytes,
// C25519_PREHASHED_CONTEXT_LENGTH otherwise: InternalError::PrehashedContextLength), one context for the batch
EXPORT int32_t ed25519_sign_batch_prehashed_dev(c25519_ctx *ctx, const uint8_t *d_seeds, const uint8_t *d_prehashes, const uint8_t *context, uint32_t context_len,
                                                uint64_t n, uint8_t *d_pks, uint8_t *d_sigs) {
    HIPCHK(hipSetDevice(ctx->device));
    const uint8_t *d_dom = nullptr; uint32_t dom_len = 0;
    int32_t r = dom2_upload(ctx, context, context_len, &d_dom, &dom_len);
    if (r) return r;
    return sign_batch_dev_impl(ctx, d_seeds, d_prehashes, nullptr, n * 64, n, d_pks, d_sigs, d_dom, dom_len);
}
EXPORT int32_t ed25519_sign_batch_prehashed(c25519_ctx *ctx, const uint8_t *seeds, const uint8_t *prehashes, const uint8_t *context, uint32_t context_len, uint64_t n,
                                            uint8_t *pks, uint8_t *sigs) {
    HIPCHK(hipSetDevice(ctx->device));
    if (context_len > 255) { ctx->err = "prehashed: the context must not be longer than 255 octets"; return C25519_PREHASHED_CONTEXT_LENGTH; }
    if (n == 0) return C25519_OK;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, n * 64 + 64)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 128 + 64))) return r;
    uint8_t *dph = (uint8_t *)ctx->tmp_a.p, *dseed = (uint8_t *)ctx->tmp_c.p, *dpk = dseed + n * 32, *dsig = dpk + n * 32;
    stream_wipe wipe(ctx->stream);
    wipe.add(dseed, n * 32);                              // the staged secret keys, on every path
    const ffi_in in[2] = {{seeds, dseed, 32}, {prehashes, dph, 64}};
    const ffi_out o[2] = {{pks, dpk, 32}, {sigs, dsig, 64}};
    // one chunk: the signing body reads its error flag back (a synchronisation per call)
    return ffi_pipeline(ctx, n, n, in, 2, o, 2, [&](uint64_t lo, uint64_t m) -> int32_t {
        return ed25519_sign_batch_prehashed_dev(ctx, dseed + lo * 32, dph + lo * 64, context, context_len, m, dpk + lo * 32, dsig + lo * 64);
    });
}
EXPORT int32_t ed25519_sign_batch(c25519_ctx *ctx, const uint8_t *seeds, const uint8_t *msgs, const uint64_t *msg_off, uint64_t n, uint8_t *pks, uint8_t *sigs) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;
    for (uint64_t i = 0; i < n; i++) if (msg_off[i] > msg_off[i + 1]) { ctx->err = "sign_batch: msg_off is not monotone"; return -(int32_t)hipErrorInvalidValue; }
    const uint64_t mlen = msg_off[n];
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, mlen + 64)) || (r = ctx_reserve(ctx, ctx->tmp_b, (n + 1) * 8)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 128 + 64))) return r;
    uint8_t *dmsg = (uint8_t *)ctx->tmp_a.p, *dseed = (uint8_t *)ctx->tmp_c.p, *dpk = dseed + n * 32, *dsig = dpk + n * 32;
    uint64_t *doff = (uint64_t *)ctx->tmp_b.p;
    stream_wipe wipe(ctx->stream);
    wipe.add(dseed, n * 32);                              // the staged secret keys, on every path
    if ((r = ffi_begin(ctx))) return r;
    ffi_guard guard(ctx);                                 // destroyed before `wipe`: the memset of the staged secrets follows completed copies
    if (mlen) HIPCHK(hipMemcpyAsync(dmsg, msgs, mlen, hipMemcpyHostToDevice, ctx->s_h2d));
    HIPCHK(hipMemcpyAsync(doff, msg_off, (n + 1) * 8, hipMemcpyHostToDevice, ctx->s_h2d));
    const ffi_in in = {seeds, dseed, 32};
    const ffi_out o[2] = {{pks, dpk, 32}, {sigs, dsig, 64}};
    guard.dismiss();
    // one chunk: ed25519_sign_batch_dev reads its error flag back (a synchronisation per call)
    return ffi_pipeline(ctx, n, n, &in, 1, o, 2, [&](uint64_t lo, uint64_t m) -> int32_t {
        return ed25519_sign_batch_dev(ctx, dseed + lo * 32, dmsg, doff + lo, mlen, m, dpk + lo * 32, dsig + lo * 64);
    }, true, mlen + (n + 1) * 8);
}
