// Callers and batch conversions either side of the MSM:
//
//   c25519_precomp_*        VartimePrecomputedStraus (backend.rs:100-192 ->
//                           scalar_mul/precomputed_straus.rs:29-127; edwards.rs:1037-1076): per static point the
//                           multiples 2^(c k) P for every window k stay resident in HBM as affine Niels records, so a
//                           call is ONE bucket accumulation + ONE bucket reduction over all digits of all scalars (no
//                           point preparation, no Horner fold); dynamic terms go through the ordinary MSM and are added
//   c25519_msm_consttime    MultiscalarMul::multiscalar_mul (straus.rs:103-144; edwards.rs:966-1000): a
//                           regular (input-independent) schedule = one radix-16 variable-base ladder per
//                           term (k_var_base) followed by a tree sum
//   c25519_double_and_compress_batch   RistrettoPoint::double_and_compress_batch (ristretto.rs:564-648)
//   c25519_scalar_invert_batch         Scalar::invert_batch_alloc (scalar.rs:802-856)
#include <hip/hip_runtime.h>
#include <string.h>
#include <thread>
#include <vector>
#include "../../include/c25519_hip.h"
#include "devio.h"
#include "sc_sha.h"
#include "sc28.h"
#include "kernels.h"
#include "ctx.h"
#include "msm_internal.h"
#include "ffi.h"
#include <stdexcept>

using namespace c25519;
#define EXPORT extern "C" __attribute__((visibility("default")))
#define HIPCHK(call)                                                \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return c25519_fail(ctx, _e, #call);   \
    } while (0)

static inline unsigned dup64(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }

struct c25519_precomp { uint32_t *d_table; uint64_t n; c25519::msm_merged m; };

namespace c25519 {

// ---- tree sum of P40 points: each wave folds 64*K inputs (K strided per lane, then a shuffle tree) ------
__global__ void __launch_bounds__(64) k_sum_p40(const u32 *__restrict__ in, u64 m, int K, u32 *__restrict__ out) {
    u64 base = (u64)blockIdx.x * 64 * K;
    ge_p3 acc = ge_identity();
#pragma unroll 1
    for (int j = 0; j < K; j++) {
        u64 idx = base + (u64)j * 64 + threadIdx.x;
        if (idx < m) acc = ge_add(acc, p40_load(in, idx));
    }
#pragma unroll 1
    for (int off = 32; off > 0; off >>= 1) {
        ge_p3 o;
        for (int i = 0; i < 10; i++) {
            o.X.v[i] = __shfl_down(acc.X.v[i], off, 64); o.Y.v[i] = __shfl_down(acc.Y.v[i], off, 64);
            o.Z.v[i] = __shfl_down(acc.Z.v[i], off, 64); o.T.v[i] = __shfl_down(acc.T.v[i], off, 64);
        }
        acc = ge_add(acc, o);
    }
    if (threadIdx.x == 0) p40_store(out, blockIdx.x, acc);
}

// ---- RistrettoPoint::double_and_compress_batch, ristretto.rs:564-648 ---------------------------------------
// state per point (6 field elements) is recomputed in the second pass instead of stored: e, f, g, h, eg, fh
struct dc_state { feT e, f, g, h, eg, fh; };
__device__ __forceinline__ dc_state dc_state_of(const ge_p3 &P) {
    feT XX = fe_sq(P.X), YY = fe_sq(P.Y), ZZ = fe_sq(P.Z);
    feT dTT = fe_mul(fe_sq(P.T), fe_d());
    dc_state s;
    s.e = fe_mul(P.X, fe_add(P.Y, P.Y));          // 2XY
    s.f = fe_carry(fe_add(ZZ, dTT));              // Z^2 + dT^2
    s.g = fe_carry(fe_add(YY, XX));               // Y^2 + X^2   (a = -1)
    s.h = fe_carry(fe_sub(ZZ, dTT));              // Z^2 - dT^2
    s.eg = fe_mul(s.e, s.g);
    s.fh = fe_mul(s.f, s.h);
    return s;
}
template <int CH>
__global__ void __launch_bounds__(256) k_double_compress(const uint8_t *__restrict__ in_raw, u32 *__restrict__ prefix, u64 n, uint8_t *__restrict__ out) {
    const u64 T = (u64)gridDim.x * blockDim.x, t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    feT acc = fe_one();
#pragma unroll 1
    for (int j = 0; j < CH; j++) {
        u64 idx = t + (u64)j * T;
        if (idx >= n) break;
        dc_state s = dc_state_of(raw160_load(in_raw, idx));
        feT efgh = fe_mul(s.eg, s.fh);
        fe48_store(prefix, idx, acc);
        acc = fe_select(fe_mul(acc, efgh), acc, fe_is_zero(efgh));      // zeros skipped (field.rs:225-273)
    }
    feT inv = fe_invert(acc);
#pragma unroll 1
    for (int j = CH - 1; j >= 0; j--) {
        u64 idx = t + (u64)j * T;
        if (idx >= n) continue;
        dc_state s = dc_state_of(raw160_load(in_raw, idx));
        feT efgh = fe_mul(s.eg, s.fh);
        bool z = fe_is_zero(efgh);
        feT einv = fe_select(fe_mul(inv, fe48_load(prefix, idx)), efgh, z);   // invert_batch leaves a zero input unchanged
        inv = fe_select(fe_mul(inv, efgh), inv, z);
        feT Zinv = fe_mul(s.eg, einv), Tinv = fe_mul(s.fh, einv);
        bool neg1 = fe_is_negative(fe_mul(s.eg, Zinv)) != 0;
        feT minus_e = fe_carry(fe_neg(s.e));
        feT f_sqrta = fe_mul(s.f, fe_sqrtm1());
        feT e = fe_select(s.e, s.g, neg1), g = fe_select(s.g, minus_e, neg1), h = fe_select(s.h, f_sqrta, neg1);
        feT magic = fe_select(fe_invsqrt_a_minus_d(), fe_sqrtm1(), neg1);
        bool neg2 = fe_is_negative(fe_mul(fe_mul(h, e), Zinv)) != 0;
        g = fe_cneg(g, neg2);
        feT sv = fe_mul(fe_sub(h, g), fe_mul(magic, fe_mul(g, Tinv)));
        sv = fe_cneg(sv, fe_is_negative(sv) != 0);
        u32 w[8];
        fe_to_words(sv, w);
        store8(out, idx, w);
    }
}

// ---- Scalar::invert_batch_alloc, scalar.rs:802-856 (all inputs must be non-zero) -----------------------------
// x^(l-2) by square-and-multiply over the bits of l - 2, on 28-bit limbs (sc28.h: no Montgomery form on the device)
__device__ __forceinline__ sc28 sc28_invert(const sc28 &x) {
    // l - 2 = 2^252 + 27742317777372353535851937790883648491, little-endian 64-bit words
    const u64 e[4] = {0x5812631a5cf5d3ebull, 0x14def9dea2f79cd6ull, 0x0000000000000000ull, 0x1000000000000000ull};
    sc28 r = sc28_zero();
    r.v[0] = 1;
#pragma unroll 1
    for (int i = 252; i >= 0; i--) {
        r = sc28_mul(r.v, r.v);
        if ((e[i >> 6] >> (i & 63)) & 1) r = sc28_mul(r.v, x.v);
    }
    return r;
}
template <int CH>
__global__ void __launch_bounds__(256) k_scalar_invert(uint8_t *__restrict__ io, u64 n, u32 *__restrict__ prefix, u32 *__restrict__ partial_inv) {
    const u64 T = (u64)gridDim.x * blockDim.x, t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    sc28 acc = sc28_zero();
    acc.v[0] = 1;
#pragma unroll 1
    for (int j = 0; j < CH; j++) {
        u64 idx = t + (u64)j * T;
        if (idx >= n) break;
        u32 w[8];
        load8(io, idx, w);
        for (int q = 0; q < 10; q++) prefix[idx * 10 + q] = acc.v[q];
        acc = sc28_mul(acc.v, sc28_from_words(w).v);
    }
    acc = sc28_invert(acc);
    // product of all inverses of this lane's chunk (the reference returns the product over the whole batch)
    for (int q = 0; q < 10; q++) partial_inv[t * 10 + q] = acc.v[q];
#pragma unroll 1
    for (int j = CH - 1; j >= 0; j--) {
        u64 idx = t + (u64)j * T;
        if (idx >= n) continue;
        u32 w[8];
        load8(io, idx, w);
        sc28 pre;
        for (int q = 0; q < 10; q++) pre.v[q] = prefix[idx * 10 + q];
        const sc28 inv = sc28_mul(acc.v, pre.v);                       // 1 / x
        acc = sc28_mul(acc.v, sc28_from_words(w).v);
        sc28_to_words(inv, w);
        store8(io, idx, w);
    }
}

}  // namespace c25519

// ---- precomputed static MSM ------------------------------------------------------------------------------------
// create: the per-window multiples 2^(c k) P_i of every static point, normalised, resident in HBM (msm_internal.h msm_merged)
EXPORT c25519_precomp *c25519_precomp_create(c25519_ctx *ctx, const uint8_t *static_points, uint64_t n, int in_fmt) {
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    if (in_fmt < 0 || in_fmt > 2) { ctx->err = "precomp_create: bad in_fmt"; return nullptr; }
    size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    c25519_precomp *p = new c25519_precomp{nullptr, n, {}};
    msm_merged_layout(n, p->m);
    if (hipMalloc((void **)&p->d_table, (size_t)(n ? n : 1) * p->m.K * PTS_BYTES) != hipSuccess) { ctx->err = "precomp_create: hipMalloc failed"; delete p; return nullptr; }
    if (n == 0) return p;
    uint32_t *bad = (uint32_t *)ctx->d_flag;
    uint32_t hb[2] = {0, 0};
    bool ok = ctx_reserve(ctx, ctx->tmp_b, n * psz + 16) == 0 && hipMemsetAsync(bad, 0, 16, ctx->stream) == hipSuccess &&
              hipMemcpyAsync(ctx->tmp_b.p, static_points, n * psz, hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
              msm_merged_build(ctx, (const uint8_t *)ctx->tmp_b.p, n, in_fmt, p->m, p->d_table, bad) == C25519_OK &&
              hipStreamSynchronize(ctx->stream) == hipSuccess && hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && hb[0] != 0) { ctx->err = "precomp_create: a static point does not decode"; ok = false; }
    else if (!ok && ctx->err.empty()) ctx->err = "precomp_create: a HIP call failed";
    if (!ok) { hipFree(p->d_table); delete p; return nullptr; }
    return p;
}
EXPORT void c25519_precomp_destroy(c25519_ctx *ctx, c25519_precomp *p) {
    if (!p) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipFree(p->d_table);
    delete p;
}
EXPORT uint64_t c25519_precomp_len(const c25519_precomp *p) { return p ? p->n : 0; }
EXPORT int32_t c25519_precomp_msm_vartime(c25519_ctx *ctx, const c25519_precomp *p, const uint8_t *static_scalars, uint64_t n_static_scalars,
                                          const uint8_t *dyn_scalars, const uint8_t *dyn_points, uint64_t n_dyn, int in_fmt, int out_fmt, uint8_t *out) {
    HIPCHK(hipSetDevice(ctx->device));
    if (out_fmt < 0 || out_fmt > 2) { ctx->err = "precomp_msm: bad out_fmt"; return -(int32_t)hipErrorInvalidValue; }
    if (n_static_scalars > p->n) { ctx->err = "precomp_msm: more static scalars than static points (precomputed_straus.rs:86)"; return -(int32_t)hipErrorInvalidValue; }
    const uint64_t ns = n_static_scalars;
    ge_p3 R = ge_identity();
    size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    int32_t r;
    hipStream_t st = ctx->stream;
    HIPCHK(hipEventRecord(ctx->ev0, st));
    if (ns) {          // static part: every digit of every scalar is a term of ONE bucket problem over the resident table
        if ((r = ctx_reserve(ctx, ctx->tmp_a, ns * 32 + 16))) return r;
        HIPCHK(hipMemcpyAsync(ctx->tmp_a.p, static_scalars, ns * 32, hipMemcpyHostToDevice, st));
        if ((r = msm_merged_core(ctx, (const uint8_t *)ctx->tmp_a.p, ns, p->m, p->d_table, R))) return r;
    }
    if (n_dyn) {       // dynamic part: the ordinary pipeline (prepare, sort, accumulate, reduce, fold)
        if ((r = ctx_reserve(ctx, ctx->tmp_a, n_dyn * 32 + 16)) || (r = ctx_reserve(ctx, ctx->tmp_b, n_dyn * psz + 16)) || (r = ctx_reserve(ctx, ctx->tmp_e, n_dyn * PTS_BYTES + 256))) return r;
        uint32_t *d_pts = (uint32_t *)ctx->tmp_e.p, *bad = (uint32_t *)ctx->d_flag;
        HIPCHK(hipMemsetAsync(bad, 0, 16, st));
        HIPCHK(hipMemcpyAsync(ctx->tmp_a.p, dyn_scalars, n_dyn * 32, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(ctx->tmp_b.p, dyn_points, n_dyn * psz, hipMemcpyHostToDevice, st));
        if ((r = prep_points(ctx, (const uint8_t *)ctx->tmp_b.p, n_dyn, in_fmt, d_pts, 0, bad))) return r;
        ge_p3 Rd;
        if ((r = msm_core(ctx, (const uint8_t *)ctx->tmp_a.p, n_dyn, d_pts, Rd))) return r;      // synchronises the stream
        uint32_t hb = 0;
        HIPCHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));                                  // (a blocking copy: nothing is pending into this frame on any path)
        if (hb) return C25519_NONE;
        R = ns ? ge_add(R, Rd) : Rd;
    }
    HIPCHK(hipEventRecord(ctx->ev1, st));
    host_encode(R, out_fmt, out);
    return C25519_OK;
}

// ---- regular-schedule multiscalar multiplication -------------------------------------------------------------------
EXPORT int32_t c25519_msm_consttime(c25519_ctx *ctx, const uint8_t *scalars, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out) {
    HIPCHK(hipSetDevice(ctx->device));
    if (out_fmt < 0 || out_fmt > 2) { ctx->err = "msm_consttime: bad out_fmt"; return -(int32_t)hipErrorInvalidValue; }
    ge_p3 R = ge_identity();
    if (n == 0) { host_encode(R, out_fmt, out); return C25519_OK; }
    size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, n * 32 + 16)) || (r = ctx_reserve(ctx, ctx->tmp_b, n * psz + 16)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 160 + n + 256))) return r;
    hipStream_t st = ctx->stream;
    HIPCHK(hipMemcpyAsync(ctx->tmp_a.p, scalars, n * 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->tmp_b.p, points, n * psz, hipMemcpyHostToDevice, st));
    uint8_t *d_raw = (uint8_t *)ctx->tmp_c.p, *d_ok = d_raw + n * 160;
    // always the constant-address table scan, whatever the context's flags: that is what this entry point is for
    r = mul_batch_impl(ctx, (const uint8_t *)ctx->tmp_a.p, (const uint8_t *)ctx->tmp_b.p, n, in_fmt, C25519_FMT_RAW160, d_raw, d_ok, true);
    hipMemsetAsync(ctx->tmp_a.p, 0, n * 32, st);          // the staged secret scalars, on every path
    if (r) return r;
    // c25519_mul_batch_dev left the products as P40 records in tmp_e: fold them
    const int K = 4;
    uint32_t *cur = (uint32_t *)ctx->tmp_e.p;
    uint64_t m = n;
    if ((r = ctx_reserve(ctx, ctx->tmp_f, (size_t)(n / (64 * K) + 2) * 160 * 2 + 512))) return r;
    uint32_t *bufA = (uint32_t *)ctx->tmp_f.p, *bufB = bufA + (size_t)(n / (64 * K) + 2) * 40;
    while (m > 64) {
        uint64_t mo = (m + 64 * K - 1) / (64 * K);
        uint32_t *dst = (cur == bufA) ? bufB : bufA;
        hipLaunchKernelGGL(k_sum_p40, dim3((unsigned)mo), dim3(64), 0, st, cur, m, K, dst);
        cur = dst; m = mo;
    }
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> tail(m * 40);
    std::vector<uint8_t> okh(n);
    HIPCHK(hipMemcpyAsync(tail.data(), cur, m * 160, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(okh.data(), d_ok, n, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (uint64_t i = 0; i < n; i++) if (!okh[i]) return C25519_NONE;
    for (uint64_t i = 0; i < m; i++) {
        ge_p3 q;
        for (int j = 0; j < 10; j++) { q.X.v[j] = tail[i * 40 + j]; q.Y.v[j] = tail[i * 40 + 10 + j]; q.Z.v[j] = tail[i * 40 + 20 + j]; q.T.v[j] = tail[i * 40 + 30 + j]; }
        R = ge_add(R, q);
    }
    host_encode(R, out_fmt, out);
    return C25519_OK;
}

// ---- Ristretto double-and-compress -------------------------------------------------------------------------------------
EXPORT int32_t c25519_double_and_compress_batch_dev(c25519_ctx *ctx, const uint8_t *d_in, uint64_t n, uint8_t *d_out) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->prefix, n * 48))) return r;
    constexpr int CH = 16;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(k_double_compress<CH>, dim3(dup64((n + CH - 1) / CH, 256)), dim3(256), 0, ctx->stream, d_in, (uint32_t *)ctx->prefix.p, n, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    return C25519_OK;
}
EXPORT int32_t c25519_double_and_compress_batch(c25519_ctx *ctx, const uint8_t *in, uint64_t n, uint8_t *out) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_a, n * 160)) || (r = ctx_reserve(ctx, ctx->tmp_b, n * 32))) return r;
    uint8_t *d_in = (uint8_t *)ctx->tmp_a.p, *d_out = (uint8_t *)ctx->tmp_b.p;
    const ffi_in i1 = {in, d_in, 160};
    const ffi_out o = {out, d_out, 32};
    return ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 17), &i1, 1, &o, 1,
                        [&](uint64_t lo, uint64_t m) -> int32_t { return c25519_double_and_compress_batch_dev(ctx, d_in + lo * 160, m, d_out + lo * 32); });
}

// ---- Scalar::invert_batch ------------------------------------------------------------------------------------------------
// io: n x 32 canonical non-zero scalars, inverted in place; prod_inv (32 bytes, may be NULL): product of all inverses
EXPORT int32_t c25519_scalar_invert_batch(c25519_ctx *ctx, uint8_t *io, uint64_t n, uint8_t *prod_inv) {
    HIPCHK(hipSetDevice(ctx->device));
    sc52 prod = sc_zero(); prod.v[0] = 1;
    if (n) {
        try {
            constexpr int CH = 16;
            const uint64_t lanes = (n + CH - 1) / CH;
            const unsigned grid = dup64(lanes, 256);
            const uint64_t T = (uint64_t)grid * 256;
            int32_t r;
            if ((r = ctx_reserve(ctx, ctx->tmp_a, n * 32 + 16)) || (r = ctx_reserve(ctx, ctx->tmp_b, n * 40 + 64)) || (r = ctx_reserve(ctx, ctx->tmp_c, T * 40 + 64))) return r;
            HIPCHK(hipMemcpyAsync(ctx->tmp_a.p, io, n * 32, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->tmp_c.p, 0, T * 40, ctx->stream));
            hipLaunchKernelGGL(k_scalar_invert<CH>, dim3(grid), dim3(256), 0, ctx->stream, (uint8_t *)ctx->tmp_a.p, n, (uint32_t *)ctx->tmp_b.p, (uint32_t *)ctx->tmp_c.p);
            HIPCHK(hipGetLastError());
            std::vector<uint32_t> parts(T * 10);
            HIPCHK(hipMemcpyAsync(io, ctx->tmp_a.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(parts.data(), ctx->tmp_c.p, T * 40, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipMemsetAsync(ctx->tmp_a.p, 0, n * 32, ctx->stream));   // zeroize (scalar.rs:852)
            HIPCHK(hipMemsetAsync(ctx->tmp_b.p, 0, n * 40, ctx->stream));
            const uint64_t active = n < T ? n : T;     // lane t owns elements t, t+T, ...: every lane below n is active
            for (uint64_t t = 0; t < active; t++) {    // the product over the lanes, on the host (5 x 52 Montgomery form, sc_sha.h)
                sc28 q28;
                for (int j = 0; j < 10; j++) q28.v[j] = parts[t * 10 + j];
                u32 w[8];
                sc28_to_words(q28, w);
                prod = sc_mul(prod, sc_from_words(w));
            }
        } catch (const std::exception &e) { ctx->err = std::string("scalar_invert_batch: ") + e.what(); return -(int32_t)hipErrorOutOfMemory; }
    }
    if (prod_inv) { u32 w[8]; sc_to_words(prod, w); memcpy(prod_inv, w, 32); }
    return C25519_OK;
}

// ---- one process, several GPUs -------------------------------------------------------------------------------------------
// The multi-GPU decomposition of SURVEY.md 8e for a host that drives all GPUs of a node from ONE process (a Rust
// caller without torch / RCCL): the terms (signatures) are cut into contiguous shards, one per context -- contexts on
// different devices, or several on one device -- every shard runs the full single-GPU path on its own context from its own
// host thread, and the partial sums (verdicts) are folded on the host: the exchange step is nctx x 160 bytes (4 bytes)
// over PCIe instead of an all_gather over xGMI.  Results are identical to the single-context calls by construction.
static bool ctxs_ok(c25519_ctx **ctxs, int32_t nctx) {
    if (nctx < 1 || !ctxs) return false;
    for (int32_t r = 0; r < nctx; r++) if (!ctxs[r]) return false;
    return true;
}
static inline void shard_of(uint64_t n, int nctx, int r, uint64_t &lo, uint64_t &cnt) {
    const uint64_t base = n / nctx, rem = n % nctx;
    lo = r * base + std::min<uint64_t>(r, rem); cnt = base + ((uint64_t)r < rem ? 1 : 0);
}
// run fn(r) for every context, context 0 on the calling thread; nothing thrown by a worker or by thread creation leaves this function
template <class F>
static int32_t on_every_context(c25519_ctx *ctx0, int32_t nctx, F &&fn) {
    try {
        std::vector<std::thread> th;
        std::vector<int> threw(nctx, 0);
        auto guarded = [&](int r) { try { fn(r); } catch (...) { threw[r] = 1; } };
        for (int r = 1; r < nctx; r++) th.emplace_back(guarded, r);
        guarded(0);
        for (auto &t : th) t.join();
        for (int r = 0; r < nctx; r++) if (threw[r]) { ctx0->err = "multi: a worker failed (out of memory?)"; return -(int32_t)hipErrorOutOfMemory; }
        return C25519_OK;
    } catch (const std::exception &e) {
        ctx0->err = std::string("multi: ") + e.what();     // (a std::thread that was started and not joined would terminate: emplace_back is the only thrower, before any later start)
        return -(int32_t)hipErrorOutOfMemory;
    }
}
EXPORT int32_t c25519_msm_vartime_multi(c25519_ctx **ctxs, int32_t nctx, const uint8_t *scalars, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out) {
    if (!ctxs_ok(ctxs, nctx)) return -(int32_t)hipErrorInvalidValue;
    c25519_ctx *ctx = ctxs[0];
    if (out_fmt < 0 || out_fmt > 2 || in_fmt < 0 || in_fmt > 2) { ctx->err = "msm_multi: bad format"; return -(int32_t)hipErrorInvalidValue; }
    const size_t psz = in_fmt == C25519_FMT_RAW160 ? 160 : 32;
    try {
        std::vector<uint8_t> part((size_t)nctx * 160);
        std::vector<int32_t> st(nctx, C25519_OK);
        int32_t r0 = on_every_context(ctx, nctx, [&](int r) {
            uint64_t lo, cnt;
            shard_of(n, nctx, r, lo, cnt);
            st[r] = c25519_msm_vartime(ctxs[r], scalars + lo * 32, points + lo * psz, cnt, in_fmt, C25519_FMT_RAW160, &part[(size_t)r * 160]);
        });
        if (r0) return r0;
        bool none = false;
        for (int r = 0; r < nctx; r++) {
            if (st[r] < 0) { if (r) ctx->err = ctxs[r]->err; return st[r]; }
            if (st[r] == C25519_NONE) none = true;
        }
        if (none) return C25519_NONE;
        return c25519_fold_partials(ctx, part.data(), (uint64_t)nctx, out_fmt, out);
    } catch (const std::exception &e) { ctx->err = std::string("msm_multi: ") + e.what(); return -(int32_t)hipErrorOutOfMemory; }
}
// verify_batch over several contexts.  C25519_Z_TRANSCRIPT: the reference's ONE transcript over the whole batch and its single
// equation -- every context hashes its shard, the host runs the transcript once over all hram_i / s_i, every
// context evaluates its share of the equation with its z_i and leaves a record, the records are folded into one identity
// check.  C25519_Z_DEVICE: independent shard checks, worst verdict in the reference's precedence.
EXPORT int32_t ed25519_verify_batch_multi(c25519_ctx **ctxs, int32_t nctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks,
                                          uint64_t n, uint32_t z_mode) {
    if (!ctxs_ok(ctxs, nctx)) return -(int32_t)hipErrorInvalidValue;
    c25519_ctx *ctx = ctxs[0];
    if (n == 0) return C25519_OK;
    if (z_mode > 1) { ctx->err = "verify_batch_multi: bad z_mode"; return -(int32_t)hipErrorInvalidValue; }
    for (uint64_t i = 0; i < n; i++) if (msg_off[i] > msg_off[i + 1]) { ctx->err = "verify_batch_multi: msg_off is not monotone"; return -(int32_t)hipErrorInvalidValue; }
    try {
        std::vector<int32_t> st(nctx, C25519_OK);
        if (z_mode == C25519_Z_DEVICE) {
            int32_t r0 = on_every_context(ctx, nctx, [&](int r) {
                uint64_t lo, cnt;
                shard_of(n, nctx, r, lo, cnt);
                if (cnt == 0) return;
                std::vector<uint64_t> off(cnt + 1);          // this shard's offsets, rebased to its first message
                for (uint64_t i = 0; i <= cnt; i++) off[i] = msg_off[lo + i] - msg_off[lo];
                st[r] = ed25519_verify_batch(ctxs[r], msgs + msg_off[lo], off.data(), sigs + lo * 64, pks + lo * 32, cnt, z_mode);
            });
            if (r0) return r0;
            bool seen[5] = {false, false, false, false, false};
            for (int r = 0; r < nctx; r++) {
                if (st[r] < 0) { if (r) ctx->err = ctxs[r]->err; return st[r]; }
                seen[st[r]] = true;
            }
            return seen[C25519_NONE] ? C25519_NONE : seen[C25519_SCALAR_FORMAT] ? C25519_SCALAR_FORMAT : seen[C25519_VERIFY] ? C25519_VERIFY : C25519_OK;
        }
        // ---- strict z-mode: hash per shard, ONE transcript, equation per shard, ONE identity check ------------------------------
        std::vector<uint8_t> hram(n * 64), z16(n * 16), records((size_t)nctx * C25519_PARTIAL_RECORD_BYTES);
        struct shard_dev { uint8_t *sig, *pk, *hr, *z, *rec; };
        std::vector<shard_dev> dv(nctx, shard_dev{nullptr, nullptr, nullptr, nullptr, nullptr});
        auto fail = [&](int r, hipError_t e, const char *what) { st[r] = c25519_fail(ctxs[r], e, what); };
        int32_t r0 = on_every_context(ctx, nctx, [&](int r) {            // phase 1: upload the shard, hash it, hram back to the host
            c25519_ctx *c = ctxs[r];
            uint64_t lo, cnt;
            shard_of(n, nctx, r, lo, cnt);
            hipError_t e;
            if ((e = hipSetDevice(c->device)) != hipSuccess) return fail(r, e, "hipSetDevice");
            const uint64_t mlen = msg_off[lo + cnt] - msg_off[lo];
            // tmp_c: sigs 64 cnt | pks 32 cnt | hram 64 cnt + 64 | z 16 cnt | record ;  tmp_a: messages ;  tmp_b: offsets
            size_t off = 0;
            auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
            const size_t oS = carve(cnt * 64), oK = carve(cnt * 32), oH = carve(cnt * 64 + 64), oZ = carve(cnt * 16 + 16), oR = carve(C25519_PARTIAL_RECORD_BYTES);
            if ((st[r] = ctx_reserve(c, c->tmp_c, off)) || (st[r] = ctx_reserve(c, c->tmp_a, mlen + 64)) || (st[r] = ctx_reserve(c, c->tmp_b, (cnt + 1) * 8))) return;
            uint8_t *ws = (uint8_t *)c->tmp_c.p;
            dv[r] = shard_dev{ws + oS, ws + oK, ws + oH, ws + oZ, ws + oR};
            std::vector<uint64_t> offs(cnt + 1);
            for (uint64_t i = 0; i <= cnt; i++) offs[i] = msg_off[lo + i] - msg_off[lo];
            if (mlen && (e = hipMemcpyAsync(c->tmp_a.p, msgs + msg_off[lo], mlen, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return fail(r, e, "H2D");
            if ((e = hipMemcpyAsync(c->tmp_b.p, offs.data(), (cnt + 1) * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return fail(r, e, "H2D");
            if (cnt && (e = hipMemcpyAsync(dv[r].sig, sigs + lo * 64, cnt * 64, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return fail(r, e, "H2D");
            if (cnt && (e = hipMemcpyAsync(dv[r].pk, pks + lo * 32, cnt * 32, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return fail(r, e, "H2D");
            if ((st[r] = ed25519_batch_hram_dev(c, (const uint8_t *)c->tmp_a.p, (const uint64_t *)c->tmp_b.p, mlen, dv[r].sig, dv[r].pk, cnt, dv[r].hr))) return;
            if (cnt && (e = hipMemcpyAsync(&hram[lo * 64], dv[r].hr, cnt * 64, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return fail(r, e, "D2H");
            if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return fail(r, e, "hipStreamSynchronize");       // (offs is a local buffer)
        });
        if (r0) return r0;
        for (int r = 0; r < nctx; r++) if (st[r]) { if (r) ctx->err = ctxs[r]->err; return st[r]; }
        ed25519_batch_transcript_zs(hram.data(), sigs, n, z16.data());          // batch.rs:168-222, once, over the whole batch
        r0 = on_every_context(ctx, nctx, [&](int r) {                           // phase 2: this shard's share of the equation
            c25519_ctx *c = ctxs[r];
            uint64_t lo, cnt;
            shard_of(n, nctx, r, lo, cnt);
            hipError_t e;
            if ((e = hipSetDevice(c->device)) != hipSuccess) return fail(r, e, "hipSetDevice");
            if (cnt && (e = hipMemcpyAsync(dv[r].z, &z16[lo * 16], cnt * 16, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return fail(r, e, "H2D");
            if ((st[r] = ed25519_verify_batch_record_dev(c, dv[r].sig, dv[r].pk, nullptr, dv[r].hr, dv[r].z, cnt, dv[r].rec))) return;
            if ((e = hipMemcpyAsync(&records[(size_t)r * C25519_PARTIAL_RECORD_BYTES], dv[r].rec, C25519_PARTIAL_RECORD_BYTES, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return fail(r, e, "D2H");
            if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return fail(r, e, "hipStreamSynchronize");
        });
        if (r0) return r0;
        for (int r = 0; r < nctx; r++) if (st[r]) { if (r) ctx->err = ctxs[r]->err; return st[r]; }
        return ed25519_fold_verify_records(ctx, records.data(), (uint64_t)nctx);
    } catch (const std::exception &e) { ctx->err = std::string("verify_batch_multi: ") + e.what(); return -(int32_t)hipErrorOutOfMemory; }
}
