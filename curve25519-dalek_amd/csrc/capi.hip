// Host side of libc25519hip.so: context, workspaces, table generation and the extern "C" entry
// points declared in include/c25519_hip.h.  All arithmetic on the data path runs in the gfx950
// kernels (kernels.hip, msm.hip); the only curve arithmetic executed on the host is one-time table
// generation and the O(windows) tail of an MSM, both through the very same fe26/ge26 headers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/c25519_hip.h"
#include "ge26.h"
#include "kernels.h"
#include "ctx.h"
#include "sc_sha.h"
#include "msm_internal.h"
#include "ffi.h"

using namespace c25519;

#define EXPORT extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------------------------------------
int32_t c25519_fail(c25519_ctx *ctx, hipError_t e, const char *where) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", where, hipGetErrorString(e));
    ctx->err = buf;
    return -(int32_t)e;
}
#define HIPCHK(call)                                                \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return c25519_fail(ctx, _e, #call);   \
    } while (0)

int32_t ctx_reserve(c25519_ctx *ctx, devbuf &b, size_t bytes) {
    if (b.cap >= bytes) return 0;
    if (b.p) { hipError_t e = hipFree(b.p); b.p = nullptr; b.cap = 0; if (e != hipSuccess) return c25519_fail(ctx, e, "hipFree"); }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) return c25519_fail(ctx, e, "hipMalloc(workspace)");
    b.cap = want;
    return 0;
}
int32_t ctx_host_stage(c25519_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->h_stage_cap) return C25519_OK;
    if (ctx->h_stage) { (void)hipHostFree(ctx->h_stage); ctx->h_stage = nullptr; ctx->h_stage_cap = 0; }
    const size_t cap = bytes + bytes / 8 + 4096;
    // (coherent + mapped, explicitly: the small paths let their kernels read this buffer in place -- ffi_small_upload zero_copy, verify.hip verify_batch_small_host --
    //  so what the host wrote before a launch must be what the kernel sees, whatever HIP_HOST_COHERENT says)
    hipError_t e = hipHostMalloc(&ctx->h_stage, cap, hipHostMallocCoherent | hipHostMallocMapped);
    if (e != hipSuccess) return c25519_fail(ctx, e, "hipHostMalloc(host staging)");
    ctx->h_stage_cap = cap;
    return C25519_OK;
}

// ---- fixed-base table, built by the `create` logic of edwards.rs:1131-1141 -----------------------
// entry j (1..HALF) of window i = j * 2^(W i) * P (P = B for the context's own tables), as canonical (y+x, y-x, 2dxy); entry 0 = identity.
static void build_window_table(ge_p3 base, int W, std::vector<uint32_t> &out) {
    const int NWIN = (256 + W - 1) / W, HALF = 1 << (W - 1), ENT = HALF + 1;
    std::vector<ge_p3> pts((size_t)NWIN * HALF);
    for (int i = 0; i < NWIN; i++) {
        ge_p3 acc = base;
        for (int j = 0; j < HALF; j++) {
            pts[(size_t)i * HALF + j] = acc;
            acc = ge_add(acc, base);
        }
        base = ge_mul_by_pow_2(base, W);
    }
    // batch-normalise (Montgomery's trick, field.rs:225-273)
    size_t m = pts.size();
    std::vector<feT> pre(m);
    feT acc = fe_one();
    for (size_t k = 0; k < m; k++) { pre[k] = acc; acc = fe_mul(acc, pts[k].Z); }
    feT inv = fe_invert(acc);
    out.assign((size_t)NWIN * ENT * 24, 0u);
    std::vector<feT> zinv(m);
    for (size_t k = m; k-- > 0;) { zinv[k] = fe_mul(inv, pre[k]); inv = fe_mul(inv, pts[k].Z); }
    for (int i = 0; i < NWIN; i++) {
        uint32_t *e0 = &out[((size_t)i * ENT) * 24];
        e0[0] = 1; e0[8] = 1;   // identity: y+x = 1, y-x = 1, 2dxy = 0
        for (int j = 0; j < HALF; j++) {
            size_t k = (size_t)i * HALF + j;
            feT x = fe_mul(pts[k].X, zinv[k]), y = fe_mul(pts[k].Y, zinv[k]);
            uint32_t *e = &out[((size_t)i * ENT + j + 1) * 24];
            fe_to_words(fe_add(y, x), e);
            fe_to_words(fe_sub(y, x), e + 8);
            fe_to_words(fe_mul(fe_mul(x, y), fe_d2()), e + 16);
        }
    }
}

// ---- signed comb table (k_mul_base_comb): entry idx of block m =
//      2^(5m) * ( 2^(240) + sum_{tau<8} (2*bit_tau(idx) - 1) * 2^(30 tau) ) * B ;  last entry = B.
static void build_comb_table(std::vector<uint32_t> &out) {
    const int T = 9, D = 30, V = 6, E = 5, ENT = 256;
    std::vector<ge_p3> pts((size_t)V * ENT + 1);
    ge_p3 G[T];
    G[0] = ge_basepoint();
    for (int t = 1; t < T; t++) G[t] = ge_mul_by_pow_2(G[t - 1], D);
    for (int m = 0; m < V; m++) {
        ge_p3 G2[T];
        for (int t = 0; t < T; t++) G2[t] = ge_dbl_p3(G[t]);
        ge_p3 e0 = G[T - 1];
        for (int t = 0; t < T - 1; t++) e0 = ge_add(e0, ge_neg(G[t]));          // all lower digits -1
        pts[(size_t)m * ENT] = e0;
        for (int idx = 1; idx < ENT; idx++) {
            int low = __builtin_ctz(idx);
            pts[(size_t)m * ENT + idx] = ge_add(pts[(size_t)m * ENT + (idx & (idx - 1))], G2[low]);   // flip digit `low` from -1 to +1
        }
        for (int t = 0; t < T; t++) G[t] = ge_mul_by_pow_2(G[t], E);
    }
    pts[(size_t)V * ENT] = ge_basepoint();
    size_t mtot = pts.size();
    std::vector<feT> pre(mtot);
    feT acc = fe_one();
    for (size_t k = 0; k < mtot; k++) { pre[k] = acc; acc = fe_mul(acc, pts[k].Z); }
    feT inv = fe_invert(acc);
    out.assign(mtot * 24, 0u);
    for (size_t k = mtot; k-- > 0;) {
        feT zi = fe_mul(inv, pre[k]);
        inv = fe_mul(inv, pts[k].Z);
        feT x = fe_mul(pts[k].X, zi), y = fe_mul(pts[k].Y, zi);
        uint32_t *e = &out[k * 24];
        fe_to_words(fe_add(y, x), e);
        fe_to_words(fe_sub(y, x), e + 8);
        fe_to_words(fe_mul(fe_mul(x, y), fe_d2()), e + 16);
    }
}

// ------------------------------------------------------------------------------------------------
// ---- wide-window fixed-base table (k_mul_base_wide): entry e of window j = e * 2^(C j) * B.  The table is made
// on the device by the engine itself: the host writes the scalars e * 2^(C j) mod l (repeated sc_add), the comb
// kernel multiplies them by B, and the batched normaliser of the MSM path packs them as affine Niels points.
static int32_t build_wide_table(c25519_ctx *ctx, int C) {
    const int nw = (256 + C - 1) / C, rem = 256 - C * (nw - 1);
    const uint64_t HALF = 1ull << (C - 1), ENT = HALF + 1, N = (uint64_t)(nw - 1) * ENT + (1ull << rem) + 1;
    std::vector<uint8_t> sc((size_t)N * 32);
    sc52 p2 = sc_zero(); p2.v[0] = 1;
    for (int j = 0; j < nw; j++) {
        const uint64_t cnt = (j == nw - 1) ? (1ull << rem) + 1 : ENT;
        sc52 acc = sc_zero();
        for (uint64_t e = 0; e < cnt; e++) {
            u32 wds[8];
            sc_to_words(acc, wds);
            memcpy(&sc[((size_t)j * ENT + e) * 32], wds, 32);
            acc = sc_add(acc, p2);
        }
        for (int k = 0; k < C; k++) p2 = sc_add(p2, p2);
    }
    struct tmp { void *p = nullptr; ~tmp() { if (p) hipFree(p); } } t_sc, t_raw, t_wide;     // freed on every path
    HIPCHK(hipMalloc(&t_sc.p, N * 32));
    HIPCHK(hipMalloc(&t_raw.p, N * 160));
    HIPCHK(hipMalloc(&t_wide.p, N * 128));     // one 128-byte limb entry per table slot: the MSM's point format (devio.h pts_store)
    HIPCHK(hipMemcpyAsync(t_sc.p, sc.data(), N * 32, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(launch_mul_base(ctx->w, (const uint8_t *)t_sc.p, N, ctx->d_table, nullptr, (uint8_t *)t_raw.p, ctx->num_cus, ctx->stream));
    int32_t r = prep_points(ctx, (const uint8_t *)t_raw.p, N, C25519_FMT_RAW160, (uint32_t *)t_wide.p, 0, (uint32_t *)ctx->d_flag);
    if (r) return r;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->prefix.p) { hipFree(ctx->prefix.p); ctx->prefix.p = nullptr; ctx->prefix.cap = 0; }    // the normaliser's scratch (48 B per entry)
    HIPCHK(hipFree(ctx->d_table));
    ctx->d_table = (uint32_t *)t_wide.p;
    t_wide.p = nullptr;
    ctx->w = C;
    return C25519_OK;
}

// streams, events and pinned staging of a context (also of a peer context)
static bool ctx_make_streams(c25519_ctx *ctx) {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return false;
    ctx->own_stream = true;
    hipEventCreate(&ctx->ev0); hipEventCreate(&ctx->ev1);
    // The second stream carries the latency-bound chains (hash tree, sort) that the VALU-bound kernels of the main stream
    // would otherwise starve (older waves win the issue arbiter): it is created with the highest priority.
    int plo = 0, phi = 0;
    if (hipDeviceGetStreamPriorityRange(&plo, &phi) == hipSuccess && phi < plo) {
        if (hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, phi) != hipSuccess) return false;
    } else if (hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking) != hipSuccess) return false;
    hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming); hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_sort, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming); hipEventCreateWithFlags(&ctx->ev_z, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_rebind, hipEventDisableTiming); hipEventCreateWithFlags(&ctx->ev_acc, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_pts, hipEventDisableTiming);
    for (int q = 0; q < 4; q++) hipEventCreateWithFlags(&ctx->ev_grp[q], hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_split, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_lists[0], hipEventDisableTiming); hipEventCreateWithFlags(&ctx->ev_lists[1], hipEventDisableTiming);
    for (int i = 0; i < c25519_ctx::RING; i++) for (int j = 0; j < c25519_ctx::RING_EV; j++) hipEventCreate(&ctx->ring[i][j]);
    // C25519_MAX_SLOTS pass slots + the context's own record (msm.hip drec)
    if (hipMalloc((void **)&ctx->d_slots, (size_t)(C25519_MAX_SLOTS + 1) * C25519_SLOT_U32 * 4) != hipSuccess) return false;
    // (coherent + mapped: the publishing kernel writes the slots and the "published" word straight into this buffer while the host polls it)
    if (hipHostMalloc(&ctx->h_msm, (size_t)(C25519_MAX_SLOTS + 1) * C25519_SLOT_U32 * 4 + 256, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return false;
    memset((uint8_t *)ctx->h_msm + (size_t)(C25519_MAX_SLOTS + 1) * C25519_SLOT_U32 * 4, 0, 256);
    if (hipHostGetDevicePointer((void **)&ctx->hd_msm, ctx->h_msm, 0) != hipSuccess) return false;
    return true;
}
EXPORT void c25519_ctx_destroy(c25519_ctx *ctx);
c25519_ctx *ctx_peer(c25519_ctx *ctx) {
    if (ctx->peer) return ctx->peer;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    c25519_ctx *p = new c25519_ctx();
    p->device = ctx->device; p->flags = ctx->flags; p->num_cus = ctx->num_cus; p->w = ctx->w;
    p->d_table = ctx->d_table; p->d_table_ct = ctx->d_table_ct; p->owns_table = false;
    if (!ctx_make_streams(p) || hipMalloc(&p->d_flag, 256) != hipSuccess || hipMemset(p->d_flag, 0, 256) != hipSuccess) { c25519_ctx_destroy(p); return nullptr; }
    ctx->peer = p;
    return p;
}

EXPORT void c25519_ctx_destroy(c25519_ctx *ctx);
EXPORT c25519_ctx *c25519_ctx_create(int device, uint32_t flags) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        fprintf(stderr, "c25519_ctx_create: no usable HIP device %d (count %d) -- this engine has no CPU fallback\n", device, ndev);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    c25519_ctx *ctx = new c25519_ctx();
    ctx->device = device;
    ctx->flags = flags;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { c25519_ctx_destroy(ctx); return nullptr; }
    ctx->num_cus = prop.multiProcessorCount;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        fprintf(stderr, "c25519_ctx_create: warning: device arch %s, kernels are built for gfx950 only\n", prop.gcnArchName);
    if (!ctx_make_streams(ctx)) { c25519_ctx_destroy(ctx); return nullptr; }
    int w = (int)(flags & 0x1f);
    const int wide = (w >= 10 && w <= 20) ? w : (w == 0 ? 16 : 0);   // default: radix 2^16 (measured best table size / speed point)
    ctx->w = (w >= 4 && w <= 6) ? w : 9;       // 4..6: per-position LDS window tables; 9: signed comb (also bootstraps the wide table)
    std::vector<uint32_t> tab;
    if (ctx->w == 9) build_comb_table(tab); else build_window_table(ge_basepoint(), ctx->w, tab);
    if (hipMalloc(&ctx->d_table, tab.size() * 4) != hipSuccess ||
        hipMemcpy(ctx->d_table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&ctx->d_flag, 256) != hipSuccess || hipMemset(ctx->d_flag, 0, 256) != hipSuccess) {
        fprintf(stderr, "c25519_ctx_create: device allocation failed\n");
        c25519_ctx_destroy(ctx);
        return nullptr;
    }
    {   // constant-time path: radix-2^5 window tables (52 x 17 entries x 96 B)
        std::vector<uint32_t> tct;
        build_window_table(ge_basepoint(), C25519_CT_W, tct);
        if (hipMalloc(&ctx->d_table_ct, tct.size() * 4) != hipSuccess || hipMemcpy(ctx->d_table_ct, tct.data(), tct.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            fprintf(stderr, "c25519_ctx_create: device allocation failed\n");
            c25519_ctx_destroy(ctx);
            return nullptr;
        }
    }
    if (wide && build_wide_table(ctx, wide) != C25519_OK) {
        fprintf(stderr, "c25519_ctx_create: building the radix-2^%d fixed-base table failed: %s\n", wide, ctx->err.c_str());
        c25519_ctx_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

EXPORT void c25519_ctx_destroy(c25519_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    devbuf *bufs[] = {&ctx->scratch, &ctx->prefix, &ctx->tmp_a, &ctx->tmp_b, &ctx->tmp_c, &ctx->tmp_c2, &ctx->tmp_d, &ctx->tmp_e, &ctx->tmp_f, &ctx->pts_all, &ctx->dom};
    for (devbuf *b : bufs) if (b->p) hipFree(b->p);
    if (ctx->peer) { c25519_ctx_destroy(ctx->peer); ctx->peer = nullptr; }
    if (ctx->d_table && ctx->owns_table) hipFree(ctx->d_table);
    if (ctx->d_table_ct && ctx->owns_table) hipFree(ctx->d_table_ct);
    if (ctx->d_flag) hipFree(ctx->d_flag);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->aux) { hipStreamSynchronize(ctx->aux); hipStreamDestroy(ctx->aux); }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    if (ctx->ev_sort) hipEventDestroy(ctx->ev_sort);
    if (ctx->ev_in) hipEventDestroy(ctx->ev_in);
    if (ctx->ev_z) hipEventDestroy(ctx->ev_z);
    if (ctx->ev_rebind) hipEventDestroy(ctx->ev_rebind);
    if (ctx->ev_acc) hipEventDestroy(ctx->ev_acc);
    if (ctx->ev_pts) hipEventDestroy(ctx->ev_pts);
    if (ctx->ev_split) hipEventDestroy(ctx->ev_split);
    for (int q = 0; q < 2; q++) if (ctx->ev_lists[q]) hipEventDestroy(ctx->ev_lists[q]);
    for (int q = 0; q < 4; q++) if (ctx->ev_grp[q]) hipEventDestroy(ctx->ev_grp[q]);
    if (ctx->s_h2d) { hipStreamSynchronize(ctx->s_h2d); hipStreamDestroy(ctx->s_h2d); }
    if (ctx->s_d2h) { hipStreamSynchronize(ctx->s_d2h); hipStreamDestroy(ctx->s_d2h); }
    for (int i = 0; i < c25519_ctx::FFI_MAXCH; i++) { if (ctx->ev_up[i]) hipEventDestroy(ctx->ev_up[i]); if (ctx->ev_kd[i]) hipEventDestroy(ctx->ev_kd[i]); }
    if (ctx->ev_ffi) hipEventDestroy(ctx->ev_ffi);
    if (ctx->h_msm) hipHostFree(ctx->h_msm);
    if (ctx->h_stage) hipHostFree(ctx->h_stage);
    if (ctx->d_slots) hipFree(ctx->d_slots);
    for (int i = 0; i < c25519_ctx::RING; i++) for (int j = 0; j < c25519_ctx::RING_EV; j++) if (ctx->ring[i][j]) hipEventDestroy(ctx->ring[i][j]);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

EXPORT int32_t c25519_ctx_set_stream(c25519_ctx *ctx, void *hip_stream) {
    HIPCHK(hipSetDevice(ctx->device));
    if ((hipStream_t)hip_stream == ctx->stream) return C25519_OK;
    // the context's workspaces may still be in use by work enqueued on the old stream: the new stream starts after it
    if (ctx->own_stream) { hipStreamSynchronize(ctx->stream); hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    else if (hipEventRecord(ctx->ev_rebind, ctx->stream) == hipSuccess) HIPCHK(hipStreamWaitEvent((hipStream_t)hip_stream, ctx->ev_rebind, 0));
    else (void)hipGetLastError();     // the caller has already destroyed the old stream: nothing of ours can be pending on it
    ctx->stream = (hipStream_t)hip_stream;
    return C25519_OK;
}
EXPORT int32_t c25519_ctx_synchronize(c25519_ctx *ctx) {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return C25519_OK;
}
EXPORT const char *c25519_last_error(const c25519_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
EXPORT float c25519_last_kernel_ms(c25519_ctx *ctx) {
    float ms = -1.f;
    hipSetDevice(ctx->device);
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.f;
    return ms;
}

// phase of one ring entry: 0 = events 0 -> 1 (the dominant kernel), 1 = 1 -> 2 (what follows it), 2 = 3 -> 2 (a whole
// MSM / verify_batch pass), 3 = 4 -> 5 (decompression of R inside a verify_batch pass)
static float ring_phase_ms(hipEvent_t *ev, int phase, uint8_t kind) {
    static const int from[4] = {0, 1, 3, 4}, to[4] = {1, 2, 2, 5};
    if (phase < 0 || phase > 3) return -1.f;
    // events 3..5 are only recorded by MSM (kind 1: phase 2) and verify_batch (kind 2: phases 2 and 3) passes; a ring entry
    // last written by another kind of call would report the stale events of an older pass
    if ((phase == 2 && kind == 0) || (phase == 3 && kind != 2 && kind != 3)) return -1.f;
    if (kind >= 3 && phase != 3) return -1.f;                // (r6) a pass whose MSM took the mid path: no per-kernel events were recorded for it (msm.hip msm_enqueue)
    if (kind == 3) {                                          // ... its decompression bracket (events 4 -> 5) is there; there is no event 2 of this call to wait for
        float ms3 = -1.f;
        if (hipEventSynchronize(ev[5]) != hipSuccess) return -1.f;
        if (hipEventElapsedTime(&ms3, ev[4], ev[5]) != hipSuccess) return -1.f;
        return ms3;
    }
    float ms = -1.f;
    if (hipEventSynchronize(ev[2]) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, ev[from[phase]], ev[to[phase]]) != hipSuccess) return -1.f;
    return ms;
}
// phase of the call (pass, for MSM / verify_batch) made `back` calls ago on THIS context (0 = latest)
EXPORT float c25519_phase_ms(c25519_ctx *ctx, uint32_t back, int phase) {
    if (back >= ctx->ncalls || back >= (uint32_t)c25519_ctx::RING) return -1.f;
    hipSetDevice(ctx->device);
    const int idx = (int)((ctx->ncalls - 1 - back) % c25519_ctx::RING);
    return ring_phase_ms(ctx->ring[idx], phase, ctx->ring_kind[idx]);
}
// the same summed over every pass of the most recent MSM / verify_batch call (passes alternate between the context
// and its peer); *passes (may be NULL) receives their number.  -1 if the call had more passes than the ring holds.
EXPORT float c25519_last_call_phase_ms(c25519_ctx *ctx, int phase, uint32_t *passes) {
    if (passes) *passes = (uint32_t)ctx->last_passes.size();
    if (ctx->last_passes.empty() || ctx->last_passes.size() > (size_t)c25519_ctx::RING) return -1.f;
    hipSetDevice(ctx->device);
    float sum = 0.f;
    for (auto &pr : ctx->last_passes) {
        float ms = ring_phase_ms(pr.first->ring[pr.second], phase, pr.first->ring_kind[pr.second]);
        if (ms < 0) return -1.f;
        sum += ms;
    }
    return sum;
}

// ---- host-pointer staging (ffi.h) ---------------------------------------------------------------------------------
double wall_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// host clock of the phases of the latest synchronous MSM / verify_batch call on this context, microseconds since its entry: [0] inputs staged / upload
// enqueued (0 for the device-pointer forms), [1] all kernels enqueued, [2] results on the host, [3] folded and encoded: the call returns
EXPORT int32_t c25519_last_call_host_us(const c25519_ctx *ctx, double *out4) {
    for (int i = 0; i < 4; i++) out4[i] = ctx->host_us[i + 1] >= ctx->host_us[0] ? ctx->host_us[i + 1] - ctx->host_us[0] : 0.0;
    return C25519_OK;
}
// event counters of a context (diagnostics of the small path's direct publication, msm.hip wait_published): which = 0 publications that outlasted the spin
// phase (the host blocked on the stream instead), 1 lost publications (each re-run through the copy path), 2 directly published calls; anything else: 0
EXPORT uint64_t c25519_ctx_counter(const c25519_ctx *ctx, int32_t which) { return (ctx && which >= 0 && which < 8) ? ctx->counters[which] : 0; }
static inline double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int32_t ffi_begin(c25519_ctx *ctx) {
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->s_h2d) {
        HIPCHK(hipStreamCreateWithFlags(&ctx->s_h2d, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&ctx->s_d2h, hipStreamNonBlocking));
        for (int i = 0; i < c25519_ctx::FFI_MAXCH; i++) {
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_up[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_kd[i], hipEventDisableTiming));
        }
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_ffi, hipEventDisableTiming));
    }
    ctx->ffi_t0 = wall_ms();
    HIPCHK(hipEventRecord(ctx->ev_ffi, ctx->stream));            // the staging buffers may still be in use (wipes of the previous call)
    HIPCHK(hipStreamWaitEvent(ctx->s_h2d, ctx->ev_ffi, 0));
    return C25519_OK;
}
int32_t ffi_end(c25519_ctx *ctx, uint64_t h2d_bytes, uint64_t d2h_bytes) {
    const hipError_t e1 = hipStreamSynchronize(ctx->s_d2h), e2 = hipStreamSynchronize(ctx->s_h2d);
    ctx->ffi_ms = wall_ms() - ctx->ffi_t0; ctx->ffi_h2d = h2d_bytes; ctx->ffi_d2h = d2h_bytes;
    if (e1 != hipSuccess) return c25519_fail(ctx, e1, "hipStreamSynchronize(d2h)");
    if (e2 != hipSuccess) return c25519_fail(ctx, e2, "hipStreamSynchronize(h2d)");
    return C25519_OK;
}
// SMALL host-pointer calls (a few hundred KB at most): the pieces of the input are staged into ONE page-locked buffer and go up with ONE asynchronous
// copy on the COMPUTE stream -- no copy stream, no events, no second synchronisation.  The chunked path above costs such a call a dozen runtime
// calls and two pageable copies (each staged by the runtime on its own): ~45 us of a 185 us MSM of 256 terms.  d[i]: where piece i landed
// (256-byte aligned, in ctx->tmp_a).  The caller synchronises ctx->stream before it returns (the staging buffer is reused by the next call).
// zero_copy (r5): no upload at all -- d[i] are the DEVICE addresses of the page-locked staging buffer, and the kernels read the inputs in place over the link
// (a small call's upload is a 6 us copy followed by ~28 us before the first kernel starts: gpurun_out/raw/kt_small2; for a few hundred KB the kernels' own
// reads are cheaper).  The caller's kernels must read every input once (small.hip does).
int32_t ffi_small_upload(c25519_ctx *ctx, int pieces, const void *const *src, const size_t *bytes, uint8_t **d, size_t min_stage, bool zero_copy) {
    HIPCHK(hipSetDevice(ctx->device));
    ctx->ffi_t0 = wall_ms();
    size_t off[8], total = 0;
    if (pieces > 8) { ctx->err = "ffi_small_upload: too many pieces"; return -(int32_t)hipErrorInvalidValue; }
    for (int i = 0; i < pieces; i++) { off[i] = total; total += (bytes[i] + 255) & ~(size_t)255; }
    int32_t r;
    // min_stage: what the rest of the call will ask of the staging buffer (the strict z-mode of verify_batch keeps its host copies there): it must not
    // be re-allocated while the upload below is still reading it
    if ((r = ctx_host_stage(ctx, std::max(total + 256, min_stage)))) return r;
    if (zero_copy) {
        uint8_t *dv = nullptr;
        HIPCHK(hipHostGetDevicePointer((void **)&dv, ctx->h_stage, 0));
        for (int i = 0; i < pieces; i++) { if (bytes[i]) memcpy((uint8_t *)ctx->h_stage + off[i], src[i], bytes[i]); d[i] = dv + off[i]; }
        return C25519_OK;
    }
    if ((r = ctx_reserve(ctx, ctx->tmp_a, total + 256))) return r;
    for (int i = 0; i < pieces; i++) { if (bytes[i]) memcpy((uint8_t *)ctx->h_stage + off[i], src[i], bytes[i]); d[i] = (uint8_t *)ctx->tmp_a.p + off[i]; }
    if (total) HIPCHK(hipMemcpyAsync(ctx->tmp_a.p, ctx->h_stage, total, hipMemcpyHostToDevice, ctx->stream));
    return C25519_OK;
}
void ffi_small_begin(c25519_ctx *ctx) { ctx->ffi_t0 = wall_ms(); }
void ffi_small_end(c25519_ctx *ctx, uint64_t h2d_bytes, uint64_t d2h_bytes) {
    ctx->ffi_ms = wall_ms() - ctx->ffi_t0; ctx->ffi_h2d = h2d_bytes; ctx->ffi_d2h = d2h_bytes;
}
// wall-clock milliseconds and bytes moved each way by the latest host-pointer call of this context (-1 if none)
EXPORT double c25519_last_ffi_ms(const c25519_ctx *ctx, uint64_t *h2d_bytes, uint64_t *d2h_bytes) {
    if (h2d_bytes) *h2d_bytes = ctx->ffi_h2d;
    if (d2h_bytes) *d2h_bytes = ctx->ffi_d2h;
    return ctx->ffi_ms;
}
// page-locked host memory for callers that want their buffers DMA-able without a first-touch penalty (ffi.h)
EXPORT void *c25519_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
EXPORT void c25519_host_free(void *p) { if (p) hipHostFree(p); }
// release the workspaces a large call left behind (a 2^24-term MSM keeps ~2.6 GB: the records prepared ahead, the
// normaliser's prefix products, the staging copies of host-pointer calls); the next call re-allocates what it needs
EXPORT int32_t c25519_ctx_trim(c25519_ctx *ctx) {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (c25519_ctx *c = ctx; c; c = c->peer) {
        if (c != ctx && c->stream) HIPCHK(hipStreamSynchronize(c->stream));
        if (c->aux) HIPCHK(hipStreamSynchronize(c->aux));
        devbuf *bufs[] = {&c->scratch, &c->prefix, &c->tmp_a, &c->tmp_b, &c->tmp_c, &c->tmp_c2, &c->tmp_d, &c->tmp_e, &c->tmp_f, &c->pts_all, &c->dom};
        for (devbuf *b : bufs) if (b->p) { HIPCHK(hipFree(b->p)); b->p = nullptr; b->cap = 0; }
    }
    return C25519_OK;
}

// name of the kernel that phase 0 (which = 0) / phase 3 (which = 1) of the latest entry point timed
EXPORT const char *c25519_last_kernel_name(const c25519_ctx *ctx, int which) { return (ctx && which >= 0 && which < 2) ? ctx->kname[which] : ""; }

// ---- fixed base --------------------------------------------------------------------------------
int32_t mul_base_impl(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, int out_fmt, uint8_t *d_out, bool secret, const uint32_t *table_ct) {
    HIPCHK(hipSetDevice(ctx->device));
    if (out_fmt != C25519_FMT_EDWARDS_Y && out_fmt != C25519_FMT_RISTRETTO && out_fmt != C25519_FMT_RAW160) { ctx->err = "mul_base: out_fmt must be 0, 1 or 2"; return -(int32_t)hipErrorInvalidValue; }
    if (out_fmt == C25519_FMT_EDWARDS_Y) {
        int32_t r;
        if ((r = ctx_reserve(ctx, ctx->scratch, n * 128)) || (r = ctx_reserve(ctx, ctx->prefix, n * 48))) return r;
    }
    if (out_fmt == C25519_FMT_RISTRETTO) { int32_t r = ctx_reserve(ctx, ctx->tmp_e, n * 160 + 256); if (r) return r; }
    stream_wipe wipe(ctx->stream);                        // the projective scratch records and prefix products are secret-derived
    if (secret && out_fmt == C25519_FMT_EDWARDS_Y) { wipe.add(ctx->scratch.p, n * 128); wipe.add(ctx->prefix.p, n * 48); }
    if (secret && out_fmt == C25519_FMT_RISTRETTO) wipe.add(ctx->tmp_e.p, n * 160);
    hipEvent_t *ring = ctx_ring_item(ctx);
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(hipEventRecord(ring[0], ctx->stream));
    ctx->kname[0] = secret ? mul_base_ct_kernel_name(n, ctx->num_cus)
                           : (ctx->w >= 10 ? "c25519::k_mul_base_wide<OUT> (radix-2^w tables in HBM)" : ctx->w == 9 ? "c25519::k_mul_base_comb" : "c25519::k_mul_base<W, BS, OUT, false>");
    auto mul = [&](uint32_t *scratch, uint8_t *out_raw) -> hipError_t {
        return secret ? launch_mul_base_ct(d_scalars, n, table_ct ? table_ct : ctx->d_table_ct, scratch, out_raw, ctx->num_cus, ctx->stream)
                      : launch_mul_base(ctx->w, d_scalars, n, ctx->d_table, scratch, out_raw, ctx->num_cus, ctx->stream);
    };
    if (out_fmt == C25519_FMT_RAW160) {
        HIPCHK(mul(nullptr, d_out));
        HIPCHK(hipEventRecord(ring[1], ctx->stream));
    } else if (out_fmt == C25519_FMT_RISTRETTO) {         // RistrettoBasepointTable * scalar, then RistrettoPoint::compress (ristretto.rs:500-533)
        HIPCHK(mul(nullptr, (uint8_t *)ctx->tmp_e.p));
        HIPCHK(hipEventRecord(ring[1], ctx->stream));
        HIPCHK(launch_compress_ristretto((const uint8_t *)ctx->tmp_e.p, n, d_out, ctx->stream));
    } else {
        HIPCHK(mul((uint32_t *)ctx->scratch.p, nullptr));
        HIPCHK(hipEventRecord(ring[1], ctx->stream));
        HIPCHK(launch_compress_p32((const uint32_t *)ctx->scratch.p, (uint32_t *)ctx->prefix.p, n, d_out, ctx->stream));
    }
    HIPCHK(hipEventRecord(ring[2], ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    return C25519_OK;
}
EXPORT int32_t c25519_mul_base_batch_dev(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, int out_fmt, uint8_t *d_out) {
    return mul_base_impl(ctx, d_scalars, n, out_fmt, d_out, ctx_secret_default(ctx), nullptr);
}
// the same for scalars the caller declares PUBLIC: always the fast tables (addresses depend on the scalar)
EXPORT int32_t c25519_mul_base_batch_vartime_dev(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, int out_fmt, uint8_t *d_out) {
    return mul_base_impl(ctx, d_scalars, n, out_fmt, d_out, false, nullptr);
}

static inline int32_t reserve2(c25519_ctx *ctx, devbuf &a, size_t na, devbuf &b, size_t nb) { int32_t r = ctx_reserve(ctx, a, na ? na : 16); return r ? r : ctx_reserve(ctx, b, nb ? nb : 16); }
static int32_t mul_base_host(c25519_ctx *ctx, const uint8_t *scalars, uint64_t n, int out_fmt, uint8_t *out, bool secret, bool clamp, const uint32_t *table_ct) {
    HIPCHK(hipSetDevice(ctx->device));
    if (out_fmt < 0 || out_fmt > 2) { ctx->err = "mul_base: out_fmt must be 0, 1 or 2"; return -(int32_t)hipErrorInvalidValue; }
    const size_t osz = out_fmt == C25519_FMT_RAW160 ? 160 : 32;
    int32_t r;
    if ((r = reserve2(ctx, ctx->tmp_a, n * 32, ctx->tmp_b, n * osz))) return r;
    uint8_t *d_in = (uint8_t *)ctx->tmp_a.p, *d_out = (uint8_t *)ctx->tmp_b.p;
    stream_wipe wipe(ctx->stream);
    if (secret) wipe.add(d_in, n * 32);                   // the staged scalars
    const ffi_in in = {scalars, d_in, 32};
    const ffi_out o = {out, d_out, osz};
    return ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 18), &in, 1, &o, 1, [&](uint64_t lo, uint64_t m) -> int32_t {
        if (clamp) HIPCHK(launch_clamp(d_in + lo * 32, m, d_in + lo * 32, ctx->stream));
        return mul_base_impl(ctx, d_in + lo * 32, m, out_fmt, d_out + lo * osz, secret, table_ct);
    });
}
EXPORT int32_t c25519_mul_base_batch(c25519_ctx *ctx, const uint8_t *scalars, uint64_t n, int out_fmt, uint8_t *out) {
    return mul_base_host(ctx, scalars, n, out_fmt, out, ctx_secret_default(ctx), false, nullptr);
}
// EdwardsPoint::mul_base_clamped (edwards.rs:948-956): out[i] = clamp_integer(bytes[i]) * B, the scalar NOT reduced mod l
EXPORT int32_t c25519_mul_base_clamped_batch_dev(c25519_ctx *ctx, const uint8_t *d_bytes, uint64_t n, int out_fmt, uint8_t *d_out) {
    HIPCHK(hipSetDevice(ctx->device));
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->tmp_c2, n * 32 + 16))) return r;
    HIPCHK(launch_clamp(d_bytes, n, (uint8_t *)ctx->tmp_c2.p, ctx->stream));
    r = mul_base_impl(ctx, (const uint8_t *)ctx->tmp_c2.p, n, out_fmt, d_out, ctx_secret_default(ctx), nullptr);
    if (n) hipMemsetAsync(ctx->tmp_c2.p, 0, n * 32, ctx->stream);      // the clamped secrets, on every path
    return r;
}
EXPORT int32_t c25519_mul_base_clamped_batch(c25519_ctx *ctx, const uint8_t *bytes, uint64_t n, int out_fmt, uint8_t *out) {
    return mul_base_host(ctx, bytes, n, out_fmt, out, ctx_secret_default(ctx), true, nullptr);
}

// ---- constant-time fixed-base tables for a caller's point: EdwardsBasepointTable::create(&P) (edwards.rs:1131-1141) and
// RistrettoBasepointTable::create (ristretto.rs:1080-1110), then `&scalar * &table` (edwards.rs:1192-1209).  The table has the
// layout of the context's own constant-time table (radix 2^5: 52 windows x 17 entries, staged in LDS by the kernel) and is
// ALWAYS read with the full-window scan of window.rs:54-76, whatever the context's flags: these tables exist for secret scalars
// (Pedersen commitments a*G + b*H, ElGamal keys ...).
struct c25519_basetable { uint32_t *d_table; };
EXPORT c25519_basetable *c25519_basetable_create(c25519_ctx *ctx, const uint8_t *point, int in_fmt) {
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    ge_p3 P;
    if (in_fmt == C25519_FMT_RAW160) P = host_from_raw160(point);
    else if (in_fmt == C25519_FMT_EDWARDS_Y || in_fmt == C25519_FMT_RISTRETTO) {
        u32 w[8];
        memcpy(w, point, 32);
        const bool ok = in_fmt == C25519_FMT_EDWARDS_Y ? ge_decompress(P, w) : ris_decompress(P, w);
        if (!ok) { ctx->err = "basetable_create: the point does not decode"; return nullptr; }
    } else { ctx->err = "basetable_create: bad in_fmt"; return nullptr; }
    std::vector<uint32_t> tab;
    try { build_window_table(P, C25519_CT_W, tab); } catch (const std::exception &e) { ctx->err = std::string("basetable_create: ") + e.what(); return nullptr; }
    c25519_basetable *t = new (std::nothrow) c25519_basetable{nullptr};
    if (!t) return nullptr;
    if (hipMalloc((void **)&t->d_table, tab.size() * 4) != hipSuccess || hipMemcpy(t->d_table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        ctx->err = "basetable_create: device allocation failed";
        if (t->d_table) hipFree(t->d_table);
        delete t;
        return nullptr;
    }
    return t;
}
EXPORT void c25519_basetable_destroy(c25519_ctx *ctx, c25519_basetable *t) {
    if (!t) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipFree(t->d_table);
    delete t;
}
EXPORT int32_t c25519_mul_table_batch_dev(c25519_ctx *ctx, const c25519_basetable *t, const uint8_t *d_scalars, uint64_t n, int out_fmt, uint8_t *d_out) {
    if (!t) { ctx->err = "mul_table: null table"; return -(int32_t)hipErrorInvalidValue; }
    return mul_base_impl(ctx, d_scalars, n, out_fmt, d_out, true, t->d_table);
}
EXPORT int32_t c25519_mul_table_batch(c25519_ctx *ctx, const c25519_basetable *t, const uint8_t *scalars, uint64_t n, int out_fmt, uint8_t *out) {
    if (!t) { ctx->err = "mul_table: null table"; return -(int32_t)hipErrorInvalidValue; }
    return mul_base_host(ctx, scalars, n, out_fmt, out, true, false, t->d_table);
}

// ---- X25519 --------------------------------------------------------------------------------------
// X25519 public keys: u([clamp(k)] B) through the fixed-base tables and the birational map, not through the ladder
EXPORT int32_t c25519_x25519_base_batch_dev(c25519_ctx *ctx, const uint8_t *d_k, uint64_t n, uint8_t *d_out) {
    HIPCHK(hipSetDevice(ctx->device));
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->scratch, n * 128)) || (r = ctx_reserve(ctx, ctx->prefix, n * 48)) || (r = ctx_reserve(ctx, ctx->tmp_e, n * 32 + 256))) return r;
    stream_wipe wipe(ctx->stream);                        // secret-derived intermediates, on every exit path
    wipe.add(ctx->scratch.p, n * 128); wipe.add(ctx->prefix.p, n * 48); wipe.add(ctx->tmp_e.p, n * 32);
    hipEvent_t *ring = ctx_ring_item(ctx);
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(hipEventRecord(ring[0], ctx->stream));
    HIPCHK(launch_clamp(d_k, n, (uint8_t *)ctx->tmp_e.p, ctx->stream));
    if (ctx_secret_default(ctx)) HIPCHK(launch_mul_base_ct((const uint8_t *)ctx->tmp_e.p, n, ctx->d_table_ct, (uint32_t *)ctx->scratch.p, nullptr, ctx->num_cus, ctx->stream));
    else HIPCHK(launch_mul_base(ctx->w, (const uint8_t *)ctx->tmp_e.p, n, ctx->d_table, (uint32_t *)ctx->scratch.p, nullptr, ctx->num_cus, ctx->stream));
    HIPCHK(hipEventRecord(ring[1], ctx->stream));
    HIPCHK(launch_ratio_p32(1, (const uint32_t *)ctx->scratch.p, (uint32_t *)ctx->prefix.p, n, d_out, ctx->stream));   // (Z+Y)/(Z-Y)
    HIPCHK(hipEventRecord(ring[2], ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    return C25519_OK;
}
EXPORT int32_t c25519_x25519_base_batch(c25519_ctx *ctx, const uint8_t *k, uint64_t n, uint8_t *out) {
    HIPCHK(hipSetDevice(ctx->device));
    int32_t r;
    if ((r = reserve2(ctx, ctx->tmp_a, n * 32, ctx->tmp_b, n * 32))) return r;
    uint8_t *d_in = (uint8_t *)ctx->tmp_a.p, *d_out = (uint8_t *)ctx->tmp_b.p;
    stream_wipe wipe(ctx->stream);
    wipe.add(d_in, n * 32);                               // the staged secrets
    const ffi_in in = {k, d_in, 32};
    const ffi_out o = {out, d_out, 32};
    return ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 18), &in, 1, &o, 1,
                        [&](uint64_t lo, uint64_t m) -> int32_t { return c25519_x25519_base_batch_dev(ctx, d_in + lo * 32, m, d_out + lo * 32); });
}
// the ladder + the batched division for n units on stream st; scratch / prefix: n x 128 / n x 48 bytes of the caller's
static int32_t x25519_enqueue(c25519_ctx *ctx, hipStream_t st, const uint8_t *d_k, const uint8_t *d_u, uint64_t n, uint32_t *scratch, uint32_t *prefix, uint8_t *d_out, hipEvent_t *ring) {
    if (ring) HIPCHK(hipEventRecord(ring[0], st));
    HIPCHK(launch_x25519(d_k, d_u, n, scratch, st));
    if (ring) HIPCHK(hipEventRecord(ring[1], st));
    HIPCHK(launch_ratio_p32(0, scratch, prefix, n, d_out, st));   // U / W, 0 -> 0
    if (ring) HIPCHK(hipEventRecord(ring[2], st));
    return C25519_OK;
}
EXPORT int32_t c25519_x25519_batch_dev(c25519_ctx *ctx, const uint8_t *d_k, const uint8_t *d_u, uint64_t n, uint8_t *d_out) {
    HIPCHK(hipSetDevice(ctx->device));
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->scratch, n * 128)) || (r = ctx_reserve(ctx, ctx->prefix, n * 48))) return r;
    stream_wipe wipe(ctx->stream);                        // the projective result is secret-derived: wiped on every exit path
    wipe.add(ctx->scratch.p, n * 128); wipe.add(ctx->prefix.p, n * 48);
    hipEvent_t *ring = ctx_ring_item(ctx);
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    ctx->kname[0] = "c25519::k_x25519";
    if ((r = x25519_enqueue(ctx, ctx->stream, d_k, d_u, n, (uint32_t *)ctx->scratch.p, (uint32_t *)ctx->prefix.p, d_out, ring))) return r;
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    return C25519_OK;
}
// contributory (may be null): n bytes, 1 where the shared secret is non-zero -- SharedSecret::was_contributory
// (x25519-dalek/src/x25519.rs:335: a low-order u gives the all-zero output, montgomery.rs:403-412)
__global__ void __launch_bounds__(256) k_nonzero32(const uint8_t *__restrict__ in, uint64_t n, uint8_t *__restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 *q = reinterpret_cast<const uint4 *>(in) + 2 * i;
    const uint4 a = q[0], b = q[1];
    flag[i] = ((a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) != 0u) ? 1 : 0;
}
EXPORT int32_t c25519_x25519_contributory_batch_dev(c25519_ctx *ctx, const uint8_t *d_k, const uint8_t *d_u, uint64_t n, uint8_t *d_out, uint8_t *d_contributory) {
    int32_t r = c25519_x25519_batch_dev(ctx, d_k, d_u, n, d_out);
    if (r || !d_contributory || !n) return r;
    hipLaunchKernelGGL(k_nonzero32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_out, n, d_contributory);
    HIPCHK(hipGetLastError());
    return C25519_OK;
}
static int32_t x25519_host(c25519_ctx *ctx, const uint8_t *k, const uint8_t *u, uint64_t n, uint8_t *out, uint8_t *contributory) {
    HIPCHK(hipSetDevice(ctx->device));
    int32_t r;
    if ((r = reserve2(ctx, ctx->tmp_a, n * 32, ctx->tmp_b, n * 32)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 33 + 16))) return r;
    uint8_t *d_k = (uint8_t *)ctx->tmp_a.p, *d_u = (uint8_t *)ctx->tmp_b.p, *d_out = (uint8_t *)ctx->tmp_c.p, *d_fl = d_out + n * 32;
    if ((r = ctx_reserve(ctx, ctx->scratch, n * 128)) || (r = ctx_reserve(ctx, ctx->prefix, n * 48))) return r;
    stream_wipe wipe(ctx->stream);
    wipe.add(d_k, n * 32);                                // the staged secret scalars ...
    wipe.add(d_out, n * 32);                              // ... the shared secrets ...
    wipe.add(ctx->scratch.p, n * 128); wipe.add(ctx->prefix.p, n * 48);      // ... and their projective forms, on every path
    const ffi_in in[2] = {{k, d_k, 32}, {u, d_u, 32}};
    const ffi_out o[2] = {{out, d_out, 32}, {contributory, d_fl, 1}};
    // chunks alternate between the context's two streams (each with its own slice of the scratch): 2^20 ladders host to host in
    // 10.6 ms when the chunks' kernels follow each other (8.9 ms of kernel for the whole batch in one launch), see profiles
    // ... and are tapered: 1/8, 3/8, 3/8, 1/8 of the batch (what is not hidden is the first upload and the last download, and a ladder
    // kernel wants a large launch: 2^20 ladders in 9.6 ms this way, 10.0 in eight equal chunks, 11.0 in four; the whole batch in one
    // launch is 8.9 ms of kernel; profiles/r03_x25519_ffi_shapes.txt)
    const bool xtaper = n >= (1u << 19);
    return ffi_pipeline(ctx, n, xtaper ? (((n / 2) + 1023) & ~(uint64_t)1023) : ffi_chunk_units(n, 1u << 17), in, 2, o, 2, [&](uint64_t lo, uint64_t m, hipStream_t st) -> int32_t {
        int32_t rr = x25519_enqueue(ctx, st, d_k + lo * 32, d_u + lo * 32, m, (uint32_t *)ctx->scratch.p + lo * 32, (uint32_t *)ctx->prefix.p + lo * 12, d_out + lo * 32, nullptr);
        if (rr || !contributory) return rr;
        hipLaunchKernelGGL(k_nonzero32, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d_out + lo * 32, m, d_fl + lo);
        HIPCHK(hipGetLastError());
        return C25519_OK;
    }, false, 0, xtaper);
}
EXPORT int32_t c25519_x25519_batch(c25519_ctx *ctx, const uint8_t *k, const uint8_t *u, uint64_t n, uint8_t *out) { return x25519_host(ctx, k, u, n, out, nullptr); }
EXPORT int32_t c25519_x25519_contributory_batch(c25519_ctx *ctx, const uint8_t *k, const uint8_t *u, uint64_t n, uint8_t *out, uint8_t *contributory) {
    return x25519_host(ctx, k, u, n, out, contributory);
}

// ---- (de)compression -------------------------------------------------------------------------------
static int32_t decompress_enqueue(c25519_ctx *ctx, const uint8_t *d_in, uint64_t n, int in_fmt, uint8_t *d_out, uint8_t *d_ok) {
    if (in_fmt == C25519_FMT_EDWARDS_Y) HIPCHK(launch_decompress_edwards(d_in, n, d_out, d_ok, (uint32_t *)ctx->d_flag, ctx->stream));
    else HIPCHK(launch_decompress_ristretto(d_in, n, d_out, d_ok, (uint32_t *)ctx->d_flag, ctx->stream));
    return C25519_OK;
}
EXPORT int32_t c25519_decompress_batch_dev(c25519_ctx *ctx, const uint8_t *d_in, uint64_t n, int in_fmt, uint8_t *d_out, uint8_t *d_ok) {
    HIPCHK(hipSetDevice(ctx->device));
    if (in_fmt != C25519_FMT_EDWARDS_Y && in_fmt != C25519_FMT_RISTRETTO) { ctx->err = "decompress: in_fmt must be 0 or 1"; return -(int32_t)hipErrorInvalidValue; }
    HIPCHK(hipMemsetAsync(ctx->d_flag, 0, 4, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    int32_t r = decompress_enqueue(ctx, d_in, n, in_fmt, d_out, d_ok);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    uint32_t *bad = (uint32_t *)ctx->h_msm;              // pinned
    HIPCHK(hipMemcpyAsync(bad, ctx->d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return *bad ? C25519_NONE : C25519_OK;
}
EXPORT int32_t c25519_decompress_batch(c25519_ctx *ctx, const uint8_t *in, uint64_t n, int in_fmt, uint8_t *out, uint8_t *ok) {
    HIPCHK(hipSetDevice(ctx->device));
    if (in_fmt != C25519_FMT_EDWARDS_Y && in_fmt != C25519_FMT_RISTRETTO) { ctx->err = "decompress: in_fmt must be 0 or 1"; return -(int32_t)hipErrorInvalidValue; }
    int32_t r;
    if ((r = reserve2(ctx, ctx->tmp_a, n * 32, ctx->tmp_b, n * 160)) || (r = ctx_reserve(ctx, ctx->tmp_c, n + 16))) return r;
    uint8_t *d_in = (uint8_t *)ctx->tmp_a.p, *d_out = (uint8_t *)ctx->tmp_b.p, *d_ok = (uint8_t *)ctx->tmp_c.p;
    HIPCHK(hipMemsetAsync(ctx->d_flag, 0, 4, ctx->stream));
    const ffi_in i1 = {in, d_in, 32};
    const ffi_out o[2] = {{out, d_out, 160}, {ok, d_ok, 1}};
    r = ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 17), &i1, 1, o, 2,
                     [&](uint64_t lo, uint64_t m) -> int32_t { return decompress_enqueue(ctx, d_in + lo * 32, m, in_fmt, d_out + lo * 160, d_ok + lo); });
    if (r) return r;
    uint32_t *bad = (uint32_t *)ctx->h_msm;
    HIPCHK(hipMemcpyAsync(bad, ctx->d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return *bad ? C25519_NONE : C25519_OK;
}
EXPORT int32_t c25519_compress_batch_dev(c25519_ctx *ctx, const uint8_t *d_in, uint64_t n, int out_fmt, uint8_t *d_out) {
    HIPCHK(hipSetDevice(ctx->device));
    if (out_fmt != C25519_FMT_EDWARDS_Y && out_fmt != C25519_FMT_RISTRETTO) { ctx->err = "compress: out_fmt must be 0 or 1"; return -(int32_t)hipErrorInvalidValue; }
    if (out_fmt == C25519_FMT_RISTRETTO) {
        HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
        HIPCHK(launch_compress_ristretto(d_in, n, d_out, ctx->stream));
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        return C25519_OK;
    }
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->scratch, n * 128)) || (r = ctx_reserve(ctx, ctx->prefix, n * 48))) return r;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    if (n < 4096) {
        HIPCHK(launch_compress_raw(d_in, n, d_out, ctx->stream));
    } else {
        HIPCHK(launch_raw_to_p32(d_in, n, (uint32_t *)ctx->scratch.p, ctx->stream));
        HIPCHK(launch_compress_p32((const uint32_t *)ctx->scratch.p, (uint32_t *)ctx->prefix.p, n, d_out, ctx->stream));
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    return C25519_OK;
}
EXPORT int32_t c25519_compress_batch(c25519_ctx *ctx, const uint8_t *in, uint64_t n, int out_fmt, uint8_t *out) {
    HIPCHK(hipSetDevice(ctx->device));
    int32_t r;
    if ((r = reserve2(ctx, ctx->tmp_a, n * 160, ctx->tmp_b, n * 32))) return r;
    uint8_t *d_in = (uint8_t *)ctx->tmp_a.p, *d_out = (uint8_t *)ctx->tmp_b.p;
    const ffi_in i1 = {in, d_in, 160};
    const ffi_out o = {out, d_out, 32};
    return ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 17), &i1, 1, &o, 1,
                        [&](uint64_t lo, uint64_t m) -> int32_t { return c25519_compress_batch_dev(ctx, d_in + lo * 160, m, out_fmt, d_out + lo * 32); });
}

// ---- EdwardsPoint::to_montgomery_batch (edwards.rs:595-612) -------------------------------------------------
EXPORT int32_t c25519_to_montgomery_batch_dev(c25519_ctx *ctx, const uint8_t *d_in, uint64_t n, uint8_t *d_out) {
    HIPCHK(hipSetDevice(ctx->device));
    int32_t r;
    if ((r = ctx_reserve(ctx, ctx->scratch, n * 128)) || (r = ctx_reserve(ctx, ctx->prefix, n * 48))) return r;
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(launch_raw_to_p32(d_in, n, (uint32_t *)ctx->scratch.p, ctx->stream));
    HIPCHK(launch_ratio_p32(1, (const uint32_t *)ctx->scratch.p, (uint32_t *)ctx->prefix.p, n, d_out, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    return C25519_OK;
}
EXPORT int32_t c25519_to_montgomery_batch(c25519_ctx *ctx, const uint8_t *in, uint64_t n, uint8_t *out) {
    HIPCHK(hipSetDevice(ctx->device));
    int32_t r;
    if ((r = reserve2(ctx, ctx->tmp_a, n * 160, ctx->tmp_b, n * 32))) return r;
    uint8_t *d_in = (uint8_t *)ctx->tmp_a.p, *d_out = (uint8_t *)ctx->tmp_b.p;
    const ffi_in i1 = {in, d_in, 160};
    const ffi_out o = {out, d_out, 32};
    return ffi_pipeline(ctx, n, ffi_chunk_units(n, 1u << 17), &i1, 1, &o, 1,
                        [&](uint64_t lo, uint64_t m) -> int32_t { return c25519_to_montgomery_batch_dev(ctx, d_in + lo * 160, m, d_out + lo * 32); });
}

