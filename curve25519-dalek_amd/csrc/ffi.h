// Host-pointer entry points: chunked, overlapped staging (internal to libc25519hip.so).
//
// A host-pointer twin used to be "one copy up, compute, one copy down" on the context's stream: the link idles while the
// kernels run and the multipliers idle while the link runs.  Here the batch is cut into up to FFI_MAXCH chunks; chunk
// c+1 goes up on the context's H2D copy stream while chunk c computes on the context's stream and chunk c-1 comes down on
// the D2H copy stream (events between the three), so a call costs about max(link, kernels) + one chunk of each instead
// of their sum.  The copies are plain hipMemcpyAsync on the caller's pointers: on this platform pageable memory whose
// pages have been touched moves at the link rate (56 GB/s measured, the same as hipHostMalloc memory --
// profiles/r03_pcie_probe.txt), so an intermediate pinned ring would only add a CPU copy; what IS slow is an output buffer
// whose pages have never been touched (first-touch faults, ~5 GB/s): callers should reuse their output buffers, or get
// them from c25519_host_alloc.  With pageable memory hipMemcpyAsync returns only when its copy is done, which is why
// the loop below issues upload c+1 and compute c+1 BEFORE download c: the device always has the next chunk queued.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <chrono>
#include "ctx.h"

struct ffi_in { const void *h; void *d; size_t bpu; };      // an input array: host source, device destination, bytes per unit
struct ffi_out { void *h; const void *d; size_t bpu; };     // an output array: host destination, device source, bytes per unit

// copy streams and events of a context (created on first use); the copy streams start after everything already enqueued on
// the context's stream (its staging buffers may still be in use by a wipe of the previous call)
int32_t ffi_begin(c25519_ctx *ctx);
// the context's stream continues after the downloads (for wipes of staged secrets); records the wall-clock of the call
int32_t ffi_end(c25519_ctx *ctx, uint64_t h2d_bytes, uint64_t d2h_bytes);

// small calls: all input pieces through one page-locked buffer and one copy on the compute stream (capi.hip)
int32_t ffi_small_upload(c25519_ctx *ctx, int pieces, const void *const *src, const size_t *bytes, uint8_t **d, size_t min_stage = 0, bool zero_copy = false);
void ffi_small_begin(c25519_ctx *ctx);      // a small call that stages nothing through ffi_small_upload: starts the clock of c25519_last_ffi_ms
void ffi_small_end(c25519_ctx *ctx, uint64_t h2d_bytes, uint64_t d2h_bytes);

// Between ffi_begin and the point where ffi_pipeline (or an explicit ffi_end) takes over, an entry point queues whole-array uploads on
// the copy stream; if one of those fails it returns at once.  The guard makes that early exit drain the copy streams too (ffi_end), so
// that no queued copy is still reading the caller's memory after the call has returned its error, the figures of c25519_last_ffi_ms
// are those of THIS call, and a stream_wipe declared before it (destroyed after it) enqueues its memsets behind completed copies.
struct ffi_guard {
    c25519_ctx *ctx; bool armed = true;
    explicit ffi_guard(c25519_ctx *c) : ctx(c) {}
    ffi_guard(const ffi_guard &) = delete;
    void dismiss() { armed = false; }
    ~ffi_guard() { if (armed) (void)ffi_end(ctx, 0, 0); }
};

// units per chunk: at most FFI_MAXCH chunks of at least min_units units, a multiple of 1024
static inline uint64_t ffi_chunk_units(uint64_t n, uint64_t min_units) {
    if (n <= min_units) return n ? n : 1;
    uint64_t nch = n / min_units;
    if (nch > (uint64_t)c25519_ctx::FFI_MAXCH) nch = c25519_ctx::FFI_MAXCH;
    const uint64_t c = (n + nch - 1) / nch;
    return (c + 1023) & ~(uint64_t)1023;
}

// compute(lo, m): enqueue the kernels for units [lo, lo + m) on ctx->stream (device arrays indexed from 0); returns a status.
// compute(lo, m, stream): the same on the given stream -- chunks then alternate between the context's two streams, so that the
// kernels of neighbouring chunks overlap (a chunk alone does not fill the GPU to the occupancy the whole batch reaches, and every
// kernel has a tail); the callee must keep the chunks' scratch apart.
template <class F>
static int32_t ffi_pipeline(c25519_ctx *ctx, uint64_t n, uint64_t chunk, const ffi_in *ins, int nin, const ffi_out *outs, int nout, F &&compute, bool begun = false,
                            uint64_t extra_up_bytes = 0, bool taper = false) {
    int32_t rc = begun ? 0 : ffi_begin(ctx);       // begun: the caller has called ffi_begin and already put whole-array uploads on ctx->s_h2d
    if (rc) return rc;
    uint64_t up_bytes = extra_up_bytes, down_bytes = 0;
    // chunk boundaries.  taper: a quarter-size chunk first and last -- what is NOT overlapped is the upload of the first chunk and the
    // download of the last one, while the kernels in between want chunks large enough to fill the GPU
    uint64_t cut[c25519_ctx::FFI_MAXCH + 1];
    uint64_t nch = 0;
    cut[0] = 0;
    if (n) {
        const uint64_t small = ((chunk / 4) + 1023) & ~(uint64_t)1023;
        if (taper && n >= 2 * chunk && small) {
            const uint64_t mid = n - 2 * small;
            uint64_t k = (mid + chunk - 1) / chunk;
            if (k > (uint64_t)c25519_ctx::FFI_MAXCH - 2) k = c25519_ctx::FFI_MAXCH - 2;
            const uint64_t per = (((mid + k - 1) / k) + 1023) & ~(uint64_t)1023;
            cut[++nch] = small;
            for (uint64_t i = 1; i < k; i++) cut[nch + 1] = cut[nch] + per, nch++;
            cut[++nch] = n - small;
            cut[++nch] = n;
        } else {
            const uint64_t k = (n + chunk - 1) / chunk;
            for (uint64_t i = 1; i < k; i++) cut[++nch] = i * chunk;
            cut[++nch] = n;
        }
    }
    // two_streams: a chunk's copies travel on ITS compute stream (stream A: up 0, run 0, down 0, up 2, ...; stream B: up 1, run 1, ...),
    // so the copies of one chunk overlap the kernels of its neighbour without a third and fourth stream -- HIP maps streams onto a
    // handful of hardware queues, and a copy stream that shares a queue with a compute stream waits behind its kernels
    // (seen in a kernel trace as chunk c+1 starting exactly when chunk c ended)
    constexpr bool two_streams = std::is_invocable_v<F, uint64_t, uint64_t, hipStream_t>;
    auto stream_of = [&](uint64_t c) -> hipStream_t { return (two_streams && (c & 1)) ? ctx->aux : ctx->stream; };
    auto up = [&](uint64_t c) -> int32_t {
        const uint64_t lo = cut[c], m = cut[c + 1] - lo;
        hipStream_t cs = two_streams ? stream_of(c) : ctx->s_h2d;
        hipError_t e;
        if (two_streams && c == 1 && (e = hipStreamWaitEvent(cs, ctx->ev_fork, 0)) != hipSuccess) return c25519_fail(ctx, e, "fork");
        for (int i = 0; i < nin; i++) {
            if (!ins[i].h) continue;
            e = hipMemcpyAsync((uint8_t *)ins[i].d + lo * ins[i].bpu, (const uint8_t *)ins[i].h + lo * ins[i].bpu, m * ins[i].bpu, hipMemcpyHostToDevice, cs);
            if (e != hipSuccess) return c25519_fail(ctx, e, "H2D");
            up_bytes += m * ins[i].bpu;
        }
        e = hipEventRecord(ctx->ev_up[c], cs);
        return e == hipSuccess ? 0 : c25519_fail(ctx, e, "hipEventRecord");
    };
    auto run = [&](uint64_t c) -> int32_t {
        const uint64_t lo = cut[c], m = cut[c + 1] - lo;
        hipStream_t st = stream_of(c);
        hipError_t e = hipStreamWaitEvent(st, ctx->ev_up[c], 0);
        if (e != hipSuccess) return c25519_fail(ctx, e, "hipStreamWaitEvent");
        int32_t r;
        if constexpr (two_streams) r = compute(lo, m, st); else r = compute(lo, m);
        if (r) return r;
        e = hipEventRecord(ctx->ev_kd[c], st);
        return e == hipSuccess ? 0 : c25519_fail(ctx, e, "hipEventRecord");
    };
    auto down = [&](uint64_t c) -> int32_t {
        const uint64_t lo = cut[c], m = cut[c + 1] - lo;
        hipStream_t cs = two_streams ? stream_of(c) : ctx->s_d2h;
        hipError_t e = hipStreamWaitEvent(cs, ctx->ev_kd[c], 0);
        if (e != hipSuccess) return c25519_fail(ctx, e, "hipStreamWaitEvent");
        for (int i = 0; i < nout; i++) {
            if (!outs[i].h) continue;
            e = hipMemcpyAsync((uint8_t *)outs[i].h + lo * outs[i].bpu, (const uint8_t *)outs[i].d + lo * outs[i].bpu, m * outs[i].bpu, hipMemcpyDeviceToHost, cs);
            if (e != hipSuccess) return c25519_fail(ctx, e, "D2H");
            down_bytes += m * outs[i].bpu;
        }
        return 0;
    };
    auto all = [&]() -> int32_t {
        int32_t r;
        if (!nch) return 0;
        if (two_streams) {                                     // the second stream starts after what the caller had enqueued on the first (before this call)
            hipError_t e = hipEventRecord(ctx->ev_fork, ctx->stream);
            if (e != hipSuccess) return c25519_fail(ctx, e, "hipEventRecord");
        }
        if ((r = up(0)) || (r = run(0))) return r;
        for (uint64_t c = 0; c < nch; c++) {
            if (c + 1 < nch && ((r = up(c + 1)) || (r = run(c + 1)))) return r;
            if ((r = down(c))) return r;
        }
        return 0;
    };
    rc = all();
    if (two_streams) {                                         // the chunks' own streams carried the copies: drain them, and the first continues after the second
        const hipError_t ea = hipStreamSynchronize(ctx->aux), eb = hipStreamSynchronize(ctx->stream);
        if (!rc && ea != hipSuccess) rc = c25519_fail(ctx, ea, "hipStreamSynchronize(aux)");
        if (!rc && eb != hipSuccess) rc = c25519_fail(ctx, eb, "hipStreamSynchronize(stream)");
    }
    const int32_t r2 = ffi_end(ctx, up_bytes, down_bytes);      // drains the copy streams on every path
    return rc ? rc : r2;
}
