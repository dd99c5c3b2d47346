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

// units per chunk: at most FFI_MAXCH chunks of at least min_units units, a multiple of 1024
static inline uint64_t ffi_chunk_units(uint64_t n, uint64_t min_units) {
    if (n <= min_units) return n ? n : 1;
    uint64_t nch = n / min_units;
    if (nch > (uint64_t)c25519_ctx::FFI_MAXCH) nch = c25519_ctx::FFI_MAXCH;
    const uint64_t c = (n + nch - 1) / nch;
    return (c + 1023) & ~(uint64_t)1023;
}

// compute(lo, m): enqueue the kernels for units [lo, lo + m) on ctx->stream (device arrays indexed from 0); returns a status
template <class F>
static int32_t ffi_pipeline(c25519_ctx *ctx, uint64_t n, uint64_t chunk, const ffi_in *ins, int nin, const ffi_out *outs, int nout, F &&compute, bool begun = false,
                            uint64_t extra_up_bytes = 0) {
    int32_t rc = begun ? 0 : ffi_begin(ctx);       // begun: the caller has called ffi_begin and already put whole-array uploads on ctx->s_h2d
    if (rc) return rc;
    uint64_t up_bytes = extra_up_bytes, down_bytes = 0;
    const uint64_t nch = n ? (n + chunk - 1) / chunk : 0;
    auto up = [&](uint64_t c) -> int32_t {
        const uint64_t lo = c * chunk, m = (lo + chunk < n ? chunk : n - lo);
        for (int i = 0; i < nin; i++) {
            if (!ins[i].h) continue;
            hipError_t e = hipMemcpyAsync((uint8_t *)ins[i].d + lo * ins[i].bpu, (const uint8_t *)ins[i].h + lo * ins[i].bpu, m * ins[i].bpu, hipMemcpyHostToDevice, ctx->s_h2d);
            if (e != hipSuccess) return c25519_fail(ctx, e, "H2D");
            up_bytes += m * ins[i].bpu;
        }
        hipError_t e = hipEventRecord(ctx->ev_up[c], ctx->s_h2d);
        return e == hipSuccess ? 0 : c25519_fail(ctx, e, "hipEventRecord");
    };
    auto run = [&](uint64_t c) -> int32_t {
        const uint64_t lo = c * chunk, m = (lo + chunk < n ? chunk : n - lo);
        hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->ev_up[c], 0);
        if (e != hipSuccess) return c25519_fail(ctx, e, "hipStreamWaitEvent");
        int32_t r = compute(lo, m);
        if (r) return r;
        e = hipEventRecord(ctx->ev_kd[c], ctx->stream);
        return e == hipSuccess ? 0 : c25519_fail(ctx, e, "hipEventRecord");
    };
    auto down = [&](uint64_t c) -> int32_t {
        const uint64_t lo = c * chunk, m = (lo + chunk < n ? chunk : n - lo);
        hipError_t e = hipStreamWaitEvent(ctx->s_d2h, ctx->ev_kd[c], 0);
        if (e != hipSuccess) return c25519_fail(ctx, e, "hipStreamWaitEvent");
        for (int i = 0; i < nout; i++) {
            if (!outs[i].h) continue;
            e = hipMemcpyAsync((uint8_t *)outs[i].h + lo * outs[i].bpu, (const uint8_t *)outs[i].d + lo * outs[i].bpu, m * outs[i].bpu, hipMemcpyDeviceToHost, ctx->s_d2h);
            if (e != hipSuccess) return c25519_fail(ctx, e, "D2H");
            down_bytes += m * outs[i].bpu;
        }
        return 0;
    };
    auto all = [&]() -> int32_t {
        int32_t r;
        if (!nch) return 0;
        if ((r = up(0)) || (r = run(0))) return r;
        for (uint64_t c = 0; c < nch; c++) {
            if (c + 1 < nch && ((r = up(c + 1)) || (r = run(c + 1)))) return r;
            if ((r = down(c))) return r;
        }
        return 0;
    };
    rc = all();
    const int32_t r2 = ffi_end(ctx, up_bytes, down_bytes);      // drains the copy streams on every path
    return rc ? rc : r2;
}
