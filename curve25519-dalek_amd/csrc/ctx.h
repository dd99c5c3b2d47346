// Internal context definition shared by capi.hip and msm.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

struct devbuf { void *p = nullptr; size_t cap = 0; };

struct c25519_ctx {
    int device = 0;
    uint32_t flags = 0;
    int num_cus = 0;
    int w = 6;                  // fixed-base window width
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // bracket of the most recent entry point
    // ring of per-call phase events: slot k = {start, after dominant kernel, end} of call (ncalls-1-k)
    static const int RING = 64;
    hipEvent_t ring[RING][3] = {};
    uint64_t ncalls = 0;
    uint32_t *d_table = nullptr;   // fixed-base table, [NWIN][HALF+1][24] u32
    void *d_flag = nullptr;        // 256 bytes of device flags / small results
    hipStream_t aux = nullptr;     // second stream: latency-bound side chains run beside VALU-bound kernels
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr, ev_sort = nullptr;
    void *h_pinned = nullptr; size_t h_pinned_cap = 0;   // pinned host staging for small read-backs
    void *h_msm = nullptr;                               // 20 KB pinned: window totals + flags of msm_core
    // a second set of streams / workspaces on the same device (shares the fixed-base table): multi-pass MSM and
    // verify_batch run alternate passes on it from a second host thread, so that the low-VALU phases of one pass
    // (normalise, sort, reduce, read-back) overlap the accumulation of the other.  Created on first use.
    c25519_ctx *peer = nullptr;
    bool owns_table = true;
    devbuf scratch, prefix;        // P32 points and 48-byte prefix products
    devbuf tmp_a, tmp_b, tmp_c, tmp_d, tmp_e, tmp_f;  // staging for the host-pointer entry points / msm
    std::string err;
};

int32_t c25519_fail(c25519_ctx *ctx, hipError_t e, const char *where);
int32_t ctx_reserve(c25519_ctx *ctx, devbuf &b, size_t bytes);
c25519_ctx *ctx_peer(c25519_ctx *ctx);      // nullptr if it cannot be created
