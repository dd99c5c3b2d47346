// Internal context definition shared by capi.hip and msm.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>

struct devbuf { void *p = nullptr; size_t cap = 0; };
constexpr int C25519_CTR_PUBLISH_BLOCKED = 0, C25519_CTR_PUBLISH_LOST = 1, C25519_CTR_PUBLISH_DIRECT = 2;
// internal status of rec_collect / wait_published (msm.hip): the stream drained without error and the small path's record never arrived -- the caller re-runs the
// call through the slot + copy path; no entry point returns this value
constexpr int32_t C25519_LOST_PUBLICATION = INT32_MIN + 7;

// result slot of one MSM / verify_batch pass (msm.hip): 56 column sums of 40 u32 + 16 u32 of flags and counters
constexpr int C25519_SLOT_U32 = 56 * 40 + 16, C25519_MAX_SLOTS = 16;

struct c25519_ctx {
    int device = 0;
    uint32_t flags = 0;
    int num_cus = 0;
    int w = 6;                  // fixed-base window width
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // bracket of the most recent entry point
    // ring of per-call (per-pass) phase events.  Fixed base / X25519: {start, after the dominant kernel, end}.
    // MSM / verify_batch pass: {before k_accumulate, after it, end of the pass, start of the pass, before / after the
    // decompression of R}
    static const int RING = 64, RING_EV = 6;
    hipEvent_t ring[RING][RING_EV] = {};
    uint8_t ring_kind[RING] = {};      // who recorded the entry: 0 a per-item call (events 0..2), 1 an MSM pass (0..3), 2 a verify_batch pass (0..5), 3 / 4 a verify_batch / MSM pass whose MSM took the mid path (no events 0..2)
    uint64_t ncalls = 0;
    std::vector<std::pair<c25519_ctx *, int>> last_passes;   // (context, ring index) of every pass of the latest MSM / verify_batch call
    uint32_t *d_table = nullptr;   // fixed-base table of algorithm `w` (LDS window / comb tables: canonical words; radix-2^C: limb records)
    uint32_t *d_table_ct = nullptr;   // radix-2^5 LDS window tables for the constant-time (full-scan) fixed-base kernel
    void *d_flag = nullptr;        // 256 bytes of device flags / small results
    hipStream_t aux = nullptr;     // second stream: latency-bound side chains run beside VALU-bound kernels
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_sort = nullptr, ev_in = nullptr, ev_z = nullptr, ev_rebind = nullptr, ev_acc = nullptr, ev_pts = nullptr;
    hipEvent_t ev_grp[4] = {nullptr, nullptr, nullptr, nullptr};   // single-pass MSM in window groups: [q] = the accumulation of group q has finished (msm.hip msm_enqueue_acc)
    hipEvent_t ev_split = nullptr;                      // verify_batch in two halves (verify.hip): the keys' records and the basepoint's are complete on the main stream
    hipEvent_t ev_lists[2] = {nullptr, nullptr};        // multi-pass MSM: [q] = recorded when the pass with list parity q was enqueued (the accumulation before it has finished)
    void *h_msm = nullptr;                               // pinned, coherent, device-mapped: C25519_MAX_SLOTS + 1 result slots and the "published" word behind them (msm.hip publish_and_wait)
    // (r5) small calls that answer on the host: the last kernel of the small path writes the record straight into the page-locked host slot (h_msm is coherent and
    // mapped: hd_msm is the device's view of it) and releases a sequence word; no slot-clearing launches before, no copy launch after (small.hip, msm.hip)
    uint32_t *hd_msm = nullptr;                          // device pointer of h_msm
    bool want_direct = false;                            // set by an entry point that will read the record on the host right away
    const uint32_t *direct_extra = nullptr;              // with direct_seq: two device words to publish as the record's counters [2], [3] (verify.hip, small batches)
    bool no_direct_once = false;                         // (r6) set by a caller that re-runs a call whose publication was lost: the next enqueue takes the slot + copy path
    uint32_t direct_seq = 0;                             // != 0: the enqueued small pass publishes itself under this sequence number (rec_collect polls for it)
    bool solo = false;                                   // set by the entry points for a call of ONE pass on this context alone: its bucket reduction may run on the main stream (msm.hip msm_enqueue_acc)
    uint32_t publish_seq = 0;                            // sequence number of the latest publication
    uint64_t counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // c25519_ctx_counter: [0] publications that outlasted the spin phase (the host blocked on the stream), [1] lost publications (re-run through the copy path), [2] directly published calls
    hipEvent_t coarse_wait = nullptr;                    // set by an enqueue function of a long call: the host blocks on it before it polls for the results
    // host clock of the latest synchronous MSM / verify_batch call, microseconds: [0] entry, [1] inputs staged / upload enqueued, [2] all kernels enqueued,
    // [3] results on the host, [4] folded / encoded (c25519_last_call_host_us)
    double host_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void *h_stage = nullptr; size_t h_stage_cap = 0;     // pinned, grown on demand: hram / s / z of the strict z-mode of verify_batch (ctx_host_stage)
    uint32_t *d_slots = nullptr;                         // device: C25519_MAX_SLOTS result slots (written by this context and its peer)
    // a second set of streams / workspaces on the same device (shares the fixed-base tables): multi-pass MSM and
    // verify_batch enqueue alternate passes on it, so that the low-VALU phases of one pass (normalise, sort, reduce)
    // overlap the accumulation of the other.  Created on first use.
    c25519_ctx *peer = nullptr;
    bool owns_table = true;
    devbuf scratch, prefix;        // P32 points and 48-byte prefix products
    devbuf tmp_a, tmp_b, tmp_c, tmp_c2, tmp_d, tmp_e, tmp_f;  // staging for the host-pointer entry points / msm
    const uint32_t *cont_buckets = nullptr;                    // where the latest MSM pass on this context keeps its bucket sums (checked by a continuing pass)
    devbuf dom; std::vector<uint8_t> h_dom;                      // Ed25519ph: dom2 of the latest prehashed call (device copy / host source of its upload)
    devbuf pts_all;                                            // gather records of passes 1.. of a multi-pass MSM (prepared ahead in one launch)
    // names of the kernels the latest entry point launched: [0] its dominant kernel (k_mul_base*, k_x25519, k_var_base, k_accumulate),
    // [1] the decompression of R_i in a verify_batch pass -- what c25519_phase_ms phases 0 and 3 time
    const char *kname[2] = {"", ""};
    // host-pointer entry points (ffi.h): copy streams, per-chunk events, figures of the latest host-pointer call
    static const int FFI_MAXCH = 8;
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_up[FFI_MAXCH] = {}, ev_kd[FFI_MAXCH] = {}, ev_ffi = nullptr;
    double ffi_t0 = 0, ffi_ms = -1;                      // wall-clock of the latest host-pointer call
    uint64_t ffi_h2d = 0, ffi_d2h = 0;                   // bytes it moved each way
    std::string err;
};

// Zeroes device buffers on the given stream when it goes out of scope: secret-derived scratch is wiped on EVERY exit path of an
// entry point (also the early returns of a failed launch), after whatever the entry point enqueued before.
struct stream_wipe {
    hipStream_t st;
    std::vector<std::pair<void *, size_t>> bufs;          // (a vector: no registration is ever dropped, however many an entry point adds)
    explicit stream_wipe(hipStream_t s) : st(s) {}
    stream_wipe(const stream_wipe &) = delete;
    void add(void *q, size_t bytes) { if (q && bytes) bufs.emplace_back(q, bytes); }
    ~stream_wipe() { for (auto &b : bufs) (void)hipMemsetAsync(b.first, 0, b.second, st); }
};

int32_t c25519_fail(c25519_ctx *ctx, hipError_t e, const char *where);
double wall_us();
int32_t ctx_reserve(c25519_ctx *ctx, devbuf &b, size_t bytes);
// page-locked host staging of at least `bytes` bytes, kept by the context (fresh pageable buffers pay first-touch faults at ~5 GB/s inside a copy)
int32_t ctx_host_stage(c25519_ctx *ctx, size_t bytes);
c25519_ctx *ctx_peer(c25519_ctx *ctx);      // nullptr if it cannot be created
// out[i] = scalars[i] * B.  secret: constant-time table scan (k_mul_base<5, CT>) and wiped scratch; otherwise the context's
// fast tables (the radix-2^16 HBM tables by default), whose addresses depend on the scalar.
// table_ct (may be null = the context's basepoint table): a caller's constant-time window table (c25519_basetable)
int32_t mul_base_impl(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, int out_fmt, uint8_t *d_out, bool secret, const uint32_t *table_ct = nullptr);
int32_t mul_batch_impl(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, int out_fmt, uint8_t *d_out, uint8_t *d_ok, bool ct);
// ring entry of a per-item call (events 0..2)
inline hipEvent_t *ctx_ring_item(c25519_ctx *ctx) { const int idx = (int)(ctx->ncalls++ % c25519_ctx::RING); ctx->ring_kind[idx] = 0; return ctx->ring[idx]; }
inline bool ctx_secret_default(const c25519_ctx *ctx) { return !(ctx->flags & 0x100u); }   // !C25519_FLAG_VARTIME_TABLES
