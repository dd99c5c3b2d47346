// Internal entry points of msm.hip reused by extra.hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "ge26.h"
#include "ctx.h"

#include "knobs.h"
namespace c25519 {
// Window layout of the bucket method.  Scalars are reduced mod l (< 2^253, scalar.rs:193-205) in every MSM the reference
// performs, so the 253 bits are shared out EVENLY (msm_layout); see msm.hip "digits".
constexpr int MSM_MAX_WIN = 56;
// first_unsigned: windows k >= first_unsigned hold unsigned digits (msm_layout: the top two), the others signed ones
// bps_log2: log2 of the buckets per slice of the two-pass sort (a (window, slice) bin must fit the LDS of k_part2);
// long_cap: bucket lists longer than this go to the wave-cooperative path (> mean + 8 sigma of a balanced bucket)
// bps[k]: log2 of the buckets per slice of window k in the chunk-local sort -- bps_log2 for a window that uses all `half` buckets, less for the
//   narrower ones (a signed window of c - 1 bits, the unsigned windows), so that every window spreads its entries over all half >> bps_log2
//   slices (msm_slice_params)
// ngroups, gstart[]: WINDOW GROUPS of a single-pass call (round 5): the windows are independent until the final fold (pippenger.rs:122-159), so the
//   bucket order, the accumulation and the bucket reduction can run group by group -- group g = windows [gstart[g], gstart[g + 1]) -- and the reduction
//   of group g overlaps the accumulation of group g + 1 instead of all of it being the exposed tail of the call.  ngroups = 1: one group (every
//   multi-pass and small call).
constexpr int MSM_MAX_GROUPS = 4;
struct msm_geom { int c, nwin, half, first_unsigned, bps_log2; u32 long_cap; u32 addk[8]; unsigned char pos[MSM_MAX_WIN], wid[MSM_MAX_WIN], bps[MSM_MAX_WIN];
                  unsigned char ngroups, gstart[MSM_MAX_GROUPS + 1]; };
__host__ __device__ inline int msm_group_of(const msm_geom &g, int k) { int r = 0; for (int i = 1; i < g.ngroups; i++) r += k >= (int)g.gstart[i] ? 1 : 0; return r; }
// Precomputed-static MSM: ONE bucket set for all windows.  The table holds T[k][i] = 2^(c k) P_i for every window k, so
// digit k of scalar i is a term of its own, (k, i) -> point k * ns + i, and all K * ns terms fall into the same 2^(c-1)
// buckets: one accumulation, one bucket reduction, no Horner fold.
struct msm_merged { int c, K; uint64_t ns; };
constexpr u32 LONG_CAP_MIN = 192;  // msm_geom.long_cap = max(this, 2.5 x mean list length)
constexpr u32 LONG_SEG = 1024;     // entries per wave in the long path (16 per lane)
}
// ---- what the translation units of the MSM share (msm.hip: driver, prep, long buckets, reduction, records; msm_sort.hip: the chunk-local
//      sort; msm_sort_matrix.hip: the digit-matrix sort (merged layout, passes below 2^16 terms); small.hip: inputs below 4096 terms; verify.hip) -------------
namespace c25519 {
struct long_item;
// bucket reduction, level A: a wave takes a SEGMENT of 64 x 2^lb buckets (2^lb consecutive ones per lane); level B: one wave (block) per window over
// its <= 64 segments.  lb = 3 for 2^15 buckets per window (c = 16), 4 for 2^16 (c = 17), less below
// (the fewest buckets per lane that leave level B its <= 64 segments: below 2^15 buckets per window a call is latency-bound and level A's serial
//  part -- two dependent additions per bucket of a lane -- is the longest link of the reduction; at least 2 per lane: the kernels start from
//  the lane's last bucket)
static inline int red_lb_log2(int half) {
    static const int lb_min = [] { const int v = C25519_KNOB("RED_LB_MIN", 1); return v < 1 ? 1 : (v > 3 ? 3 : v); }();   // A/B knob: 3 = rounds 2-3
    if (half <= 512) return 3;                      // a single segment per window: level A writes the column sums itself, no level B
    int lb = lb_min;
    while ((half >> (lb + 6)) > 64) lb++;
    return lb;
}
static inline int red_nseg(int half) { const int seg = 64 << red_lb_log2(half); return (half + seg - 1) / seg; }
// partial-result record = result slot: 56 column sums of 40 u32, then 16 words -- [0..7] counters, [8, 9] the term count the window layout
// was derived from, [10] passes summed, [11] a magic word
constexpr int REC_TERMS_LO = 8, REC_TERMS_HI = 9, REC_PASSES = 10, REC_MAGIC = 11, REC_C = 12;      // REC_C: the window width of the layout (a record with terms > 0 and no width is rejected by the fold)
constexpr u32 REC_MAGIC_VALUE = 0x52503235u;               // "52PR"
// inputs up to this many terms take the single-pass small path (small.hip): 5-bit windows below 1024 terms, 6-bit ones from there (A/B knobs of the tuning
// build: MSM_SMALL_MAX, and MSM_SMALL_C = the width from 1024 terms).  Rounds 4 and early 5: 4095 terms, 7-bit windows from 2048; round 5 .. late round 6: 12 287
// terms -- measured against the bucket pipeline's fifteen launches (profiles/r05_ab_small_path_range.txt).  Against the MID path (mid.hip, round 6) the small path
// holds only up to ~6 000 terms of raw points (profiles/r06_ab_small_mid_boundary.txt, device-resident: 6144 terms 0.164 either way; 8192 0.199 -> 0.177 ms, 10 240
// 0.237 -> 0.182, 12 287 0.266 -> 0.185 with the 12- / 13-bit windows msm_layout gives those sizes) ...
inline uint64_t msm_small_max() { static const uint64_t v = (uint64_t)C25519_KNOB("MSM_SMALL_MAX", 6143); return v; }
// ... and verify_batch -- prepared (affine) records, half of the scalars 128 bits long: nine of a mid-path layout's windows instead of all of the small path's -- only
// up to 2047 signatures (same profile, device z-mode: 2048 signatures 0.296 -> 0.273 ms, 3072 0.327 -> 0.284, 4096 0.358 -> 0.291, 6143 0.417 -> 0.309; 1024: 0.252
// against 0.257).  The term count up to which a batch's 2n + 1-term MSM -- and an MSM over ENCODED points (msm.hip msm_record_enqueue) -- stays on the small path
// (A/B knob VERIFY_SMALL_MAX; never above msm_small_max()):
inline uint64_t verify_small_max() { static const uint64_t v = (uint64_t)C25519_KNOB("VERIFY_SMALL_MAX", 4095); return v < msm_small_max() ? v : msm_small_max(); }
// up to how many terms a host-pointer verify_batch sends its five arrays up through the staged ONE-copy upload (capi.hip ffi_small_upload), whichever path its MSM takes.
// The general route's per-array pageable copies on the copy stream cost ~70 us more up to 8192 signatures, ~40 at 16 384, nothing from 32 768
// (profiles/r06_ab_verify_staged_upload.txt; rounds 5 - 6: 12 287 terms, the small path's range)
inline uint64_t small_upload_max_terms() { static const uint64_t v = (uint64_t)C25519_KNOB_LL("VERIFY_STAGED_MAX", 32769); return v; }      // (A/B knob of the tuning build)
}
static inline unsigned div_up64(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }
static inline uint32_t *slot_flags(uint32_t *slot) { return slot + c25519::MSM_MAX_WIN * 40; }
static inline uint32_t *dslot(c25519_ctx *ctx, int i) { return ctx->d_slots + (size_t)i * C25519_SLOT_U32; }
static inline const uint32_t *hslot(c25519_ctx *ctx, int i) { return (const uint32_t *)ctx->h_msm + (size_t)i * C25519_SLOT_U32; }
// the context's own record (slot C25519_MAX_SLOTS of d_slots / h_msm): where a call that answers on the host sums its passes
static inline uint32_t *drec(c25519_ctx *ctx) { return dslot(ctx, C25519_MAX_SLOTS); }
// An MSM pass is enqueued in two halves so that a caller can put other work between them (msm.hip):
//   msm_enqueue_sort  the counting sort of the term indices by bucket, bucket order, long-bucket work list -- needs only the SCALARS
//   msm_enqueue_acc   accumulation (+ the long-bucket path on the second stream) and bucket reduction -- needs the POINTS (affine Niels records)
struct msm_plan {
    c25519::msm_geom g; uint64_t n, nb; int nseg; uint32_t max_items, max_long;
    uint32_t *base, *sorted, *buckets, *perm, *SW, *counters, *lgids, *lfirst, *segs, *bad_ws, *bad_sticky = nullptr; c25519::long_item *items;
    hipStream_t sort_stream;
    hipEvent_t ev_partition = nullptr;      // in: if set, recorded on the sort's stream after the partition half (k_sweep_local, k_bin_totals) of the chunk-local sort
};
void msm_sort_params(uint64_t n, c25519::msm_geom &g);
void msm_slice_params(c25519::msm_geom &g);
int32_t msm_enqueue_sort(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n_scalars, const c25519::msm_geom &g, uint32_t *d_slot, hipStream_t sort_stream, msm_plan &pl,
                         const c25519::msm_merged *md = nullptr, uint64_t n_carve = 0, hipEvent_t lists_free = nullptr, int parity = -1);
struct msm_matrix_sort_args {
    const uint8_t *d_scalars; uint64_t n_scalars, n; int nchunk; bool use_part; int SL, PART_CHUNK, pchunks;
    uint16_t *D; uint32_t *counts, *P1, *cc, *bin_base, *flags, *totals, *ord_hist;
};
int32_t msm_matrix_sort_enqueue(c25519_ctx *ctx, const c25519::msm_geom &g, const c25519::msm_merged *md, msm_plan &pl, const msm_matrix_sort_args &a, hipStream_t st);
int32_t msm_enqueue_acc(c25519_ctx *ctx, const msm_plan &pl, const uint32_t *d_pts, uint32_t *d_slot, hipEvent_t *ring, hipEvent_t wait_acc, bool cont = false, bool reduce = true,
                        const uint32_t *d_bad_sticky = nullptr);
struct mid_run;
// sort + accumulate + reduce of one pass over prepared records; inputs of at most msm_small_max() terms take the small path (run: see msm_mid_enqueue; honoured by the mid path only)
int32_t msm_enqueue(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, const c25519::msm_geom &g, uint32_t *d_slot, hipEvent_t *ring,
                    hipStream_t sort_stream, hipEvent_t wait_acc = nullptr, const struct mid_run *run = nullptr);
// the whole MSM of at most msm_small_max() terms in two launches, column sums of the layout g to d_slot (small.hip).  src_fmt: 0 = raw 160-byte points,
// 1 = affine Niels records (128 bytes); flags: the slot's counters (bit 255 of a scalar is ORed into flags[0])
int32_t msm_small_enqueue(c25519_ctx *ctx, const uint8_t *d_scalars, const void *d_points, int src_fmt, uint64_t n, const c25519::msm_geom &g, uint32_t *d_slot, hipStream_t st);
c25519::ge_p3 host_p40(const uint32_t *t);
c25519::ge_p3 msm_horner(const uint32_t *cols, const c25519::msm_geom &g);
int32_t slots_collect(c25519_ctx *ctx, int count);
int32_t rec_collect(c25519_ctx *ctx);
void slot_init(uint32_t *d_slot, uint64_t terms, const uint32_t *d_pre, hipStream_t st, int c);
int32_t records_fold(const uint8_t *records, uint64_t count, c25519::ge_p3 &R, uint32_t flags[8], std::string *err);
struct pass_set { c25519_ctx *c[4]; int lanes; };
int32_t passes_begin(c25519_ctx *ctx, uint64_t passes, pass_set &ps);
int32_t passes_join(c25519_ctx *ctx, pass_set &ps);
hipEvent_t *pass_ring(c25519_ctx *owner, c25519_ctx *c, uint8_t kind);
void launch_merged_table(const uint8_t *in_raw, uint64_t ns, int c, int K, uint8_t *out_raw, hipStream_t st);
// bucket reduction with four waves per point operation (reduce.hip)
// (windows [k0, k1): one window group, msm_geom)
void launch_bucket_reduce4(const uint32_t *buckets, const c25519::msm_geom &g, int nseg, uint32_t *SW, uint32_t *d_slot, const uint32_t *bad_ws, hipStream_t st, int k0 = 0, int k1 = -1);
void msm_set_groups(c25519::msm_geom &g, int groups, int last);
// (r6) the mid path (mid.hip): 12 288 .. msm_mid_max() terms in four launches on one stream.  reduce_publish: what the fused bucket reduction (reduce.hip
// k_reduce_b4pub) does once its last window is through -- hdr: write the record header (terms, width; the MSM) or leave the slot's own (verify_batch: k_slot_init made it);
// on: the columns went to the context's page-locked host slot, release `seq` into the host's sequence word
// dev_flags (hdr 0 and on): the counters of the DEVICE slot (k_slot_init's header, what the hash and decompression kernels counted) -- copied into the published record
namespace c25519 { struct reduce_publish { int on; uint32_t *host_flag; uint32_t seq, terms_lo, terms_hi, c; int hdr; const uint32_t *dev_flags; }; }
void launch_bucket_reduce_pub(const uint32_t *buckets, const c25519::msm_geom &g, int nseg, uint32_t *SW, uint32_t *out, const uint32_t *blockflags, int nflags, uint32_t *done_cnt,
                              const c25519::reduce_publish &pub, hipStream_t st);
void launch_order_place(const uint32_t *totals, uint64_t nb, const uint32_t *ord_hist, uint32_t *ord_cursor, uint32_t *perm, const c25519::msm_geom &g, hipStream_t st);
uint64_t msm_mid_max();
bool msm_mid_serves_terms(uint64_t n);      // prepared records: is this term count inside the path's range (before a layout exists)
bool msm_mid_serves(uint64_t n, const c25519::msm_geom &g, bool prepared);      // prepared: the records exist (a decompression made them)
// run (may be null; prepared records only): the pass runs on run->stream instead of the context's main stream (verify_batch: the stream its scalars were made on -- no
// hand-over in front of the digits), waits for run->recs_ready (the records, made on the other stream) only in front of the accumulation, and negates there the
// records sign_first .. sign_first + sign_count - 1 whose sign_z16 entry (16 bytes each, bit 127) is set (verify.hip k_apply_sign: the sign of the device z-mode's z_i);
// the context's main stream continues behind the pass
struct mid_run { hipStream_t stream; hipEvent_t recs_ready; const uint8_t *sign_z16; uint64_t sign_first, sign_count; };
int32_t msm_mid_enqueue(c25519_ctx *ctx, const uint8_t *d_scalars, const void *points, int src_fmt, uint64_t n, const c25519::msm_geom &g, uint32_t *d_slot, int hdr, uint64_t terms, hipEvent_t *ring,
                        const mid_run *run = nullptr);
void launch_apply_sign(uint32_t *pts, uint64_t dst0, const uint8_t *z16, uint64_t n, hipStream_t st);      // verify.hip
void launch_prep_basepoint(uint32_t *pts, uint64_t dst, hipStream_t st);
void launch_record_sum(uint32_t *rec, const uint32_t *slots, int cnt, int nwin, int first, hipStream_t st);
// bucket accumulation (accum.hip); returns the kernel's name for the timing records
const char *launch_accumulate(const uint32_t *pts, const uint32_t *sorted, const uint32_t *base, const uint32_t *perm, uint64_t count, uint64_t n, const c25519::msm_geom &g, uint32_t *buckets, int cont, hipStream_t st);
// the same with long_blocks blocks in front that fold the over-long lists of the mid path (items: mid_item work list of mid.hip; counters[0] = its length)
const char *launch_accumulate_long(const uint32_t *pts, const uint32_t *sorted, const uint32_t *base, const uint32_t *perm, uint64_t count, uint64_t n, const c25519::msm_geom &g, uint32_t *buckets,
                                   const void *items, const uint32_t *counters, uint32_t *seg_sums, uint32_t *long_done, uint32_t max_items, uint32_t long_blocks, hipStream_t st);
// cmax (0 = the default, 17): upper limit of the window width -- verify_batch asks for 16 (its z_i are 128-bit: eight 16-bit windows exactly);
// c_exact (0 = choose): the width itself (records_fold re-derives a record's layout from its header)
void msm_layout(uint64_t n, c25519::msm_geom &g, int cmax = 0, int c_exact = 0);
// sum_i scalars[i] * pts[i] over packed affine Niels points already on the device (enqueue, one read-back, host fold)
int32_t msm_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, c25519::ge_p3 &R);
// window width / count of the merged layout for ns static points; sum_i s_i P_i over the table (first n scalars), result to R
void msm_merged_layout(uint64_t ns, c25519::msm_merged &m);
int32_t msm_merged_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const c25519::msm_merged &m, const uint32_t *d_table, c25519::ge_p3 &R);
// the table itself: d_table[(k * ns + i)] = affine Niels record of 2^(c k) * P_i, from packed affine Niels / raw points
int32_t msm_merged_build(c25519_ctx *ctx, const uint8_t *d_points, uint64_t ns, int in_fmt, const c25519::msm_merged &m, uint32_t *d_table, uint32_t *d_badcount);
// any point format -> packed affine Niels at d_pts[dst0 ..]; *d_badcount counts encodings that do not decode
// expect_affine (raw points): normally Z = 1 (VerifyingKey points): one cheap pass first, the general normaliser behind it only if a point had another Z (msm.hip k_prep_affine)
int32_t prep_points(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_pts, uint64_t dst0, uint32_t *d_badcount, bool expect_affine = false);
int32_t prep_points_on(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_pts, uint64_t dst0, uint32_t *d_badcount, hipStream_t st, devbuf &pre, bool expect_affine = false);
void host_encode(const c25519::ge_p3 &R, int out_fmt, uint8_t *out);
void host_raw160(const c25519::ge_p3 &p, uint8_t *out);
c25519::ge_p3 host_from_raw160(const uint8_t *in);
