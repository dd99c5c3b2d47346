// Internal entry points of msm.hip reused by extra.hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ge26.h"
#include "ctx.h"

namespace c25519 {
// Window layout of the bucket method.  Scalars are reduced mod l (< 2^253, scalar.rs:193-205) in every MSM the reference
// performs, so the 253 bits are shared out EVENLY (msm_layout); see msm.hip "digits".
constexpr int MSM_MAX_WIN = 56;
// first_unsigned: windows k >= first_unsigned hold unsigned digits (msm_layout: the top two), the others signed ones
// bps_log2: log2 of the buckets per slice of the two-pass sort (a (window, slice) bin must fit the LDS of k_part2);
// long_cap: bucket lists longer than this go to the wave-cooperative path (> mean + 8 sigma of a balanced bucket)
struct msm_geom { int c, nwin, half, first_unsigned, bps_log2; u32 long_cap; u32 addk[8]; unsigned char pos[MSM_MAX_WIN], wid[MSM_MAX_WIN]; };
// Precomputed-static MSM: ONE bucket set for all windows.  The table holds T[k][i] = 2^(c k) P_i for every window k, so
// digit k of scalar i is a term of its own, (k, i) -> point k * ns + i, and all K * ns terms fall into the same 2^(c-1)
// buckets: one accumulation, one bucket reduction, no Horner fold.
struct msm_merged { int c, K; uint64_t ns; };
constexpr u32 LONG_CAP_MIN = 192;  // msm_geom.long_cap = max(this, 2.5 x mean list length)
constexpr u32 LONG_SEG = 1024;     // entries per wave in the long path (16 per lane)
}
// bucket accumulation (accum.hip); returns the kernel's name for the timing records
const char *launch_accumulate(const uint32_t *pts, const uint32_t *sorted, const uint32_t *base, const uint32_t *perm, uint64_t count, uint64_t n, const c25519::msm_geom &g, uint32_t *buckets, int cont, hipStream_t st);
void msm_layout(uint64_t n, c25519::msm_geom &g);
// sum_i scalars[i] * pts[i] over packed affine Niels points already on the device (enqueue, one read-back, host fold)
int32_t msm_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, c25519::ge_p3 &R);
// window width / count of the merged layout for ns static points; sum_i s_i P_i over the table (first n scalars), result to R
void msm_merged_layout(uint64_t ns, c25519::msm_merged &m);
int32_t msm_merged_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const c25519::msm_merged &m, const uint32_t *d_table, c25519::ge_p3 &R);
// the table itself: d_table[(k * ns + i)] = affine Niels record of 2^(c k) * P_i, from packed affine Niels / raw points
int32_t msm_merged_build(c25519_ctx *ctx, const uint8_t *d_points, uint64_t ns, int in_fmt, const c25519::msm_merged &m, uint32_t *d_table, uint32_t *d_badcount);
// any point format -> packed affine Niels at d_pts[dst0 ..]; *d_badcount counts encodings that do not decode
int32_t prep_points(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_pts, uint64_t dst0, uint32_t *d_badcount);
void host_encode(const c25519::ge_p3 &R, int out_fmt, uint8_t *out);
void host_raw160(const c25519::ge_p3 &p, uint8_t *out);
c25519::ge_p3 host_from_raw160(const uint8_t *in);
