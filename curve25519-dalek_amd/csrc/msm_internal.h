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
struct msm_geom { int c, nwin, half; u32 addk[8]; unsigned char pos[MSM_MAX_WIN], wid[MSM_MAX_WIN]; };
constexpr u32 LONG_CAP = 192;      // buckets with more entries go to the wave-cooperative path: > mean + 8 sigma of a balanced bucket (mean <= 96)
constexpr u32 LONG_SEG = 1024;     // entries per wave in the long path (16 per lane)
}
// bucket accumulation (accum.hip), chained-carry (c1) and ten-column (c0) field arithmetic
void launch_accumulate_c0(int pipe, const uint32_t *pts, const uint32_t *sorted, const uint32_t *base, const uint32_t *perm, uint64_t count, uint64_t n, const c25519::msm_geom &g, uint32_t *buckets, hipStream_t st);
void launch_accumulate_c1(int pipe, const uint32_t *pts, const uint32_t *sorted, const uint32_t *base, const uint32_t *perm, uint64_t count, uint64_t n, const c25519::msm_geom &g, uint32_t *buckets, hipStream_t st);
void msm_layout(uint64_t n, c25519::msm_geom &g);
// sum_i scalars[i] * pts[i] over packed affine Niels points already on the device (enqueue, one read-back, host fold)
int32_t msm_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, c25519::ge_p3 &R);
// any point format -> packed affine Niels at d_pts[dst0 ..]; *d_badcount counts encodings that do not decode
int32_t prep_points(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_pts, uint64_t dst0, uint32_t *d_badcount);
void host_encode(const c25519::ge_p3 &R, int out_fmt, uint8_t *out);
void host_raw160(const c25519::ge_p3 &p, uint8_t *out);
c25519::ge_p3 host_from_raw160(const uint8_t *in);
