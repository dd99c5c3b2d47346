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
}
void msm_layout(uint64_t n, c25519::msm_geom &g);
// sum_i scalars[i] * pts[i] over packed affine Niels points already on the device (enqueue, one read-back, host fold)
int32_t msm_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, c25519::ge_p3 &R);
// any point format -> packed affine Niels at d_pts[dst0 ..]; *d_badcount counts encodings that do not decode
int32_t prep_points(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_pts, uint64_t dst0, uint32_t *d_badcount);
void host_encode(const c25519::ge_p3 &R, int out_fmt, uint8_t *out);
void host_raw160(const c25519::ge_p3 &p, uint8_t *out);
c25519::ge_p3 host_from_raw160(const uint8_t *in);
