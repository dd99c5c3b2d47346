// Internal entry points of msm.hip reused by extra.hip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ge26.h"
#include "ctx.h"

// sum_i scalars[i] * pts[i] over packed affine Niels points already on the device
int32_t msm_core(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, const uint32_t *d_pts, c25519::ge_p3 &R, hipEvent_t *ring, hipStream_t sort_stream = nullptr,
                 void *extra_dst = nullptr, const void *extra_src = nullptr, size_t extra_bytes = 0);   // one more small D2H before the final sync
// any point format -> packed affine Niels at d_pts[dst0 ..]; *d_badcount counts encodings that do not decode
int32_t prep_points(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, uint32_t *d_pts, uint64_t dst0, uint32_t *d_badcount);
void host_encode(const c25519::ge_p3 &R, int out_fmt, uint8_t *out);
void host_raw160(const c25519::ge_p3 &p, uint8_t *out);
c25519::ge_p3 host_from_raw160(const uint8_t *in);
