// Variable-time multiscalar multiplication and verify_batch (placeholder translation unit: the
// kernels land here next; until then the entry points report hipErrorNotSupported loudly).
#include <hip/hip_runtime.h>
#include "../../include/c25519_hip.h"
#include "ge26.h"
#include "kernels.h"
#include "ctx.h"
#define EXPORT extern "C" __attribute__((visibility("default")))
static int32_t unsupported(c25519_ctx *ctx, const char *what) { ctx->err = std::string(what) + ": not implemented yet"; return -(int32_t)hipErrorNotSupported; }
EXPORT int32_t c25519_msm_vartime_dev(c25519_ctx *ctx, const uint8_t *, const uint8_t *, uint64_t, int, int, uint8_t *) { return unsupported(ctx, "msm_vartime_dev"); }
EXPORT int32_t c25519_msm_vartime(c25519_ctx *ctx, const uint8_t *, const uint8_t *, uint64_t, int, int, uint8_t *) { return unsupported(ctx, "msm_vartime"); }
EXPORT int32_t c25519_msm_partial_dev(c25519_ctx *ctx, const uint8_t *, const uint8_t *, uint64_t, int, uint8_t *) { return unsupported(ctx, "msm_partial_dev"); }
EXPORT int32_t c25519_fold_partials(c25519_ctx *ctx, const uint8_t *, uint64_t, int, uint8_t *) { return unsupported(ctx, "fold_partials"); }
EXPORT int32_t ed25519_verify_batch_dev(c25519_ctx *ctx, const uint8_t *, const uint64_t *, uint64_t, const uint8_t *, const uint8_t *, uint64_t, uint32_t) { return unsupported(ctx, "verify_batch_dev"); }
EXPORT int32_t ed25519_verify_batch(c25519_ctx *ctx, const uint8_t *, const uint64_t *, const uint8_t *, const uint8_t *, uint64_t, uint32_t) { return unsupported(ctx, "verify_batch"); }
