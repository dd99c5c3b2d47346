/* The C ABI from plain C (C11, gcc): proves include/c25519_hip.h is a C header and that the library can be driven
 * without Python or C++.  RFC 7748 section 6.1 (Alice/Bob) through c25519_x25519_batch, 8 * B through
 * c25519_mul_base_batch against the reference's BASE8 constant source (edwards.rs test module), a 3-term MSM,
 * ed25519_verify_batch on RFC 8032 7.1 TEST 1 / TEST 2 (both z-modes, the three verdicts, bad offsets) and the
 * one-process multi-context entry points.
 * Exit code 0 = all good.  Built and run by tests/test_gpu_abi_c.py. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/c25519_hip.h"

static int hex2bin(const char *h, uint8_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) { unsigned v; if (sscanf(h + 2 * i, "%2x", &v) != 1) return -1; out[i] = (uint8_t)v; }
    return 0;
}

int main(void) {
    c25519_ctx *ctx = c25519_ctx_create(0, 0);
    if (!ctx) { fprintf(stderr, "no context\n"); return 2; }
    /* RFC 7748 6.1 */
    uint8_t k[2][32], u[2][32], out[2][32], want[32];
    hex2bin("77076d0a7318a57d3c16c17251b26645df4c2f87ebc0992ab177fba51db92c2a", k[0], 32);   /* Alice private */
    hex2bin("5dab087e624a8a4b79e17f8b83800ee66f3bb1292618b6fd1c2f8b27ff88e0eb", k[1], 32);   /* Bob private   */
    hex2bin("de9edb7d7b7dc1b4d35b61c2ece435373f8343c85b78674dadfc7e146f882b4f", u[0], 32);   /* Bob public    */
    hex2bin("8520f0098930a754748b7ddcb43ef75a0dbf3a0d26381af4eba4a98eaa9b4e6a", u[1], 32);   /* Alice public  */
    hex2bin("4a5d9d5ba4ce2de1728e3bf480350f25e07e21c947d19e3376f09b3c1e161742", want, 32);   /* shared secret */
    if (c25519_x25519_batch(ctx, &k[0][0], &u[0][0], 2, &out[0][0]) != C25519_OK) { fprintf(stderr, "x25519: %s\n", c25519_last_error(ctx)); return 3; }
    if (memcmp(out[0], want, 32) || memcmp(out[1], want, 32)) { fprintf(stderr, "x25519 mismatch\n"); return 4; }
    /* fixed base: scalars 1 and 8; 1 * B must be the basepoint encoding 5866...66 */
    uint8_t s[2][32] = {{1}, {8}}, enc[2][32], bp[32];
    memset(bp, 0x66, 32); bp[0] = 0x58;
    if (c25519_mul_base_batch(ctx, &s[0][0], 2, C25519_FMT_EDWARDS_Y, &enc[0][0]) != C25519_OK) return 5;
    if (memcmp(enc[0], bp, 32)) { fprintf(stderr, "1*B mismatch\n"); return 6; }
    /* MSM: 3*B + 5*B == 8*B, points given compressed */
    uint8_t sc[2][32] = {{3}, {5}}, pts[2][32], sum[32];
    memcpy(pts[0], bp, 32); memcpy(pts[1], bp, 32);
    if (c25519_msm_vartime(ctx, &sc[0][0], &pts[0][0], 2, C25519_FMT_EDWARDS_Y, C25519_FMT_EDWARDS_Y, sum) != C25519_OK) return 7;
    if (memcmp(sum, enc[1], 32)) { fprintf(stderr, "msm mismatch\n"); return 8; }
    /* an encoding that is not on the curve -> NONE, like Option::None of optional_multiscalar_mul */
    memset(pts[1], 0, 32); pts[1][0] = 2;
    if (c25519_msm_vartime(ctx, &sc[0][0], &pts[0][0], 2, C25519_FMT_EDWARDS_Y, C25519_FMT_EDWARDS_Y, sum) != C25519_NONE) return 9;
    /* ed25519_dalek::verify_batch: RFC 8032 7.1 TEST 1 (empty message) and TEST 2 (one byte) as one batch, both z-modes */
    uint8_t vpk[2][32], vsig[2][64], vmsg[1] = {0x72};
    uint64_t voff[3] = {0, 0, 1};
    hex2bin("d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", vpk[0], 32);
    hex2bin("3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", vpk[1], 32);
    hex2bin("e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b", vsig[0], 64);
    hex2bin("92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00", vsig[1], 64);
    for (uint32_t z = 0; z < 2; z++) {
        if (ed25519_verify_batch(ctx, vmsg, voff, &vsig[0][0], &vpk[0][0], 2, z) != C25519_OK) { fprintf(stderr, "verify_batch (z_mode %u): %s\n", z, c25519_last_error(ctx)); return 10; }
        vsig[1][7] ^= 1;                                                   /* a forged R */
        if (ed25519_verify_batch(ctx, vmsg, voff, &vsig[0][0], &vpk[0][0], 2, z) != C25519_VERIFY) return 11;
        vsig[1][7] ^= 1;
        vsig[0][63] |= 0x20;                                               /* s >= 2^253: non-canonical */
        if (ed25519_verify_batch(ctx, vmsg, voff, &vsig[0][0], &vpk[0][0], 2, z) != C25519_SCALAR_FORMAT) return 12;
        vsig[0][63] &= (uint8_t)~0x20;
    }
    voff[1] = 5;                                                           /* offsets that are not monotone: refused, nothing read out of bounds */
    if (ed25519_verify_batch(ctx, vmsg, voff, &vsig[0][0], &vpk[0][0], 2, 1) >= 0) return 13;
    /* the same batch over two contexts driven from this one process (c25519_msm_vartime_multi's sibling) */
    voff[1] = 0;
    c25519_ctx *ctx2 = c25519_ctx_create(0, 0);
    if (!ctx2) return 14;
    c25519_ctx *both[2] = {ctx, ctx2};
    if (ed25519_verify_batch_multi(both, 2, vmsg, voff, &vsig[0][0], &vpk[0][0], 2, 0) != C25519_OK) return 15;
    memcpy(pts[1], bp, 32);
    if (c25519_msm_vartime_multi(both, 2, &sc[0][0], &pts[0][0], 2, C25519_FMT_EDWARDS_Y, C25519_FMT_EDWARDS_Y, sum) != C25519_OK || memcmp(sum, enc[1], 32)) return 16;
    /* (r3) a constant-time table for a caller's point: 8 * (3 B) == 24 B through c25519_basetable_create / c25519_mul_table_batch */
    uint8_t three[1][32] = {{3}}, p3[32], s24[1][32] = {{24}}, e24[32], t24[32];
    if (c25519_mul_base_batch(ctx, &three[0][0], 1, C25519_FMT_EDWARDS_Y, p3) != C25519_OK) return 17;
    c25519_basetable *tab = c25519_basetable_create(ctx, p3, C25519_FMT_EDWARDS_Y);
    if (!tab) { fprintf(stderr, "basetable: %s\n", c25519_last_error(ctx)); return 18; }
    if (c25519_mul_table_batch(ctx, tab, &s[1][0], 1, C25519_FMT_EDWARDS_Y, t24) != C25519_OK) return 19;
    if (c25519_mul_base_batch(ctx, &s24[0][0], 1, C25519_FMT_EDWARDS_Y, e24) != C25519_OK || memcmp(t24, e24, 32)) { fprintf(stderr, "mul_table mismatch\n"); return 20; }
    c25519_basetable_destroy(ctx, tab);
    /* (r3) buffers from c25519_host_alloc, mul_base_clamped == x25519 public key in Edwards form, was_contributory on a low-order point */
    uint8_t *hb = (uint8_t *)c25519_host_alloc(4 * 32);
    if (!hb) return 21;
    memcpy(hb, k[0], 32);
    if (c25519_mul_base_clamped_batch(ctx, hb, 1, C25519_FMT_EDWARDS_Y, hb + 32) != C25519_OK) return 22;
    uint8_t zero_u[32] = {0}, fl[1] = {9};
    if (c25519_x25519_contributory_batch(ctx, k[0], zero_u, 1, hb + 64, fl) != C25519_OK || fl[0] != 0) return 23;          /* u = 0 is of low order: all-zero secret */
    if (c25519_x25519_contributory_batch(ctx, k[0], u[0], 1, hb + 64, fl) != C25519_OK || fl[0] != 1 || memcmp(hb + 64, want, 32)) return 24;
    uint64_t up = 0, down = 0;
    if (c25519_last_ffi_ms(ctx, &up, &down) < 0 || up != 64 || down != 33) return 25;
    c25519_host_free(hb);
    /* (r3) the multi-rank exchange without torch: two partial-result records (a packed point each) fold to the sum */
    uint8_t raw3[160], raw5[160], recs[2 * C25519_PARTIAL_RECORD_BYTES], folded[32];
    uint8_t five[1][32] = {{5}};
    if (c25519_mul_base_batch(ctx, &three[0][0], 1, C25519_FMT_RAW160, raw3) != C25519_OK || c25519_mul_base_batch(ctx, &five[0][0], 1, C25519_FMT_RAW160, raw5) != C25519_OK) return 26;
    if (c25519_partial_record_pack(raw3, C25519_OK, NULL, recs) != C25519_OK || c25519_partial_record_pack(raw5, C25519_OK, NULL, recs + C25519_PARTIAL_RECORD_BYTES) != C25519_OK) return 27;
    if (c25519_fold_partial_records(NULL, recs, 2, C25519_FMT_EDWARDS_Y, folded) != C25519_OK || memcmp(folded, enc[1], 32)) { fprintf(stderr, "record fold mismatch\n"); return 28; }
    if (c25519_partial_record_pack(raw5, C25519_NONE, NULL, recs + C25519_PARTIAL_RECORD_BYTES) != C25519_OK) return 29;
    if (c25519_fold_partial_records(NULL, recs, 2, C25519_FMT_EDWARDS_Y, folded) != C25519_NONE) return 30;
    if (c25519_ctx_trim(ctx) != C25519_OK) return 31;
    /* order checks: B is torsion-free and not of small order; the identity is both; y = 2 does not decode */
    uint8_t chk[3][32], fl3[3];
    memcpy(chk[0], bp, 32); memset(chk[1], 0, 32); chk[1][0] = 1; memset(chk[2], 0, 32); chk[2][0] = 2;
    if (c25519_point_order_checks_batch(ctx, &chk[0][0], 3, C25519_FMT_EDWARDS_Y, C25519_POINT_SMALL_ORDER | C25519_POINT_TORSION_FREE, fl3) != C25519_OK) return 32;
    if (fl3[0] != (C25519_POINT_DECODES | C25519_POINT_TORSION_FREE) || fl3[1] != (C25519_POINT_DECODES | C25519_POINT_SMALL_ORDER | C25519_POINT_TORSION_FREE) || fl3[2] != 0) { fprintf(stderr, "order checks: %d %d %d\n", fl3[0], fl3[1], fl3[2]); return 33; }
    c25519_ctx_destroy(ctx2);
    c25519_ctx_destroy(ctx);
    printf("abi_c_smoke ok\n");
    return 0;
}
