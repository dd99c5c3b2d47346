// Launchers for the gfx950 kernels in kernels.hip / msm.hip (internal to libc25519hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace c25519 {

hipError_t launch_mul_base(int w, const uint8_t *scalars, uint64_t n, const uint32_t *tab, uint32_t *scratch,
                           uint8_t *out_raw, int num_cus, hipStream_t st);
constexpr int C25519_CT_W = 5;     // window width of the constant-time fixed-base tables (52 windows x 17 entries, 85 KB of LDS)
const char *mul_base_ct_kernel_name(uint64_t n, int num_cus);
hipError_t launch_mul_base_ct(const uint8_t *scalars, uint64_t n, const uint32_t *tab_ct, uint32_t *scratch, uint8_t *out_raw, int num_cus, hipStream_t st, bool p40 = false);
hipError_t launch_prep_compressed_keys_and_r(const uint8_t *pks, const uint8_t *sigs, uint64_t n, uint32_t *pts, uint32_t *bad_count, hipStream_t st);
hipError_t launch_prep_small_verify(const uint8_t *pks, const uint8_t *pk_points /* may be null */, const uint8_t *sigs, uint64_t n, uint32_t *pts, uint32_t *bad, hipStream_t st);   // n <= 128, one block, record 0 = B
hipError_t launch_prep_compressed(int fmt, const uint8_t *in, uint64_t stride_items, uint64_t n, uint32_t *pts, uint64_t dst0, uint32_t *bad_count, bool shared, hipStream_t st);
hipError_t launch_clamp(const uint8_t *in, uint64_t n, uint8_t *out, hipStream_t st);
hipError_t launch_mul_base_p40(int w, const uint8_t *scalars, uint64_t n, const uint32_t *tab, uint32_t *out40, int num_cus, hipStream_t st);
// flags[0] += signatures with a non-canonical s; flags[1] |= 1 if the message offsets are not monotone / run past msgs_len
hipError_t launch_hram(const uint8_t *msgs, const uint64_t *msg_off, uint64_t msgs_len, const uint8_t *sigs, const uint8_t *pks, uint64_t n, uint8_t *hram, uint32_t *flags, hipStream_t st);
// Ed25519ph / Ed25519ctx: SHA-512(dom2 || R || A || M); msg_off == nullptr: messages of fixed_len bytes (verify.hip k_hram_dom)
hipError_t launch_hram_dom(const uint8_t *dom, uint32_t dom_len, const uint8_t *msgs, const uint64_t *msg_off, uint64_t msgs_len, uint32_t fixed_len, const uint8_t *sigs, const uint8_t *pks,
                           uint64_t n, uint8_t *hram, uint32_t *flags, hipStream_t st);
hipError_t launch_compress_p32(const uint32_t *scratch, uint32_t *prefix, uint64_t n, uint8_t *out, hipStream_t st);
hipError_t launch_x25519(const uint8_t *k, const uint8_t *u, uint64_t n, uint32_t *scratch, hipStream_t st);
hipError_t launch_ratio_p32(int mode, const uint32_t *scratch, uint32_t *prefix, uint64_t n, uint8_t *out, hipStream_t st);
hipError_t launch_decompress_edwards(const uint8_t *in, uint64_t n, uint8_t *out_raw, uint8_t *ok, uint32_t *any_bad, hipStream_t st);
hipError_t launch_decompress_ristretto(const uint8_t *in, uint64_t n, uint8_t *out_raw, uint8_t *ok, uint32_t *any_bad, hipStream_t st);
hipError_t launch_compress_ristretto(const uint8_t *in_raw, uint64_t n, uint8_t *out, hipStream_t st);
hipError_t launch_compress_raw(const uint8_t *in_raw, uint64_t n, uint8_t *out, hipStream_t st);
hipError_t launch_raw_to_p32(const uint8_t *in_raw, uint64_t n, uint32_t *scratch, hipStream_t st);
hipError_t launch_selftest_c0(int op, const uint32_t *a, const uint32_t *b, uint64_t n, uint8_t *out, hipStream_t st);
hipError_t launch_selftest_c1(int op, const uint32_t *a, const uint32_t *b, uint64_t n, uint8_t *out, hipStream_t st);
hipError_t launch_probe(int which, uint32_t *out, int iters, unsigned grid, hipStream_t st);
hipError_t launch_selftest_scalar(int op, const uint32_t *a, const uint32_t *b, uint64_t n, uint32_t *out, hipStream_t st);

}  // namespace c25519
