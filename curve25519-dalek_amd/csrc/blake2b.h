// BLAKE2b (RFC 7693): the compression function F, host + device.  Used by ONE thing: the hash tree of verify_batch's device z-mode (verify.hip k_ztree_*, k_zderive,
// and the host restatement ztree_host_zs) -- a construction of this library, NOT part of Ed25519 and NOT of the reference (include/c25519_hip.h C25519_Z_DEVICE);
// H(R || A || M) stays SHA-512 as RFC 8032 and the reference (ed25519-dalek verifying.rs) define it.
//
// Why a second hash.  The levels of that tree are dependent compressions executed by ONE wave per SIMD, and a lone wave issues one vector instruction every ~8.7
// cycles whatever the instruction is (profiles/r06_instruction_rates.txt): a compression costs its instruction count.  SHA-512: 80 rounds, ~5 000 instructions,
// 16 - 18 us per level; BLAKE2b: 12 rounds of eight G, ~2 300 instructions for the same 128 bytes -- and a tree of 2^14 signatures is nine dependent compressions
// between H(R || A || M) and the first scalar of the batch equation (round 6: 143 of a 450 us call).  Both are 512-bit-state, 128-byte-block hashes of the 128-bit
// collision level the 32-byte nodes are cut to.
//
// Every value the tree computes is a plain unkeyed BLAKE2b digest of a byte string (tests/pyref.py device_zs restates it with hashlib.blake2b), so the construction is
// pinned against an independent implementation; tests/test_fe26_host.py checks this F against hashlib on random inputs and on RFC 7693's "abc" vector.
#pragma once
#include "sc_sha.h"

namespace c25519 {

C25519_HD void blake2b_g(u64 &a, u64 &b, u64 &c, u64 &d, u64 x, u64 y) {      // RFC 7693 section 3.1
    a = a + b + x; d = rotr64(d ^ a, 32);
    c = c + d;     b = rotr64(b ^ c, 24);
    a = a + b + y; d = rotr64(d ^ a, 16);
    c = c + d;     b = rotr64(b ^ c, 63);
}
// unkeyed, sequential mode: h0 = IV ^ parameter block (digest length, key length 0, fanout 1, depth 1); IV = SHA-512's (RFC 7693 section 2.6)
C25519_HD void blake2b_init(u64 h[8], u32 outlen) {
    const u64 iv[8] = C25519_SHA512_IV;
    for (int i = 0; i < 8; i++) h[i] = iv[i];
    h[0] ^= 0x01010000ull ^ (u64)outlen;
}
// F (RFC 7693 section 3.2): m = the 128-byte block as sixteen LITTLE-endian words, t = bytes absorbed so far including this block (below 2^64 here), last = final block
C25519_HD void blake2b_compress(u64 h[8], const u64 m[16], u64 t, bool last) {
    const u64 iv[8] = C25519_SHA512_IV;
    u64 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    u64 v8 = iv[0], v9 = iv[1], v10 = iv[2], v11 = iv[3], v12 = iv[4] ^ t, v13 = iv[5], v14 = last ? ~iv[6] : iv[6], v15 = iv[7];
    // a round with the message schedule sigma written out (the indices must be compile-time constants for m[] to stay in registers on the device)
#define C25519_B2B_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15)                        \
    blake2b_g(v0, v4, v8, v12, m[s0], m[s1]);   blake2b_g(v1, v5, v9, v13, m[s2], m[s3]);                               \
    blake2b_g(v2, v6, v10, v14, m[s4], m[s5]);  blake2b_g(v3, v7, v11, v15, m[s6], m[s7]);                              \
    blake2b_g(v0, v5, v10, v15, m[s8], m[s9]);  blake2b_g(v1, v6, v11, v12, m[s10], m[s11]);                            \
    blake2b_g(v2, v7, v8, v13, m[s12], m[s13]); blake2b_g(v3, v4, v9, v14, m[s14], m[s15]);
    C25519_B2B_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    C25519_B2B_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
    C25519_B2B_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
    C25519_B2B_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
    C25519_B2B_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
    C25519_B2B_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
    C25519_B2B_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
    C25519_B2B_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
    C25519_B2B_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
    C25519_B2B_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
    C25519_B2B_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
    C25519_B2B_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
#undef C25519_B2B_ROUND
    h[0] ^= v0 ^ v8;  h[1] ^= v1 ^ v9;  h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

}  // namespace c25519
