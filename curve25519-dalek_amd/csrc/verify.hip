// ed25519-dalek verify_batch (batch.rs:146-251) on the GPU: SHA-512(R || A || M) per signature, the z_i (the reference's transcript on the
// host, or the device hash tree), the batch scalars mod l (sc28.h), and the 2n+1-term MSM through the Pippenger pipeline of msm.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <stdexcept>
#include <string>
#include <vector>
#include <functional>
#include "../../include/c25519_hip.h"
#include "devio.h"
#include "sc_sha.h"
#include "blake2b.h"
#include "sc28.h"
#include "kernels.h"
#include "ctx.h"
#include "msm_internal.h"
#include "msm_sort.h"
#include "ffi.h"

using namespace c25519;
#define EXPORT extern "C" __attribute__((visibility("default")))
#define HIPCHK(call)                                                \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return c25519_fail(ctx, _e, #call);   \
    } while (0)

namespace c25519 {

// ================================================================================================
// verify_batch kernels
// ================================================================================================
// hram_i = SHA-512(R_i || A_i || M_i) (batch.rs:179-191): 64-byte digest out; flags[0] += non-canonical s,
// flags[1] |= 1 if the message offsets are not monotone or run past msgs_len (that message is hashed as empty)
__global__ void __launch_bounds__(256) k_hram(const uint8_t *__restrict__ msgs, const u64 *__restrict__ msg_off, u64 msgs_len, const uint8_t *__restrict__ sigs,
                                              const uint8_t *__restrict__ pks, u64 n, uint8_t *__restrict__ hram, u32 *__restrict__ flags, uint8_t *__restrict__ hred = nullptr) {
    C25519_PRIO_CHAIN();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 r[8], a[8], s[8];
    load8(sigs, 2 * i, r);
    load8(sigs, 2 * i + 1, s);
    load8(pks, i, a);
    if (!sc28_words_canonical(s)) atomicAdd(&flags[0], 1u);     // signature.rs:89-94 check_scalar: s < l, a word-wise comparison
    sha512_stream st;
    st.init();
    for (int j = 0; j < 4; j++) st.w[j] = bswap64((u64)r[2 * j] | ((u64)r[2 * j + 1] << 32));       // R || A fills the
    for (int j = 0; j < 4; j++) st.w[4 + j] = bswap64((u64)a[2 * j] | ((u64)a[2 * j + 1] << 32));   // first 64 bytes
    st.fill = 64; st.total = 64;
    const u64 o0 = msg_off[i], o1 = msg_off[i + 1];
    const bool okoff = o0 <= o1 && o1 <= msgs_len;
    if (!okoff) atomicOr(&flags[1], 1u);
    const uint8_t *m = msgs + o0;
    const u64 len = okoff ? o1 - o0 : 0;
    st.put_bytes(m, len);
    st.finish();
    u32 w[16];
    sha512_digest_words(st.h, w);
    uint4 *q = reinterpret_cast<uint4 *>(hram) + 4 * i;
    for (int j = 0; j < 4; j++) q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
    if (hred) {                                              // h_i mod l, 32 bytes: what the device z-tree commits to (below)
        u32 o[8];
        sc28_to_words(sc28_from_wide(w), o);
        store8(hred, i, o);
    }
}
// Ed25519ph / Ed25519ctx (RFC 8032 5.1; verifying.rs:520-534 RCompute::new with prehash_ctx = Some(ctx)): hram_i = SHA-512(dom2 || R_i || A_i || M_i),
// dom2 = "SigEd25519 no Ed25519 collisions" || 0x01 || len(ctx) || ctx (built by the host, `dom_len` bytes in device memory: the same for the whole
// batch).  The prefix has any length from 34 to 289 bytes, so R and A are absorbed at an arbitrary byte position (sha512_stream::put_bytes) -- a
// kernel of its own: k_hram keeps its register-resident first block.  msg_off == nullptr: messages of `fixed_len` bytes each (the 64-byte prehashes).
__global__ void __launch_bounds__(256) k_hram_dom(const uint8_t *__restrict__ dom, u32 dom_len, const uint8_t *__restrict__ msgs, const u64 *__restrict__ msg_off, u64 msgs_len,
                                                  u32 fixed_len, const uint8_t *__restrict__ sigs, const uint8_t *__restrict__ pks, u64 n, uint8_t *__restrict__ hram, u32 *__restrict__ flags) {
    C25519_PRIO_CHAIN();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 s[8];
    load8(sigs, 2 * i + 1, s);
    if (!sc28_words_canonical(s)) atomicAdd(&flags[0], 1u);     // signature.rs:89-94 check_scalar
    sha512_stream st;
    st.init();
    st.put_bytes(dom, dom_len);
    st.put_bytes(sigs + 64 * i, 32);                            // R
    st.put_bytes(pks + 32 * i, 32);                             // A
    u64 o0 = i * (u64)fixed_len, o1 = o0 + fixed_len;
    if (msg_off) { o0 = msg_off[i]; o1 = msg_off[i + 1]; }
    const bool okoff = o0 <= o1 && o1 <= msgs_len;
    if (!okoff) atomicOr(&flags[1], 1u);
    st.put_bytes(msgs + o0, okoff ? o1 - o0 : 0);
    st.finish();
    u32 w[16];
    sha512_digest_words(st.h, w);
    uint4 *q = reinterpret_cast<uint4 *>(hram) + 4 * i;
    for (int j = 0; j < 4; j++) q[j] = make_uint4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
}
// the same reduction for hashes that were computed elsewhere
__global__ void __launch_bounds__(256) k_hram_mod_l(const uint8_t *__restrict__ hram, u64 n, uint8_t *__restrict__ hred) {
    C25519_PRIO_CHAIN();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 *hw = reinterpret_cast<const u32 *>(hram) + 16 * i;
    u32 w[16], o[8];
    for (int j = 0; j < 16; j++) w[j] = hw[j];
    sc28_to_words(sc28_from_wide(w), o);
    store8(hred, i, o);
}

// device z-mode (C25519_Z_DEVICE; NOT the reference's derivation -- see include/c25519_hip.h).  The z_i must depend on
// every input bit of the batch (a per-signature or per-subtree derivation allows a 2^64 meet-in-the-middle forgery), so
// they are derived from the root of a hash tree over what the reference's transcript absorbs (batch.rs:191-199) -- hram_i =
// H(R_i || A_i || M_i) and the 32-byte s_i of every signature -- with hram_i taken mod l (32 bytes instead of 64: the batch
// equation only ever sees h_i mod l, batch.rs:213-217, so that is the value to bind; two blocks per four signatures instead of three).
// v5 (round 6): the tree's hash is BLAKE2b (RFC 7693; blake2b.h has the reasons -- 12 rounds instead of 80 on a chain of dependent compressions
// that one wave per SIMD executes; rounds 2-5: SHA-512, v1-v4).  Every node is a PLAIN unkeyed BLAKE2b-256 digest:
//   node = BLAKE2b-256( TAG(level, inputs, n) || data ), where TAG is one 128-byte block (domain separation and shape binding: the tag string, zeros, then
//   level, number of inputs of the level and batch size as little-endian u64) whose compression is done once on the host (the per-level states below), and
//   `data` has a fixed length per level.  32-byte nodes give the 128-bit level of the z_i.
//   level 0: data = (hram_4j mod l) || s_4j || ... || (hram_4j+3 mod l) || s_4j+3 (absent = zero bytes): 2 blocks per 4 signatures, one lane each
//   level l: data = four children, ONE compression per node (out[j] = node_l(in[4j] || .. || in[4j+3])).  These levels are pure latency
//            (one dependent compression is ~8 us for a single wave; SHA-512: 16 - 18): five of them run inside one block (k_ztree_block).
constexpr int ZTREE_MAX_LEVELS = 16;
struct ztree_ivs { u64 iv[ZTREE_MAX_LEVELS][8]; };       // iv[l]: the BLAKE2b state after TAG(l, ..)
__global__ void __launch_bounds__(256) k_ztree_first(const uint8_t *__restrict__ hred, const uint8_t *__restrict__ sigs, u64 n, ztree_ivs ivs, uint8_t *__restrict__ out) {
    C25519_PRIO_CHAIN();
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 m_out = (n + 3) / 4;
    if (j >= m_out) return;
    u64 hs[8];
    for (int q = 0; q < 8; q++) hs[q] = ivs.iv[0][q];
    u64 rec[32];                                          // 4 records of 64 bytes = 2 blocks (little-endian words: the bytes as they are)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const u64 c = 4 * j + r;
        const u64 *h = reinterpret_cast<const u64 *>(hred) + 4 * c, *sg = reinterpret_cast<const u64 *>(sigs) + 8 * c + 4;
#pragma unroll
        for (int q = 0; q < 4; q++) rec[8 * r + q] = c < n ? h[q] : 0ull;
#pragma unroll
        for (int q = 0; q < 4; q++) rec[8 * r + 4 + q] = c < n ? sg[q] : 0ull;
    }
#pragma unroll 1
    for (int blk = 0; blk < 2; blk++) {
        u64 w[16];
#pragma unroll
        for (int q = 0; q < 16; q++) w[q] = blk == 0 ? rec[q] : rec[16 + q];
        blake2b_compress(hs, w, 256 + 128 * (u64)blk, blk == 1);
    }
    u64 *o = reinterpret_cast<u64 *>(out) + 4 * j;
    for (int q = 0; q < 4; q++) o[q] = hs[q];
}
// FIVE levels per launch: a block owns 1024 consecutive nodes of level `level` - 1 and reduces them to ONE node of level `level` + 4 through LDS
// (4^5 = 1024: the partition is aligned, so every node is the same function of the same children, with the same per-level tag, as when each level
// was a launch of its own -- rounds 2-3: four launches of one level each and a single-block tail, 317 us of launch gaps and single-wave
// compressions on the critical chain of verify_batch; now one launch of 256 blocks and the tail).  nlev: levels to do (5), or "until one
// node is left" for the tail (m_in <= 1024, one block).  The global node counts decide which children exist.
__global__ void __launch_bounds__(256) k_ztree_block(const uint8_t *__restrict__ in, u64 m_in, u32 level, ztree_ivs ivs, uint8_t *__restrict__ out, int nlev) {
    C25519_PRIO_CHAIN();
    __shared__ u64 buf0[1024 * 4], buf1[256 * 4];
    const u64 base = (u64)blockIdx.x * 1024;
    for (u32 i = threadIdx.x; i < 1024 * 4; i += 256) buf0[i] = base + i / 4 < m_in ? reinterpret_cast<const u64 *>(in)[base * 4 + i] : 0ull;
    __syncthreads();
    u64 *cur = buf0, *nxt = buf1;
    u64 m = m_in, gbase = base;                                        // nodes of the current level (global), global index of this block's first one
    u32 cnt = 1024;                                                    // nodes of the current level held by this block
    for (int lv = 0; lv < nlev && m > 1; lv++) {
        const u64 mo = (m + 3) / 4, gb = gbase / 4;
        const u32 co = cnt / 4;
        if (threadIdx.x < co && gb + threadIdx.x < mo) {
            u64 hs[8], w[16];
            for (int q = 0; q < 8; q++) hs[q] = ivs.iv[level][q];
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
                const u32 c = 4 * threadIdx.x + ch;
#pragma unroll
                for (int q = 0; q < 4; q++) w[4 * ch + q] = gbase + c < m ? cur[4 * c + q] : 0ull;
            }
            blake2b_compress(hs, w, 256, true);
            for (int q = 0; q < 4; q++) nxt[4 * threadIdx.x + q] = hs[q];
        }
        __syncthreads();
        u64 *t = cur; cur = nxt; nxt = t;
        m = mo; gbase = gb; cnt = co; level++;
    }
    if (threadIdx.x < 4) reinterpret_cast<u64 *>(out)[(u64)blockIdx.x * 4 + threadIdx.x] = cur[threadIdx.x];
}
// step 3: (z_4j .. z_4j+3) = the four 16-byte quarters of BLAKE2b-512(root || LE64(j)); n4 = ceil(n/4)
// lanes, z16 has room for 4*n4 entries.  A quarter is read as SIGN-MAGNITUDE: bit 127 = sign, bits 0..126 = |z_i|, i.e.
// z_i is uniform on {-(2^127-1) .. 2^127-1} (2^128 - 1 values; a forged batch passes with probability <= 2^-127.99
// against the reference's 2^-128).  Why signed: the MSM recodes scalars into signed windows, and a magnitude below 2^127
// never carries out of its eighth 16-bit window, so the R_i terms stay out of windows 8..15; an unsigned 128-bit z_i
// (C25519_Z_TRANSCRIPT) puts the carry digit +1 of about half of all R_i into ONE bucket of window 8 (the long-bucket
// path takes it).  The sign is applied to the stored point (k_apply_sign), the MSM scalar is |z_i|.
__global__ void __launch_bounds__(256) k_zderive(const uint8_t *__restrict__ root, u64 n4, uint8_t *__restrict__ z16) {
    C25519_PRIO_CHAIN();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const u64 *h = reinterpret_cast<const u64 *>(root);
    u64 hs[8], w[16];   // 40-byte message: one block
    blake2b_init(hs, 64);
    for (int q = 0; q < 4; q++) w[q] = h[q];
    w[4] = i;
    for (int q = 5; q < 16; q++) w[q] = 0;
    blake2b_compress(hs, w, 40, true);
    u64 *o = reinterpret_cast<u64 *>(z16) + 8 * i;
    for (int q = 0; q < 8; q++) o[q] = hs[q];
}
// R_i <- -R_i where z_i is negative (device z-mode): swap y+x / y-x, negate 2dxy of the stored affine Niels record
__global__ void __launch_bounds__(256) k_apply_sign(u32 *__restrict__ pts, u64 dst0, const uint8_t *__restrict__ z16, u64 n) {
    C25519_PRIO_CHAIN();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!(reinterpret_cast<const u32 *>(z16)[4 * i + 3] >> 31)) return;
    ge_aniels A = pts_load(pts, dst0 + i);
    feT t = fe_carry(fe_neg(A.xy2d));
    u32 w[32];
    for (int q = 0; q < 10; q++) { w[q] = A.ymx.v[q]; w[10 + q] = A.ypx.v[q]; w[20 + q] = t.v[q]; }
    w[30] = 0; w[31] = 0;
    uint4 *q4 = reinterpret_cast<uint4 *>(pts) + PTS_Q * (dst0 + i);
    for (int q = 0; q < PTS_Q; q++) q4[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
// scalars of the batch equation (batch.rs:213-233): msm_scalars[1+i] = |z_i|, [1+n+i] = z_i*h_i;
// per-block partial sums of z_i*s_i (mod l) to `partial` (ten 28-bit limbs each).  signed_z: z16 is sign-magnitude (device z-mode).
// Arithmetic: sc28.h -- radix 2^28, folding with l = 2^252 + c; per signature one 512-bit reduction (95 multiplier instructions)
// and two 5 x 10 limb products with their reductions (95 each), against ~1200 in the 5 x 52 Montgomery form of rounds 1-2.
__global__ void __launch_bounds__(256) k_batch_scalars(const uint8_t *__restrict__ hram, const uint8_t *__restrict__ sigs, const uint8_t *__restrict__ z16,
                                                       u64 n, int signed_z, uint8_t *__restrict__ msm_scalars, u32 *__restrict__ partial, int store_r = 1) {
    C25519_PRIO_CHAIN();
    __shared__ u32 red[256][10];
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    sc28 zs = sc28_zero();
    if (i < n) {
        const u32 *hw = reinterpret_cast<const u32 *>(hram) + 16 * i;
        u32 h16[16];
        for (int j = 0; j < 16; j++) h16[j] = hw[j];
        const u32 *zw = reinterpret_cast<const u32 *>(z16) + 4 * i;
        const bool neg = signed_z && (zw[3] >> 31);
        u32 zwords[8] = {zw[0], zw[1], zw[2], signed_z ? (zw[3] & 0x7fffffffu) : zw[3], 0, 0, 0, 0};
        u32 s[8], zl[5];
        load8(sigs, 2 * i + 1, s);
        sc28_limbs_from_words<4, 5>(zwords, zl);
        const sc28 h = sc28_from_wide(h16);
        zs = sc28_mul_5x10(zl, sc28_from_words(s).v);        // |z| s   (s < 2^256: a non-canonical s is reduced here and fails the batch through the flag)
        sc28 hz = sc28_mul_5x10(zl, h.v);                    // |z| h
        if (neg) { zs = sc28_neg(zs); hz = sc28_neg(hz); }   // z = -|z|
        u32 out[8];
        sc28_to_words(hz, out);
        store8(msm_scalars, 1 + n + i, out);
        if (store_r) store8(msm_scalars, 1 + i, zwords);      // (split batches: k_z_expand has written it, and the R half's sort may be reading it)
    }
    for (int j = 0; j < 10; j++) red[threadIdx.x][j] = zs.v[j];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sc28 a, b;
            for (int j = 0; j < 10; j++) { a.v[j] = red[threadIdx.x][j]; b.v[j] = red[threadIdx.x + off][j]; }
            a = sc28_add(a, b);
            for (int j = 0; j < 10; j++) red[threadIdx.x][j] = a.v[j];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) for (int j = 0; j < 10; j++) partial[(u64)blockIdx.x * 10 + j] = red[0][j];
}

// msm_scalars[0] = -(sum of the per-block partial sums) mod l: one block, strided sums then a tree
__global__ void __launch_bounds__(256) k_bsum_finish(const u32 *__restrict__ partial, u32 nblk, uint8_t *__restrict__ msm_scalars, u64 also_at = 0) {
    C25519_PRIO_CHAIN();
    __shared__ u32 red[256][10];
    sc28 acc = sc28_zero();
    for (u32 b = threadIdx.x; b < nblk; b += 256) {
        sc28 p;
        for (int j = 0; j < 10; j++) p.v[j] = partial[(u64)b * 10 + j];
        acc = sc28_add(acc, p);
    }
    for (int j = 0; j < 10; j++) red[threadIdx.x][j] = acc.v[j];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sc28 a, b;
            for (int j = 0; j < 10; j++) { a.v[j] = red[threadIdx.x][j]; b.v[j] = red[threadIdx.x + off][j]; }
            a = sc28_add(a, b);
            for (int j = 0; j < 10; j++) red[threadIdx.x][j] = a.v[j];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sc28 t;
        for (int j = 0; j < 10; j++) t.v[j] = red[0][j];
        u32 w[8];
        sc28_to_words(sc28_neg(t), w);
        store8(msm_scalars, 0, w);
        if (also_at) store8(msm_scalars, also_at, w);         // (split batches: B rides at the end of the A half)
    }
}
// msm_scalars[1 + i] = |z_i| zero-extended to 32 bytes (device z-mode: sign-magnitude z16): the R half's scalars, available as soon as the z_i are
__global__ void __launch_bounds__(256) k_z_expand(const uint8_t *__restrict__ z16, u64 n, uint8_t *__restrict__ msm_scalars) {
    C25519_PRIO_CHAIN();
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 *zw = reinterpret_cast<const u32 *>(z16) + 4 * i;
    const u32 w[8] = {zw[0], zw[1], zw[2], zw[3] & 0x7fffffffu, 0, 0, 0, 0};
    store8(msm_scalars, 1 + i, w);
}

hipError_t launch_hram(const uint8_t *msgs, const uint64_t *msg_off, uint64_t msgs_len, const uint8_t *sigs, const uint8_t *pks, uint64_t n, uint8_t *hram, uint32_t *flags, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hram, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, msgs, msg_off, msgs_len, sigs, pks, n, hram, flags);
    return hipGetLastError();
}
hipError_t launch_hram_dom(const uint8_t *dom, uint32_t dom_len, const uint8_t *msgs, const uint64_t *msg_off, uint64_t msgs_len, uint32_t fixed_len, const uint8_t *sigs, const uint8_t *pks,
                           uint64_t n, uint8_t *hram, uint32_t *flags, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_hram_dom, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dom, dom_len, msgs, msg_off, msgs_len, fixed_len, sigs, pks, n, hram, flags);
    return hipGetLastError();
}

}  // namespace c25519


void launch_apply_sign(uint32_t *pts, uint64_t dst0, const uint8_t *z16, uint64_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_apply_sign, dim3(div_up64(n, 256)), dim3(256), 0, st, pts, dst0, z16, n);
}

// ---- verify_batch ---------------------------------------------------------------------------------------
#include "transcript_host.h"

// states of the z tree: iv[l] = the BLAKE2b-256 state after the one-block tag of level l (see k_ztree_first)
static void ztree_make_ivs(uint64_t n, ztree_ivs &ivs) {
    uint64_t count = n;                                  // inputs of level 0: signatures
    for (int l = 0; l < ZTREE_MAX_LEVELS; l++) {
        if (l > 0 && count <= 1) { for (int q = 0; q < 8; q++) ivs.iv[l][q] = 0; continue; }      // (a level the tree does not have: its state is never read)
        const char tag[] = "c25519-hip/verify_batch/z-tree/v5";
        static_assert(sizeof(tag) - 1 <= 64, "tag fits the first half of the block");
        uint8_t blk[128] = {0};
        memcpy(blk, tag, sizeof(tag) - 1);
        u64 w[16];
        memcpy(w, blk, 128);                             // (little-endian host: the block's bytes as sixteen LE words)
        w[13] = (u64)l; w[14] = count; w[15] = n;        // level, number of inputs of this level, batch size
        blake2b_init(ivs.iv[l], 32);
        blake2b_compress(ivs.iv[l], w, 128, false);
        count = (count + 3) / 4;
    }
}
// the z_i of one pass (n signatures) by the device derivation; z16: room for 4 * ceil(n/4) entries.  t0 / t1: tree scratch
// ((n/4 + 1) * 32 bytes each).  Enqueued on `sa`.
static int32_t zchain_enqueue(c25519_ctx *ctx, hipStream_t sa, const uint8_t *hred, const uint8_t *d_sigs, uint64_t n, uint8_t *t0, uint8_t *t1, uint8_t *z16) {
    ztree_ivs ivs;
    ztree_make_ivs(n, ivs);
    uint64_t mm = (n + 3) / 4; uint8_t *a = t0, *b = t1;
    uint32_t level = 1;
    hipLaunchKernelGGL(k_ztree_first, dim3(div_up64(mm, 256)), dim3(256), 0, sa, hred, d_sigs, n, ivs, a);
    while (mm > 1024) {                                   // five levels per launch (4^5 = 1024 nodes per block)
        const uint64_t mo = (mm + 1023) / 1024;
        hipLaunchKernelGGL(k_ztree_block, dim3((unsigned)mo), dim3(256), 0, sa, a, mm, level, ivs, b, 5);
        mm = mo; level += 5; std::swap(a, b);
    }
    hipLaunchKernelGGL(k_ztree_block, dim3(1), dim3(256), 0, sa, a, mm, level, ivs, b, 64);      // the last <= 1024 nodes: leaves the 32-byte root at b
    hipLaunchKernelGGL(k_zderive, dim3(div_up64((n + 3) / 4, 256)), dim3(256), 0, sa, b, (n + 3) / 4, z16);
    HIPCHK(hipGetLastError());
    return C25519_OK;
}

// The same derivation on the HOST (the tree of k_ztree_first / k_ztree_block and k_zderive, word for word): for the small batches whose hashes the host
// computes anyway (verify_batch_small_host) -- a tree over <= 128 signatures is ~50 SHA-512 compressions.  hred: n x 32 bytes (h_i mod l, canonical);
// z16: room for 4 * ceil(n / 4) entries.  Equality with the device derivation: tests/test_gpu_verify.py (c25519_debug_batch_zs, z_mode 2 against 1).
static inline u64 host_le64(const uint8_t *p) { u64 v; memcpy(&v, p, 8); return v; }
static void ztree_host_zs(const uint8_t *hred, const uint8_t *sigs, uint64_t n, uint8_t *z16) {
    ztree_ivs ivs;
    ztree_make_ivs(n, ivs);
    uint64_t m = (n + 3) / 4;
    std::vector<u64> cur(4 * m), nxt;
    for (uint64_t j = 0; j < m; j++) {                       // level 0: (h_c mod l) || s_c of four signatures, two blocks
        u64 hs[8], rec[32];
        for (int q = 0; q < 8; q++) hs[q] = ivs.iv[0][q];
        for (int r = 0; r < 4; r++) {
            const uint64_t c = 4 * j + r;
            for (int q = 0; q < 4; q++) rec[8 * r + q] = c < n ? host_le64(hred + 32 * c + 8 * q) : 0ull;
            for (int q = 0; q < 4; q++) rec[8 * r + 4 + q] = c < n ? host_le64(sigs + 64 * c + 32 + 8 * q) : 0ull;
        }
        blake2b_compress(hs, rec, 256, false);
        blake2b_compress(hs, rec + 16, 384, true);
        for (int q = 0; q < 4; q++) cur[4 * j + q] = hs[q];
    }
    for (uint32_t level = 1; m > 1; level++) {               // upper levels: four children, one block
        const uint64_t mo = (m + 3) / 4;
        nxt.assign(4 * mo, 0);
        for (uint64_t j = 0; j < mo; j++) {
            u64 hs[8], w[16];
            for (int q = 0; q < 8; q++) hs[q] = ivs.iv[level][q];
            for (int ch = 0; ch < 4; ch++) for (int q = 0; q < 4; q++) w[4 * ch + q] = 4 * j + ch < m ? cur[4 * (4 * j + ch) + q] : 0ull;
            blake2b_compress(hs, w, 256, true);
            for (int q = 0; q < 4; q++) nxt[4 * j + q] = hs[q];
        }
        cur.swap(nxt); m = mo;
    }
    for (uint64_t i = 0; i < (n + 3) / 4; i++) {             // k_zderive: the four quarters of BLAKE2b-512(root || LE64(i))
        u64 hs[8], w[16];
        blake2b_init(hs, 64);
        for (int q = 0; q < 4; q++) w[q] = cur[q];
        w[4] = i;
        for (int q = 5; q < 16; q++) w[q] = 0;
        blake2b_compress(hs, w, 40, true);
        memcpy(z16 + 64 * i, hs, 64);
    }
}

// One random-linear-combination check over at most VERIFY_PASS_MAX signatures (an MSM of 2n+1 terms), enqueued on
// context ctx (the caller's or its peer); column sums and counters go to d_slot, nothing waits for the host.
// d_pk_points (may be NULL): the keys' decompressed points, n x 160 raw -- what VerifyingKey carries beside its bytes
// (verifying.rs:64-71), so that, like the reference (batch.rs:236), the batch does not decompress A_i again.
// d_hram_pre / d_z_pre (transcript z-mode): H(R||A||M) and the z_i of these signatures, computed over the whole batch.
// stage (host-pointer calls, may be null): called right before the first kernels that need an input array are enqueued, in the
// order 0 = signatures, 1 = key bytes, 2 = messages + offsets, 3 = the keys' points (only if given); it starts the upload of
// that array's slice for THIS pass on the copy stream and returns the event to wait for.  So R_i is being decompressed
// while the keys and messages travel, and the hash chain runs while the (five times larger) key points travel.
typedef std::function<int32_t(int what, hipEvent_t *ready)> verify_stage;
// (r6) device z-mode, inputs on the device, up to 2^16 signatures.  The ORDER in which the host enqueues the two chains (it needs ~4 us per launch or event, ~80 us
// for the whole call, and each chain can only run as far as it has been enqueued): 0 = k_hram, the three decompression launches, then tree / z_i / batch scalars;
// 1 = k_hram and the tree ahead of the decompression; 2 = the decompression ahead of everything.  A/B knob VERIFY_ORDER of the tuning build (-1 = the rule below); verify_pass_enqueue has the numbers.
// (last) The rule (knob -1): order 1 for key BYTES up to 2^14 signatures -- their one decompression launch of 2n lanes is enqueued in a moment, and behind the three launches
// of order 0 k_ztree_first started 20 us after the 18 us k_hram of a 4096-signature batch had ended (profiles/r06_timeline_mid_boundary_sizes.txt): 2048 .. 16 384
// signatures -3 .. -8 us in every pair of runs (profiles/r06_ab_verify_order_small.txt); order 0 with the keys' cached points (level, 16 384 signatures +8 us with order 1).
static int verify_order(uint64_t n, bool key_bytes) {
    static const int k = C25519_KNOB("VERIFY_ORDER", -1);
    if (n > (1ull << 16)) return 0;
    return k >= 0 ? k : (key_bytes && n <= (1ull << 14)) ? 1 : 0;
}
// ... and a single-pass batch whose 2n + 1-term MSM the mid path serves runs it on the hash chain's stream and publishes its record itself
static bool verify_on_chain(uint64_t n, const msm_geom &g, bool staged) {
    static const int k = C25519_KNOB("MID_ON_CHAIN", 1);     // A/B knob of the tuning build
    return k != 0 && !staged && n <= (1ull << 16) && msm_mid_serves(2 * n + 1, g, true);
}
// keys as BYTES: up to this many signatures A_i and R_i are decompressed by ONE launch of 2n lanes (one latency chain instead of two; rounds 3-5: 4096 -- A/B knob
// VERIFY_BOTH_MAX of the tuning build; profiles/r06_ab_verify_both.txt)
static uint64_t verify_both_max() { static const uint64_t v = (uint64_t)C25519_KNOB_LL("VERIFY_BOTH_MAX", 1 << 16); return v; }
// pre (small batches of the transcript z-mode, may be null): the records of A_i and R_i are ALREADY at their place in the context's record buffer
// (or will be once `ready` has fired) -- decompressed on the second stream while the hashes went to the host and the z_i came back -- with
// cnt[0] keys and cnt[1] R_i that do not decode
struct verify_pre { hipEvent_t ready; const uint32_t *cnt; };
__global__ void k_add_point_counters(u32 *__restrict__ d_cnt, const u32 *__restrict__ cnt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { d_cnt[2] += cnt[0]; d_cnt[3] += cnt[1]; }
}
static int32_t verify_pass_enqueue(c25519_ctx *owner, c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                   const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, uint64_t n, uint32_t z_mode,
                                   const uint8_t *d_hram_pre, const uint8_t *d_z_pre, const uint32_t *d_pre_flags, const msm_geom &g, uint64_t terms, uint32_t *d_slot, hipEvent_t wait_acc,
                                   const verify_stage *stage = nullptr, const verify_pre *pre = nullptr, c25519_ctx *split_peer = nullptr, uint32_t *d_slot_b = nullptr) {
    hipStream_t st = ctx->stream;
    const uint64_t m = 2 * n + 1;
    int32_t r;
    const bool split = split_peer != nullptr && d_slot_b != nullptr && !d_z_pre && !stage && !pre;      // (r6) the batch in two halves: see the end of this function
    if ((r = ctx_reserve(ctx, ctx->tmp_e, (m + 1) * PTS_BYTES + 256))) return r;      // (+ 1: split batches keep a second copy of B's record behind the keys')
    // tmp_f: hram (64n) | z16 (16n) | msm scalars (32m) | tree scratch | partial sums
    const unsigned nblk = div_up64(n, 256);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t oH = carve(n * 64), oZ = carve((n + 4) * 16), oSc = carve((m + 1) * 32), oT0 = carve((n / 4 + 2) * 32), oT1 = carve((n / 4 + 2) * 32), oP = carve((size_t)nblk * 40);
    const size_t oHr = carve(d_z_pre ? 0 : n * 32);       // h_i mod l, for the device z-tree
    if ((r = ctx_reserve(ctx, ctx->tmp_f, off))) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_f.p;
    uint8_t *hram = ws + oH, *z16 = ws + oZ, *msc = ws + oSc, *t0 = ws + oT0, *t1 = ws + oT1, *hred = d_z_pre ? nullptr : ws + oHr;
    uint32_t *partial = (uint32_t *)(ws + oP);
    uint32_t *d_pts = (uint32_t *)ctx->tmp_e.p;
    uint32_t *d_cnt = slot_flags(d_slot);             // [2] bad A, [3] bad R, [4] bad s, [5] bad offsets
    hipEvent_t *ring = pass_ring(owner, ctx, 2);
    HIPCHK(hipEventRecord(ring[3], st));
    slot_init(d_slot, terms, d_pre_flags, st, g.c);
    // Two independent chains: (S) decompress R_i and A_i -- VALU-bound; (A) hash, derive z_i, batch scalars, sort --
    // partly latency-bound (the tree levels).  They run on two streams and join before the accumulation.
    hipStream_t sa = ctx->aux;
    HIPCHK(hipEventRecord(ctx->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(sa, ctx->ev_fork, 0));
    hipEvent_t ev_sig = nullptr, ev_pk = nullptr, ev_msg = nullptr, ev_pts = nullptr;
    // (r5) The hash chain is the longer of the two (1.5 against 1.4 ms at 2^20 signatures, everything at 2^14): its first kernel is ENQUEUED first.  The host
    // needs 5 - 8 us per launch, and with the three decompression launches ahead of it k_hram started 40 us into the call (gpurun_out/r05_timeline_verify_*).
    // (Host-pointer calls keep the arrival order of their inputs.)
    const uint8_t *hr = d_hram_pre;
    static const int hram_first_knob = C25519_KNOB("HRAM_FIRST", 1);      // A/B knob: 0 = behind the decompression launches (rounds 1-4)
    // (up to 2^18 signatures: 0.594 -> 0.581 ms at 2^14, 1.034 -> 1.018 at 2^18; at 2^20 the hash kernels then take the compute units ahead of the decompression
    //  and the call is 0.6 % slower: profiles/r05_ab_midrange_streams.txt)
    const int order = (!stage && !hr && !split) ? verify_order(n, d_pk_points == nullptr) : 0;
    const bool hram_first = hram_first_knob && !stage && !hr && n <= (1ull << 18) && order != 2;
    if (hram_first) { hipLaunchKernelGGL(k_hram, dim3(nblk), dim3(256), 0, sa, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, hram, d_cnt + 4, hred); hr = hram; }
    // (r6, late) order 1: the tree and the z_i ahead of the decompression launches as well.  With SHA-512 in the tree (until call 30 of round 6) the chain was the longer
    // one up to 2^16 signatures and k_ztree_first, enqueued behind the three decompression launches, started 20 us after k_hram had ended
    // (profiles/r06_timeline_mid_verify_2p14.txt); with BLAKE2b it is the shorter one and the decompression must not start 60 us into the call.
    const bool chain_first = hram_first && order == 1 && !d_z_pre;
    // A batch whose MSM takes the mid path (mid.hip) runs that MSM on THIS stream, right behind its scalars: the digits and the sort need nothing else; the records
    // (main stream) are waited for once, in front of the accumulation, and the sign of z_i is applied to R_i there (msm_mid_enqueue, mid_run).  Before: k_bsum_finish ->
    // 30 us (event, k_apply_sign on the main stream, event) -> k_mid_front at 2^14 signatures, plus an event record between k_zderive and k_batch_scalars.
    const bool on_chain = !stage && !d_hram_pre && !d_z_pre && !split && ctx->solo && !wait_acc && !pre && verify_on_chain(n, g, false);
    if (chain_first) {
        if ((r = zchain_enqueue(ctx, sa, hred, d_sigs, n, t0, t1, z16))) return r;
        if (!on_chain) HIPCHK(hipEventRecord(ctx->ev_z, sa));
    }
    // (S) points: [0] = B, [1..n] = R_i, [n+1..2n] = A_i     (batch.rs:235-244)
    launch_prep_basepoint(d_pts, 0, st);
    auto prep_A = [&]() -> int32_t {
        if (d_pk_points) {
            if (stage) { int32_t q = (*stage)(3, &ev_pts); if (q) return q; HIPCHK(hipStreamWaitEvent(st, ev_pts, 0)); }
            return prep_points(ctx, d_pk_points, n, C25519_FMT_RAW160, d_pts, n + 1, d_cnt + 2, true);      // (the keys' cached points: Z = 1 unless the caller made them otherwise)
        }
        HIPCHK(launch_prep_compressed(0, d_pks, 1, n, d_pts, n + 1, d_cnt + 2, true, st));
        return C25519_OK;
    };
    auto prep_R = [&]() -> int32_t {   // R_i = the first half of every 64-byte signature (stride 2)
        HIPCHK(hipEventRecord(ring[4], st));
        ctx->kname[1] = "c25519::k_prep_compressed<0> (decompression of R_i)";
        HIPCHK(launch_prep_compressed(0, d_sigs, 2, n, d_pts, 1, d_cnt + 3, true, st));
        HIPCHK(hipEventRecord(ring[5], st));
        return C25519_OK;
    };
    if (stage) {
        // host-pointer call: in the order the inputs arrive -- signatures, then R_i decompresses while keys and messages travel
        if ((r = (*stage)(0, &ev_sig))) return r;
        HIPCHK(hipStreamWaitEvent(st, ev_sig, 0));
        if ((r = prep_R())) return r;
        if ((r = (*stage)(1, &ev_pk)) || (r = (*stage)(2, &ev_msg))) return r;
        HIPCHK(hipStreamWaitEvent(sa, ev_sig, 0)); HIPCHK(hipStreamWaitEvent(sa, ev_pk, 0)); HIPCHK(hipStreamWaitEvent(sa, ev_msg, 0));
        if (!d_pk_points) { HIPCHK(hipStreamWaitEvent(st, ev_pk, 0)); if ((r = prep_A())) return r; }
    } else if (pre) {
        HIPCHK(hipStreamWaitEvent(st, pre->ready, 0));
        hipLaunchKernelGGL(k_add_point_counters, dim3(1), dim3(64), 0, st, d_cnt, pre->cnt);
        HIPCHK(hipEventRecord(ring[4], st)); HIPCHK(hipEventRecord(ring[5], st));
    } else if (!d_pk_points && n <= verify_both_max()) {
        // small batches of key BYTES: both decompressions in one launch (one latency chain instead of two; d_cnt[2] = bad A, d_cnt[3] = bad R)
        HIPCHK(hipEventRecord(ring[4], st));
        ctx->kname[1] = "c25519::k_prep_compressed_keys_and_r (decompression of A_i and R_i)";
        HIPCHK(launch_prep_compressed_keys_and_r(d_pks, d_sigs, n, d_pts, d_cnt + 2, st));
        HIPCHK(hipEventRecord(ring[5], st));
    } else {
        if ((r = prep_A())) return r;
        if (split) {                                       // B once more, behind the keys' records: the A half is terms n + 1 .. 2n + 1
            launch_prep_basepoint(d_pts, 2 * n + 1, st);
            HIPCHK(hipEventRecord(ctx->ev_split, st));
        }
        if ((r = prep_R())) return r;
    }
    // (A)
    if (hram_first) { }
    else if (!hr) { hipLaunchKernelGGL(k_hram, dim3(nblk), dim3(256), 0, sa, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, hram, d_cnt + 4, hred); hr = hram; }
    else if (hred) hipLaunchKernelGGL(k_hram_mod_l, dim3(nblk), dim3(256), 0, sa, hr, n, hred);
    HIPCHK(hipGetLastError());
    const uint8_t *zz = d_z_pre;
    if (!zz) {
        if (!chain_first) {
            if ((r = zchain_enqueue(ctx, sa, hred, d_sigs, n, t0, t1, z16))) return r;
            if (!on_chain) HIPCHK(hipEventRecord(ctx->ev_z, sa));
        }
        zz = z16;
        // the sign of z_i goes onto the stored R_i (main stream, beside the sort on the second one)
        if (!on_chain) {
            HIPCHK(hipStreamWaitEvent(st, ctx->ev_z, 0));
            hipLaunchKernelGGL(k_apply_sign, dim3(nblk), dim3(256), 0, st, d_pts, (uint64_t)1, zz, n);
        }
    }
    // (r6) TWO HALVES (round-5 verdict, item 6).  The 2n + 1 terms are two different halves: the R half's scalars are the |z_i| themselves (128 bits: 8 windows) and exist
    // when k_zderive ends; the A half's are z_i h_i mod l (253 bits) and exist only after k_batch_scalars / k_bsum_finish.  As ONE pass the sort of everything waits for the
    // last scalar and the accumulation for the whole sort (k_accumulate started 1.56 ms into a 2.61 ms call at 2^20 signatures: profiles/r05_verify_timeline.txt).  As two
    // passes on the two stream sets -- the R half here, the A half (with B behind it) on the peer context -- the R half's sort and accumulation run while the A half's
    // scalars and sort are still being made; the accumulations follow each other, each half reduces its own buckets (same layout), and the two column-sum slots are added
    // before the one identity check (k_record_sum).  Device z-mode, single-pass batches of the bucket pipeline only.
    // MEASURED AND NOT ADOPTED (profiles/r06_ab_verify_split.txt, bit-exact in the full verify / multi / ffi modules): 2^20 signatures 2.80 against 2.60 ms, 2^19 1.66
    // against 1.50.  The premise does not hold: the R half cannot accumulate before R is decompressed (1.37 ms into the call: 0.33 ms of key normalisation, then 0.97 ms
    // of decompression on the main stream), its sort beside that decompression takes 0.53 instead of 0.25 ms, and two accumulations + two reductions cost two ramp-downs.
    // The arm stays behind VERIFY_SPLIT_MIN of the tuning build (default 0 = never) for one round.
    if (split) {
        c25519_ctx *pc = split_peer;
        hipStream_t pa = pc->aux, ps = pc->stream;
        // R half: |z_i| right behind k_zderive on the second stream; the sort follows there, the accumulation on the main stream behind the decompression of R
        // (ev_z was recorded above, BEFORE the expansion: the sign application needs only z16; record it again behind the expansion for the A chain)
        hipLaunchKernelGGL(k_z_expand, dim3(nblk), dim3(256), 0, sa, zz, n, msc);
        HIPCHK(hipEventRecord(ctx->ev_z, sa));
        // A chain on the PEER's second stream: it must not queue behind the R half's long-bucket kernels, which wait for the main stream
        HIPCHK(hipStreamWaitEvent(pa, ctx->ev_z, 0));
        hipLaunchKernelGGL(k_batch_scalars, dim3(nblk), dim3(256), 0, pa, hr, d_sigs, zz, n, 1, msc, partial, 0);
        hipLaunchKernelGGL(k_bsum_finish, dim3(1), dim3(256), 0, pa, partial, nblk, msc, (u64)(2 * n + 1));
        HIPCHK(hipGetLastError());
        // the peer's main stream: behind the keys' records and B's second copy (ev_split, recorded by prep below / above) and behind the R half's accumulation (wait_acc)
        HIPCHK(hipStreamWaitEvent(ps, ctx->ev_split, 0));
        hipEvent_t *ring_b = pass_ring(owner, pc, 1);
        HIPCHK(hipEventRecord(ring_b[3], ps));
        slot_init(d_slot_b, terms, nullptr, ps, g.c);
        ctx->solo = false; pc->solo = false;
        if ((r = msm_enqueue(ctx, msc + 32, n, d_pts + (size_t)1 * (PTS_BYTES / 4), g, d_slot, ring, sa, wait_acc))) return r;
        if ((r = msm_enqueue(pc, msc + 32 * (n + 1), n + 1, d_pts + (size_t)(n + 1) * (PTS_BYTES / 4), g, d_slot_b, ring_b, pa, ctx->ev_acc))) { if (ctx->err.empty()) ctx->err = pc->err; return r; }
        // join: the caller's stream continues behind the peer's half; the two slots become one
        HIPCHK(hipEventRecord(pc->ev_in, ps));
        HIPCHK(hipStreamWaitEvent(st, pc->ev_in, 0));
        launch_record_sum(d_slot, d_slot, 2, g.nwin, 1, st);
        HIPCHK(hipGetLastError());
        return C25519_OK;
    }
    hipLaunchKernelGGL(k_batch_scalars, dim3(nblk), dim3(256), 0, sa, hr, d_sigs, zz, n, z_mode == C25519_Z_DEVICE ? 1 : 0, msc, partial);
    // the basepoint coefficient -sum z_i s_i (batch.rs:240): the per-block partial sums are folded by one more block
    hipLaunchKernelGGL(k_bsum_finish, dim3(1), dim3(256), 0, sa, partial, nblk, msc);
    HIPCHK(hipGetLastError());
    if (on_chain) {
        HIPCHK(hipEventRecord(ctx->ev_pts, st));                       // the records of B, R_i and A_i
        const mid_run run = {sa, ctx->ev_pts, z16, 1, n};
        return msm_enqueue(ctx, msc, m, d_pts, g, d_slot, ring, sa, wait_acc, &run);
    }
    if (stage && d_pk_points && (r = prep_A())) return r;   // the keys' points come last: only the accumulation needs them
    // the MSM's digit/sort phase continues on the second stream while (S) is still decompressing
    return msm_enqueue(ctx, msc, m, d_pts, g, d_slot, ring, sa, wait_acc);
}
// Batches beyond ~1.5 * 2^20 signatures are checked as several independent random linear combinations of about
// 2^20 signatures each (same reason as MSM_PASS_MAX; in the device z-mode every pass derives its own z_i from its own
// tree; in the transcript z-mode the z_i come from ONE transcript over the whole batch, exactly the reference's).  Every
// pass keeps its own identity check.  All passes run even after a failure so that the reference's precedence -- key
// decoding, then ScalarFormat for ANY non-canonical s (batch.rs:208-211), then Verify -- does not depend on where the
// batch was cut.
static const int VERIFY_PASS_LOG2 = [] { int v = C25519_KNOB("VERIFY_PASS_LOG2", 20); return v < 15 ? 15 : (v > 21 ? 21 : v); }();   // A/B knob
static const uint64_t VERIFY_PASS = 1ull << VERIFY_PASS_LOG2, VERIFY_PASS_MAX = 3ull << (VERIFY_PASS_LOG2 - 1);
// flags of a folded verify_batch record + its point -> the reference's verdict (precedence: key decoding, then ScalarFormat
// for ANY non-canonical s, batch.rs:208-211, then Verify, :244-250)
static int32_t verify_record_verdict(c25519_ctx *ctx, const ge_p3 &R, const uint32_t flags[8]) {
    if (flags[5]) { if (ctx) ctx->err = "verify_batch: msg_off is not monotone or runs past msgs_len"; return -(int32_t)hipErrorInvalidValue; }
    if (flags[0]) { if (ctx) ctx->err = "verify_batch: internal error (batch scalar with bit 255 set)"; return -(int32_t)hipErrorInvalidValue; }
    if (flags[2]) return C25519_NONE;                       // a key that VerifyingKey::from_bytes rejects
    if (flags[4]) return C25519_SCALAR_FORMAT;
    if (flags[3]) return C25519_VERIFY;                     // batch.rs:244 (an R that fails to decompress)
    return ge_is_identity(R) ? C25519_OK : C25519_VERIFY;   // batch.rs:246-250
}
// Every pass of a batch whose z_i are GIVEN (transcript z-mode: d_hram = H(R||A||M) of these n signatures followed by a
// 64-byte trailer of counters -- [0] non-canonical s, [1] bad offsets -- as ed25519_batch_hram_dev leaves them; d_z16 = their
// z_i), summed into ONE record at d_record: the reference's single equation (batch.rs:235-250) whatever the pass split.
static int32_t verify_record_enqueue(c25519_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, const uint8_t *d_hram, const uint8_t *d_z16,
                                     uint64_t n, uint32_t *d_record, const verify_pre *pre = nullptr) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n >= (1ull << 40)) { ctx->err = "verify_batch: n too large"; return -(int32_t)hipErrorInvalidValue; }
    const uint32_t *d_pre = (const uint32_t *)(d_hram + n * 64);
    if (n == 0) { ctx->last_passes.clear(); slot_init(d_record, 0, d_pre, ctx->stream, 0); HIPCHK(hipGetLastError()); return C25519_OK; }
    const uint64_t passes = n <= VERIFY_PASS_MAX ? 1 : (n + VERIFY_PASS - 1) / VERIFY_PASS, per = (n + passes - 1) / passes;
    msm_geom g;
    msm_layout(2 * per + 1, g, 16);        // (the z_i are 128-bit: with 16-bit windows they end on a window boundary; a 17-bit layout leaves a 9-bit stub of 2^20 equal-ish digits)
    int32_t r;
    pass_set ps;
    if ((r = passes_begin(ctx, passes, ps))) return r;
    ctx->solo = passes == 1;               // one pass on this context alone: its reduction runs on the main stream (msm.hip msm_enqueue_acc)
    hipEvent_t prev_acc = nullptr;
    for (uint64_t p0 = 0; p0 < passes; p0 += C25519_MAX_SLOTS) {
        const int cnt = (int)std::min<uint64_t>(C25519_MAX_SLOTS, passes - p0);
        if (p0 && ps.lanes > 1) {
            HIPCHK(hipEventRecord(ctx->ev_in, ctx->stream));
            for (int l = 1; l < ps.lanes; l++) HIPCHK(hipStreamWaitEvent(ps.c[l]->stream, ctx->ev_in, 0));
        }
        for (int i = 0; i < cnt; i++) {
            const uint64_t lo = (p0 + i) * per, m = std::min(per, n - lo);
            c25519_ctx *c = ps.c[(p0 + i) % ps.lanes];
            uint32_t *slot = passes == 1 ? d_record : dslot(ctx, i);
            r = verify_pass_enqueue(ctx, c, nullptr, nullptr, 0, d_sigs + lo * 64, d_pks + lo * 32, d_pk_points ? d_pk_points + lo * 160 : nullptr, m, C25519_Z_TRANSCRIPT,
                                    d_hram + lo * 64, d_z16 + lo * 16, (p0 + i == 0) ? d_pre : nullptr, g, 2 * per + 1, slot, prev_acc, nullptr, passes == 1 ? pre : nullptr);
            if (r) { if (ctx->err.empty()) ctx->err = c->err; return r; }
            prev_acc = ps.lanes > 1 ? c->ev_acc : nullptr;
        }
        if ((r = passes_join(ctx, ps))) return r;
        if (passes > 1) launch_record_sum(d_record, ctx->d_slots, cnt, g.nwin, p0 == 0 ? 1 : 0, ctx->stream);
    }
    HIPCHK(hipGetLastError());
    return C25519_OK;
}
// H(R_i || A_i || M_i) of n signatures to d_hram (n x 64 bytes) followed by a 64-byte trailer of counters ([0] signatures
// with a non-canonical s, [1] bad message offsets): the per-signature half of the transcript z-mode, enqueue only.
static int32_t batch_hram_enqueue(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len, const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, uint8_t *d_hram) {
    HIPCHK(hipSetDevice(ctx->device));
    uint32_t *fl = (uint32_t *)(d_hram + n * 64);
    HIPCHK(hipMemsetAsync(fl, 0, 64, ctx->stream));
    HIPCHK(launch_hram(d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, d_hram, fl, ctx->stream));
    return C25519_OK;
}
EXPORT int32_t ed25519_batch_hram_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len, const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n,
                                      uint8_t *d_hram) {
    return batch_hram_enqueue(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, d_hram);
}
// the reference's z_i from the bytes its transcript absorbs (batch.rs:168-222): host arithmetic, no context, sequential
EXPORT int32_t ed25519_batch_transcript_zs(const uint8_t *hram, const uint8_t *sigs, uint64_t n, uint8_t *z16) {
    c25519_transcript_zs(hram, sigs, n, z16);
    return C25519_OK;
}
EXPORT int32_t ed25519_verify_batch_record_dev(c25519_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, const uint8_t *d_hram, const uint8_t *d_z16,
                                               uint64_t n, uint8_t *d_record) {
    return verify_record_enqueue(ctx, d_sigs, d_pks, d_pk_points, d_hram, d_z16, n, (uint32_t *)d_record);
}
EXPORT int32_t ed25519_fold_verify_records(c25519_ctx *ctx, const uint8_t *records, uint64_t count) {
    ge_p3 R;
    uint32_t flags[8];
    int32_t r = records_fold(records, count, R, flags, ctx ? &ctx->err : nullptr);
    if (r) return r;
    return verify_record_verdict(ctx, R, flags);
}

// pass_stage (host-pointer calls, device z-mode; may be null): (first signature of the pass, its length, which array, event out)
typedef std::function<int32_t(uint64_t lo, uint64_t m, int what, hipEvent_t *ready)> verify_fetch;
static int32_t verify_batch_impl(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                 const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, uint64_t n, uint32_t z_mode, const verify_fetch *fetch) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;                      // batch.rs: 1-term MSM 0*B = identity
    if (n >= (1ull << 40)) { ctx->err = "verify_batch: n too large"; return -(int32_t)hipErrorInvalidValue; }
    if (z_mode > 1) { ctx->err = "verify_batch: bad z_mode"; return -(int32_t)hipErrorInvalidValue; }
    if (!fetch) ctx->host_us[0] = ctx->host_us[1] = wall_us();      // (c25519_last_call_host_us: entered; the collect functions stamp "enqueued" and "results on the host")
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    int32_t r;
    if (z_mode == C25519_Z_TRANSCRIPT) {
        // the reference's sequential Merlin transcript (batch.rs:168-222) over the WHOLE batch, on one host core; then ONE
        // equation over the whole batch (the passes' column sums are added on the device), exactly batch.rs:235-250
        try {
            if ((r = ctx_reserve(ctx, ctx->tmp_c2, n * 80 + 128))) return r;
            uint8_t *d_hram_all = (uint8_t *)ctx->tmp_c2.p, *d_z_all = d_hram_all + n * 64 + 64;
            // small batches of key bytes: A_i and R_i are decompressed on the second stream NOW -- a ~70 us latency chain that needs none of what follows --
            // while the hashes go to the host, the transcript runs and the z_i come back (about as long at these sizes); the pass then finds its records ready
            verify_pre pre_pts = {nullptr, nullptr};
            if (!d_pk_points && n <= 4096) {
                if ((r = ctx_reserve(ctx, ctx->tmp_e, (2 * n + 1) * PTS_BYTES + 256))) return r;
                uint32_t *cnt = (uint32_t *)ctx->d_flag + 44;
                HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));                 // (the inputs are on the device once the main stream gets here)
                HIPCHK(hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
                HIPCHK(hipMemsetAsync(cnt, 0, 8, ctx->aux));
                HIPCHK(launch_prep_compressed_keys_and_r(d_pks, d_sigs, n, (uint32_t *)ctx->tmp_e.p, cnt, ctx->aux));
                HIPCHK(hipEventRecord(ctx->ev_pts, ctx->aux));
                pre_pts.ready = ctx->ev_pts; pre_pts.cnt = cnt;
            }
            if ((r = batch_hram_enqueue(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, n, d_hram_all))) return r;
            // (r4) the host copies live in a page-locked buffer the context keeps instead of three fresh std::vectors per call (144 MB at 2^20
            // signatures).  Measured neutral (468 ms per 2^20 signatures either way): the call IS the sponge -- ~410 ns per signature inside the
            // library on the GPU box's host (1.73 permutations of ~185 ns + framing), tools/keccak_bench.cpp, tools/strict_rate.py
            if ((r = ctx_host_stage(ctx, n * 144 + 64))) return r;
            uint8_t *hh = (uint8_t *)ctx->h_stage, *hs = hh + n * 64, *hz = hs + n * 64;
            HIPCHK(hipMemcpyAsync(hh, d_hram_all, n * 64, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(hs, d_sigs, n * 64, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            c25519_transcript_zs(hh, hs, n, hz);
            HIPCHK(hipMemcpyAsync(d_z_all, hz, n * 16, hipMemcpyHostToDevice, ctx->stream));
            if ((r = verify_record_enqueue(ctx, d_sigs, d_pks, d_pk_points, d_hram_all, d_z_all, n, drec(ctx), pre_pts.ready ? &pre_pts : nullptr))) return r;
        } catch (const std::exception &e) { ctx->err = std::string("verify_batch: ") + e.what(); return -(int32_t)hipErrorOutOfMemory; }
        if (n >= (1ull << 16)) ctx->coarse_wait = ctx->ev_acc;          // a long call blocks on its accumulation before it polls for the published record (msm.hip publish_and_wait)
        if ((r = rec_collect(ctx))) return r;
        HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
        ge_p3 R;
        uint32_t flags[8];
        if ((r = records_fold((const uint8_t *)hslot(ctx, C25519_MAX_SLOTS), 1, R, flags, &ctx->err))) return r;
        return verify_record_verdict(ctx, R, flags);
    }
    // device z-mode: every pass derives its own z_i from its own tree and is its own random linear combination (summing
    // passes with independent z_i would open a 2^126 birthday attack across passes); all passes run even after a failure so
    // that the precedence does not depend on where the batch was cut
    const uint64_t passes = n <= VERIFY_PASS_MAX ? 1 : (n + VERIFY_PASS - 1) / VERIFY_PASS, per = (n + passes - 1) / passes;
    msm_geom g;
    // (r6, late) the window width of a mid-size batch: the MSM's rule (msm.hip pick_window) was tuned on 253-bit scalars; here half of the terms carry 127-bit ones
    // (nine windows instead of nineteen at 14 bits) and 16-bit windows end exactly on the z_i (no top window of a few bits whose lists are over-long).  Measured
    // (profiles/r06_ab_verify_window.txt): 2^15 signatures 14 bits 0.411 against the rule's 15 0.417 ms; 2^16: 16 bits 0.476 against 15 bits 0.501; 2^13 / 2^14: the rule (13 / 14).
    static const int verify_c = C25519_KNOB("VERIFY_C", 0);      // A/B knob of the tuning build: 0 = this rule; 8 .. 16 = that width; -1 = the MSM's rule
    const uint64_t vterms = 2 * per + 1;
    int vc = 0;
    // (late) ... and 12 bits below 8192 terms, where the mid path takes over from the small one at 2048 signatures (profiles/r06_ab_small_mid_boundary.txt: 3072 signatures 12 bits
    // 0.284, 13 bits 0.293, 11 bits 0.289 ms)
    if (verify_c >= 8 && verify_c <= 16 && vterms > verify_small_max()) vc = verify_c;
    else if (verify_c == 0 && passes == 1 && msm_mid_serves_terms(vterms)) vc = vterms < 8192 ? 12 : vterms < 24576 ? 13 : vterms < 98304 ? 14 : 16;
    msm_layout(vterms, g, 16, vc);        // (the z_i are 128-bit: with 16-bit windows they end on a window boundary; a 17-bit layout leaves a 9-bit stub of 2^20 equal-ish digits)
    pass_set ps;
    if ((r = passes_begin(ctx, passes, ps))) return r;
    ctx->solo = passes == 1;
    // (r6) a single pass through the mid path, inputs on the device: the last reduction block publishes the record -- columns and the slot's counters -- into the
    // context's page-locked host slot and the host polls the sequence word (msm.hip wait_published, with the small path's recovery: a lost publication re-runs the
    // batch once through the slot + copy path) instead of a copy-engine launch and a blocking synchronisation behind the last kernel
    static const int verify_direct_knob = C25519_KNOB("VERIFY_DIRECT", 1);
    const bool direct = passes == 1 && !fetch && verify_direct_knob && !ctx->no_direct_once && ps.c[0] == ctx && verify_on_chain(n, g, false);
    ctx->no_direct_once = false;
    ctx->direct_seq = 0;
    if (direct) { ctx->direct_seq = ++ctx->publish_seq; if (!ctx->direct_seq) ctx->direct_seq = ++ctx->publish_seq; }
    bool seen[5] = {false, false, false, false, false}, bad_off = false, bad_scalar = false;
    hipEvent_t prev_acc = nullptr;
    for (uint64_t p0 = 0; p0 < passes; p0 += C25519_MAX_SLOTS) {
        const int cnt = (int)std::min<uint64_t>(C25519_MAX_SLOTS, passes - p0);
        for (int i = 0; i < cnt; i++) {
            const uint64_t lo = (p0 + i) * per, m = std::min(per, n - lo);
            c25519_ctx *c = ps.c[(p0 + i) % ps.lanes];
            const verify_stage stage = [&](int what, hipEvent_t *ready) -> int32_t { return (*fetch)(lo, m, what, ready); };
            // (r6) a single-pass batch of the bucket pipeline in two halves on the two stream sets (verify_pass_enqueue; A/B knob VERIFY_SPLIT_MIN, 0 = never)
            static const uint64_t split_min = (uint64_t)C25519_KNOB_LL("VERIFY_SPLIT_MIN", 0);      // MEASURED AND NOT ADOPTED (profiles/r06_ab_verify_split.txt): 2.80 against 2.60 ms at 2^20
            c25519_ctx *split_peer = (passes == 1 && !fetch && split_min && m >= split_min && !msm_mid_serves(2 * m + 1, g, true) && 2 * m + 1 > verify_small_max()) ? ctx_peer(ctx) : nullptr;
            r = verify_pass_enqueue(ctx, c, d_msgs, d_msg_off + lo, msgs_len, d_sigs + lo * 64, d_pks + lo * 32, d_pk_points ? d_pk_points + lo * 160 : nullptr, m, z_mode,
                                    nullptr, nullptr, nullptr, g, 2 * per + 1, dslot(ctx, i), prev_acc, fetch ? &stage : nullptr, nullptr, split_peer, split_peer ? dslot(ctx, 1) : nullptr);
            if (r) { ctx->direct_seq = 0; if (ctx->err.empty()) ctx->err = c->err; return r; }
            prev_acc = ps.lanes > 1 ? c->ev_acc : nullptr;
            if (n >= (1ull << 16) && !direct) ctx->coarse_wait = c->ev_acc;
        }
        if ((r = passes_join(ctx, ps))) { ctx->direct_seq = 0; return r; }
        if (direct) {
            r = rec_collect(ctx);                           // (polls for ctx->direct_seq and clears it)
            if (r == C25519_LOST_PUBLICATION) {             // never observed (profiles/r06_soak_small.txt); tests/test_gpu_verify.py injects it
                ctx->no_direct_once = true;
                const int32_t r2 = verify_batch_impl(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, d_pk_points, n, z_mode, fetch);
                if (r2 >= 0) ctx->err.clear();              // (the note of the lost publication: the call has its verdict)
                return r2;
            }
            if (r) return r;
        } else if ((r = slots_collect(ctx, cnt))) return r;
        for (int i = 0; i < cnt; i++) {
            const uint32_t *s = hslot(ctx, direct ? C25519_MAX_SLOTS : i), *f = s + MSM_MAX_WIN * 40;
            if (f[0]) bad_scalar = true;
            if (f[5]) bad_off = true;
            uint32_t fl[8] = {0, 0, f[2], f[3], f[4], 0, 0, 0};
            const bool clean = !(f[2] | f[3] | f[4]);
            seen[verify_record_verdict(nullptr, clean ? msm_horner(s, g) : ge_identity(), fl)] = true;
        }
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    if (!fetch) ctx->host_us[4] = wall_us();
    if (bad_off) { ctx->err = "verify_batch: msg_off is not monotone or runs past msgs_len"; return -(int32_t)hipErrorInvalidValue; }
    if (bad_scalar) { ctx->err = "verify_batch: internal error (batch scalar with bit 255 set)"; return -(int32_t)hipErrorInvalidValue; }
    return seen[C25519_NONE] ? C25519_NONE : seen[C25519_SCALAR_FORMAT] ? C25519_SCALAR_FORMAT : seen[C25519_VERIFY] ? C25519_VERIFY : C25519_OK;
}

EXPORT int32_t ed25519_verify_batch_keys_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                             const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, uint64_t n, uint32_t z_mode) {
    return verify_batch_impl(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, d_pk_points, n, z_mode, nullptr);
}
EXPORT int32_t ed25519_verify_batch_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                        const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, uint32_t z_mode) {
    return ed25519_verify_batch_keys_dev(ctx, d_msgs, d_msg_off, msgs_len, d_sigs, d_pks, nullptr, n, z_mode);
}
// host-side check of the offsets array (the _dev entry points check on the device, inside k_hram)
static bool offsets_ok(const uint64_t *msg_off, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) if (msg_off[i] > msg_off[i + 1]) return false;
    return true;
}
// (r5) Small batches (keys as bytes or with their cached points; either z-mode), host pointers (the reference's own benchmark sizes, ed25519_benchmarks.rs:53): the host does what is
// sequential or tiny anyway, the GPU does the curve arithmetic, and nothing crosses the link twice.
//   rounds 3-4: upload -> k_hram -> hashes to the host -> transcript -> z_i back -> k_batch_scalars -> k_bsum_finish -> small MSM -> record copy: ten launches, two
//   round trips, 236 us for 4 signatures of which the GPU was busy 205 (profiles/r05_small_call_phases.txt).
//   now: ONE decompression launch (k_prep_small_verify: A_i, R_i and B into records; reads the page-locked staging buffer in place) -- and while its ~70 us chain
//   of 252 squarings runs, the host hashes R || A || M (batch.rs:185-189), checks the s_i (signature.rs:89-94), runs the transcript (batch.rs:195-222) and forms
//   the 2n + 1 scalars (batch.rs:225-240; sc_sha.h, the reference's Scalar52 arithmetic) into the same buffer; then the small MSM (small.hip) reads them in place
//   and publishes its record, with the decode counters of the first kernel, straight into host memory (small_direct).  Three launches, no copy, no slot.
// Same verdicts as the general path by construction of verify_record_verdict: [2] keys that do not decode, [4] non-canonical s (counted by the host here),
// [3] R_i that do not decode, then the identity check.
// H(R || A || M) on the host, block-wise (sha512_stream's select chains are written for registers of a GPU lane; a 64-bit core wants plain buffers)
static void host_hram_words(const uint8_t *R, const uint8_t *A, const uint8_t *m, uint64_t len, uint32_t w16[16]) {
    u64 h[8];
    sha512_init(h);
    uint8_t blk[128];
    memcpy(blk, R, 32); memcpy(blk + 32, A, 32);
    size_t fill = 64;
    const uint64_t bits = (64 + len) * 8;
    auto flush = [&]() {
        u64 w[16];
        for (int q = 0; q < 16; q++) { u64 v; memcpy(&v, blk + 8 * q, 8); w[q] = bswap64(v); }
        sha512_compress(h, w);
        fill = 0;
    };
    while (len) {
        const size_t k = std::min<uint64_t>(len, 128 - fill);
        memcpy(blk + fill, m, k);
        fill += k; m += k; len -= k;
        if (fill == 128) flush();
    }
    blk[fill++] = 0x80;
    if (fill > 112) { memset(blk + fill, 0, 128 - fill); flush(); }
    memset(blk + fill, 0, 120 - fill);                       // (the upper 64 bits of the 128-bit length are zero)
    for (int b = 0; b < 8; b++) blk[120 + b] = (uint8_t)(bits >> (56 - 8 * b));
    flush();
    sha512_digest_words(h, w16);
}
static bool verify_small_host_ok(uint64_t n, uint32_t z_mode, msm_geom &g) {
    static const int host_max = C25519_KNOB("VERIFY_HOST_MAX", 128);      // A/B knob: 0 = the general path at every size (128 = the one-block decompression kernel's limit;
    //                                                                         the host needs 0.7 - 0.85 us per signature, hidden behind that kernel up to ~100: profiles/r05_small_call_phases.txt)
    if (z_mode > 1 || n == 0 || n > (uint64_t)host_max || n > 128) return false;
    msm_layout(2 * n + 1, g, 16);
    return g.half <= 64 && g.nwin <= 64;
}
static int32_t verify_batch_small_host(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_points, uint64_t n,
                                       uint32_t z_mode, const msm_geom &g) {
    const uint64_t m = 2 * n + 1;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t oS = 0, oK = al(n * 64), oC = oK + al(n * 32), oH = oC + al(m * 32), oZ = oH + al(n * 64), oP = oZ + al((n + 4) * 16 + n * 32), total = oP + (pk_points ? al(n * 160) : 0);
    int32_t r;
    ctx->host_us[0] = wall_us();
    ffi_small_begin(ctx);
    if ((r = ctx_host_stage(ctx, total + 256)) || (r = ctx_reserve(ctx, ctx->tmp_e, m * PTS_BYTES + 256))) return r;
    uint8_t *hs = (uint8_t *)ctx->h_stage, *dv = nullptr;
    HIPCHK(hipHostGetDevicePointer((void **)&dv, hs, 0));
    memcpy(hs + oS, sigs, n * 64);
    memcpy(hs + oK, pks, n * 32);
    if (pk_points) memcpy(hs + oP, pk_points, n * 160);
    uint32_t *d_pts = (uint32_t *)ctx->tmp_e.p, *cnt = (uint32_t *)ctx->d_flag + 44;
    ctx->last_passes.clear();
    HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
    HIPCHK(launch_prep_small_verify(dv + oK, pk_points ? dv + oP : nullptr, dv + oS, n, d_pts, cnt, ctx->stream));
    ctx->host_us[1] = wall_us();
    // ---- the host's share, beside the decompression ----
    uint32_t bad_s = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t w[16];
        host_hram_words(sigs + i * 64, pks + i * 32, msgs + msg_off[i], msg_off[i + 1] - msg_off[i], w);
        memcpy(hs + oH + i * 64, w, 64);
    }
    const bool dev_z = z_mode == C25519_Z_DEVICE;
    if (!dev_z) c25519_transcript_zs(hs + oH, hs + oS, n, hs + oZ);
    else {
        // the device z-mode's derivation (header: NOT the reference's), computed here: h_i mod l, the hash tree, the quarters (ztree_host_zs)
        uint8_t *hred = hs + oZ + (n + 4) * 16;
        for (uint64_t i = 0; i < n; i++) {
            uint32_t hw[16], o[8];
            memcpy(hw, hs + oH + i * 64, 64);
            sc_to_words(sc_from_wide(hw), o);
            memcpy(hred + 32 * i, o, 32);
        }
        ztree_host_zs(hred, hs + oS, n, hs + oZ);
    }
    uint8_t *msc = hs + oC;
    sc52 sum = sc_zero();
    for (uint64_t i = 0; i < n; i++) {
        uint32_t zw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sw[8], hw[16], o[8];
        memcpy(zw, hs + oZ + i * 16, 16);
        memcpy(sw, sigs + i * 64 + 32, 32);
        memcpy(hw, hs + oH + i * 64, 64);
        const bool canon = sc_is_canonical(sw);
        if (!canon) bad_s++;                                       // (the verdict is then ScalarFormat or an earlier one whatever the sum is)
        // (device z-mode: sign-magnitude, bit 127 = sign.  The general path keeps |z_i| as the scalar and negates the stored R_i so that the R terms stay out of
        //  the upper windows of a 2^20-term sort; here every term goes through every window anyway: the scalar is z_i mod l)
        const bool neg = dev_z && (zw[3] >> 31);
        if (dev_z) zw[3] &= 0x7fffffffu;
        sc52 z = sc_from_words(zw);
        if (neg) { z = sc_neg(z); sc_to_words(z, zw); }
        const sc52 sc = canon ? sc_from_words(sw) : sc_zero();
        sum = sc_add(sum, sc_mul(z, sc));
        sc_to_words(sc_mul(z, sc_from_wide(hw)), o);
        memcpy(msc + 32 * (1 + i), zw, 32);                        // R_i: z_i
        memcpy(msc + 32 * (1 + n + i), o, 32);                     // A_i: z_i h_i
    }
    { uint32_t o[8]; sc_to_words(sc_neg(sum), o); memcpy(msc, o, 32); }      // B: -sum z_i s_i   (batch.rs:240)
    // ---- the MSM of 2n + 1 terms over the records, published by its last kernel ----
    ctx->direct_seq = ++ctx->publish_seq;
    if (!ctx->direct_seq) ctx->direct_seq = ++ctx->publish_seq;
    ctx->direct_extra = cnt;
    r = msm_small_enqueue(ctx, dv + oC, d_pts, 1, m, g, drec(ctx), ctx->stream);
    ctx->direct_extra = nullptr;
    if (r) { ctx->direct_seq = 0; (void)hipStreamSynchronize(ctx->stream); return r; }
    r = rec_collect(ctx);
    uint32_t late_cnt[2] = {0, 0};
    if (r == C25519_LOST_PUBLICATION) {
        // (r6) never observed (msm.hip wait_published): the small MSM once more, into the device slot, and the record by copy; the decode counters of the
        // first kernel (still in d_flag) with it.  The staged scalars and the records of A_i, R_i, B are where they were.
        slot_init(drec(ctx), m, nullptr, ctx->stream, g.c);
        if ((r = msm_small_enqueue(ctx, dv + oC, d_pts, 1, m, g, drec(ctx), ctx->stream))) { (void)hipStreamSynchronize(ctx->stream); return r; }
        if ((r = rec_collect(ctx))) return r;
        HIPCHK(hipMemcpy(late_cnt, cnt, 8, hipMemcpyDeviceToHost));
        ctx->err.clear();
    }
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
    ge_p3 R;
    uint32_t flags[8];
    if ((r = records_fold((const uint8_t *)hslot(ctx, C25519_MAX_SLOTS), 1, R, flags, &ctx->err))) return r;
    flags[2] += late_cnt[0]; flags[3] += late_cnt[1];
    flags[4] += bad_s;
    r = verify_record_verdict(ctx, R, flags);
    ffi_small_end(ctx, 0, 0);                                      // (nothing is copied: the kernels read and write host memory in place)
    ctx->host_us[4] = wall_us();
    return r;
}
EXPORT int32_t ed25519_verify_batch_keys(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks,
                                         const uint8_t *pk_points, uint64_t n, uint32_t z_mode) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;
    if (z_mode > 1) { ctx->err = "verify_batch: bad z_mode"; return -(int32_t)hipErrorInvalidValue; }
    if (!offsets_ok(msg_off, n)) { ctx->err = "verify_batch: msg_off is not monotone"; return -(int32_t)hipErrorInvalidValue; }
    const uint64_t mlen = msg_off[n];
    int32_t r;
    if (2 * n + 1 <= small_upload_max_terms() && mlen <= (1u << 20)) {
        { msm_geom gh; if (verify_small_host_ok(n, z_mode, gh)) return verify_batch_small_host(ctx, msgs, msg_off, sigs, pks, pk_points, n, z_mode, gh); }
        // the reference's own benchmark sizes (ed25519_benchmarks.rs:53: 4 .. 256 signatures) and everything else whose MSM takes the small path:
        // all five arrays through one staged copy on the compute stream (capi.hip ffi_small_upload)
        const void *src[5] = {msgs, msg_off, sigs, pks, pk_points};
        const size_t bytes[5] = {(size_t)mlen, (size_t)(n + 1) * 8, (size_t)n * 64, (size_t)n * 32, pk_points ? (size_t)n * 160 : 0};
        uint8_t *d[5];
        ctx->host_us[0] = wall_us();
        if ((r = ffi_small_upload(ctx, 5, src, bytes, d, (size_t)n * 144 + 64))) return r;
        ctx->host_us[1] = wall_us();
        r = verify_batch_impl(ctx, d[0], (const uint64_t *)d[1], mlen, d[2], d[3], pk_points ? d[4] : nullptr, n, z_mode, nullptr);
        if (r < 0) (void)hipStreamSynchronize(ctx->stream);      // (an error before the call's own synchronisation: the upload may still be reading the staging buffer)
        ffi_small_end(ctx, mlen + (n + 1) * 8 + n * 96 + (pk_points ? n * 160 : 0), 0);
        ctx->host_us[4] = wall_us();
        return r;
    }
    if ((r = ctx_reserve(ctx, ctx->tmp_a, mlen + 64)) || (r = ctx_reserve(ctx, ctx->tmp_b, (n + 1) * 8)) || (r = ctx_reserve(ctx, ctx->tmp_c, n * 64)) ||
        (r = ctx_reserve(ctx, ctx->scratch, n * 32 + (pk_points ? n * 160 : 0) + 16)))
        return r;
    uint8_t *d_msg = (uint8_t *)ctx->tmp_a.p, *d_sig = (uint8_t *)ctx->tmp_c.p, *d_pk = (uint8_t *)ctx->scratch.p, *d_pp = pk_points ? d_pk + n * 32 : nullptr;
    uint64_t *d_off = (uint64_t *)ctx->tmp_b.p;
    if ((r = ffi_begin(ctx))) return r;
    ffi_guard guard(ctx);                                 // the early exits of the uploads below drain the copy stream as well
    uint64_t up = 0;
    if (z_mode == C25519_Z_TRANSCRIPT) {
        // the whole batch is hashed before anything else can start and the sequential host transcript dominates: upload everything
        if (mlen) HIPCHK(hipMemcpyAsync(d_msg, msgs, mlen, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(hipMemcpyAsync(d_off, msg_off, (n + 1) * 8, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(hipMemcpyAsync(d_sig, sigs, n * 64, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(hipMemcpyAsync(d_pk, pks, n * 32, hipMemcpyHostToDevice, ctx->s_h2d));
        if (pk_points) HIPCHK(hipMemcpyAsync(d_pp, pk_points, n * 160, hipMemcpyHostToDevice, ctx->s_h2d));
        HIPCHK(hipEventRecord(ctx->ev_up[0], ctx->s_h2d));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_up[0], 0));
        up = mlen + (n + 1) * 8 + n * 96 + (pk_points ? n * 160 : 0);
        r = verify_batch_impl(ctx, d_msg, d_off, mlen, d_sig, d_pk, d_pp, n, z_mode, nullptr);
    } else {
        // device z-mode: every array goes up right before the first kernels that need it (verify_pass_enqueue), pass by pass
        int slot = 0;
        const verify_fetch fetch = [&](uint64_t lo, uint64_t m, int what, hipEvent_t *ready) -> int32_t {
            if (what == 0) { HIPCHK(hipMemcpyAsync(d_sig + lo * 64, sigs + lo * 64, m * 64, hipMemcpyHostToDevice, ctx->s_h2d)); up += m * 64; }
            else if (what == 1) { HIPCHK(hipMemcpyAsync(d_pk + lo * 32, pks + lo * 32, m * 32, hipMemcpyHostToDevice, ctx->s_h2d)); up += m * 32; }
            else if (what == 2) {
                // the kernels index the blob through ABSOLUTE offsets: this pass's offsets and the bytes they span, in place
                const uint64_t b0 = msg_off[lo], b1 = msg_off[lo + m];
                if (b1 > b0) HIPCHK(hipMemcpyAsync(d_msg + b0, msgs + b0, b1 - b0, hipMemcpyHostToDevice, ctx->s_h2d));
                HIPCHK(hipMemcpyAsync(d_off + lo, msg_off + lo, (m + 1) * 8, hipMemcpyHostToDevice, ctx->s_h2d));
                up += (b1 - b0) + (m + 1) * 8;
            } else { HIPCHK(hipMemcpyAsync(d_pp + lo * 160, pk_points + lo * 160, m * 160, hipMemcpyHostToDevice, ctx->s_h2d)); up += m * 160; }
            HIPCHK(hipEventRecord(ctx->ev_up[slot], ctx->s_h2d));
            *ready = ctx->ev_up[slot];
            slot = (slot + 1) % c25519_ctx::FFI_MAXCH;
            return C25519_OK;
        };
        r = verify_batch_impl(ctx, d_msg, d_off, mlen, d_sig, d_pk, d_pp, n, z_mode, &fetch);
    }
    guard.dismiss();
    const int32_t r2 = ffi_end(ctx, up, 0);
    return (r < 0 || !r2) ? r : r2;
}
EXPORT int32_t ed25519_verify_batch(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks,
                                    uint64_t n, uint32_t z_mode) {
    return ed25519_verify_batch_keys(ctx, msgs, msg_off, sigs, pks, nullptr, n, z_mode);
}

// diagnostics: the z_i a batch of n <= VERIFY_PASS_MAX signatures gets (16 bytes each to the HOST buffer out_z16;
// device z-mode: sign-magnitude, see k_zderive).  For the tests that pin the derivation's dependence on every input.
EXPORT int32_t c25519_debug_batch_zs(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks, uint64_t n,
                                     uint32_t z_mode, uint8_t *out_z16) {
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return C25519_OK;
    if (n > VERIFY_PASS_MAX || z_mode > 2 || !offsets_ok(msg_off, n)) { ctx->err = "debug_batch_zs: bad arguments"; return -(int32_t)hipErrorInvalidValue; }
    if (z_mode == 2) {
        // the device z-mode's values computed by the HOST restatement (ztree_host_zs): must equal z_mode 1 byte for byte
        std::vector<uint8_t> hred(n * 32), z((n + 4) * 16);
        for (uint64_t i = 0; i < n; i++) {
            sha512_stream st;
            st.init();
            st.put_bytes(sigs + i * 64, 32); st.put_bytes(pks + i * 32, 32); st.put_bytes(msgs + msg_off[i], msg_off[i + 1] - msg_off[i]);
            st.finish();
            uint32_t w[16], o[8];
            sha512_digest_words(st.h, w);
            sc_to_words(sc_from_wide(w), o);
            memcpy(&hred[32 * i], o, 32);
        }
        ztree_host_zs(hred.data(), sigs, n, z.data());
        memcpy(out_z16, z.data(), n * 16);
        return C25519_OK;
    }
    const uint64_t mlen = msg_off[n];
    int32_t r;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t oM = carve(mlen + 64), oO = carve((n + 1) * 8), oS = carve(n * 64), oK = carve(n * 32), oH = carve(n * 64), oZ = carve((n + 4) * 16), oT0 = carve((n / 4 + 2) * 32), oT1 = carve((n / 4 + 2) * 32);
    const size_t oHr = carve(n * 32);
    if ((r = ctx_reserve(ctx, ctx->tmp_f, off))) return r;
    uint8_t *ws = (uint8_t *)ctx->tmp_f.p;
    hipStream_t st = ctx->stream;
    if (mlen) HIPCHK(hipMemcpyAsync(ws + oM, msgs, mlen, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ws + oO, msg_off, (n + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ws + oS, sigs, n * 64, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ws + oK, pks, n * 32, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(ctx->d_flag, 0, 16, st));
    HIPCHK(launch_hram(ws + oM, (const uint64_t *)(ws + oO), mlen, ws + oS, ws + oK, n, ws + oH, (uint32_t *)ctx->d_flag, st));
    if (z_mode == C25519_Z_DEVICE) {
        hipLaunchKernelGGL(k_hram_mod_l, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws + oH, n, ws + oHr);
        if ((r = zchain_enqueue(ctx, st, ws + oHr, ws + oS, n, ws + oT0, ws + oT1, ws + oZ))) return r;
        HIPCHK(hipMemcpyAsync(out_z16, ws + oZ, n * 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
    } else {
        std::vector<uint8_t> hh(n * 64);
        HIPCHK(hipMemcpyAsync(hh.data(), ws + oH, n * 64, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        c25519_transcript_zs(hh.data(), sigs, n, out_z16);
    }
    return C25519_OK;
}
