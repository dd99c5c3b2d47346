// Host build of the DEVICE headers (fe26.h / ge26.h are __host__ __device__) with bound checking
// enabled, so the exact arithmetic the HIP kernels run can be fuzzed on a machine with no GPU.
// Test-only: built by tests/test_fe26_host.py into tests/host/libfe26host.so.
#define C25519_CHECK_BOUNDS 1
#include "../../curve25519-dalek_amd/csrc/ge26.h"
#include "../../curve25519-dalek_amd/csrc/fe9_probe.h"
#include <string.h>
#include <vector>
using namespace c25519;

static feT load(const uint8_t b[32]) { u32 w[8]; memcpy(w, b, 32); return fe_from_words(w); }
static void store(uint8_t b[32], const feW &a) { u32 w[8]; fe_to_words(a, w); memcpy(b, w, 32); }

extern "C" {
// the nine-limb probe representation (csrc/fe9_probe.h): raw limbs in, raw limbs out (9 x u32), op 0 = mul, 1 = sq
void h_fe9(int op, const uint32_t *a, const uint32_t *b, uint32_t *o) {
    fe9 x, y;
    for (int i = 0; i < 9; i++) { x.v[i] = a[i]; y.v[i] = b[i]; }
    const fe9 r = op ? fe9_sq(x) : fe9_mul(x, y);
    for (int i = 0; i < 9; i++) o[i] = r.v[i];
}
void h_fe_mul(const uint8_t *a, const uint8_t *b, uint8_t *o) { store(o, fe_mul(load(a), load(b))); }
void h_fe_sq(const uint8_t *a, uint8_t *o) { store(o, fe_sq(load(a))); }
void h_fe_add(const uint8_t *a, const uint8_t *b, uint8_t *o) { store(o, fe_add(load(a), load(b))); }
void h_fe_sub(const uint8_t *a, const uint8_t *b, uint8_t *o) { store(o, fe_sub(load(a), load(b))); }
void h_fe_invert(const uint8_t *a, uint8_t *o) { store(o, fe_invert(load(a))); }
void h_fe_pow_p58(const uint8_t *a, uint8_t *o) { store(o, fe_pow_p58(load(a))); }
void h_fe_canon(const uint8_t *a, uint8_t *o) { store(o, load(a)); }
void h_fe_mul_small(const uint8_t *a, uint32_t c, uint8_t *o) { store(o, fe_mul_small(load(a), c)); }
// raw-limb entry points: f wide-bounded, g loose-bounded
void h_fe_mul_limbs(const uint32_t *f, const uint32_t *g, uint8_t *o) {
    feW F; feL G; for (int i = 0; i < 10; i++) { F.v[i] = f[i]; G.v[i] = g[i]; }
    store(o, fe_mul(F, G));
}
void h_fe_sq_limbs(const uint32_t *f, uint8_t *o) { feL F; for (int i = 0; i < 10; i++) F.v[i] = f[i]; store(o, fe_sq(F)); }
void h_fe_carry_limbs(const uint32_t *f, uint8_t *o) { feW F; for (int i = 0; i < 10; i++) F.v[i] = f[i]; store(o, fe_carry(F)); }
void h_fe_tobytes_limbs(const uint32_t *f, uint8_t *o) { feW F; for (int i = 0; i < 10; i++) F.v[i] = f[i]; store(o, F); }
int h_fe_sqrt_ratio_i(const uint8_t *u, const uint8_t *v, uint8_t *o) { feT r; bool ok = fe_sqrt_ratio_i(r, load(u), load(v)); store(o, r); return ok; }

// points: 4 x 32 canonical bytes (X,Y,Z,T)
static ge_p3 pload(const uint8_t *p) { ge_p3 r; r.X = load(p); r.Y = load(p + 32); r.Z = load(p + 64); r.T = load(p + 96); return r; }
static void pstore(uint8_t *p, const ge_p3 &r) { store(p, r.X); store(p + 32, r.Y); store(p + 64, r.Z); store(p + 96, r.T); }
void h_ge_basepoint(uint8_t *o) { pstore(o, ge_basepoint()); }
void h_ge_identity(uint8_t *o) { pstore(o, ge_identity()); }
void h_ge_add(const uint8_t *a, const uint8_t *b, uint8_t *o) { pstore(o, ge_add(pload(a), pload(b))); }
void h_ge_dbl(const uint8_t *a, uint8_t *o) { pstore(o, ge_dbl_p3(pload(a))); }
void h_ge_neg(const uint8_t *a, uint8_t *o) { pstore(o, ge_neg(pload(a))); }
void h_ge_mul_by_pow_2(const uint8_t *a, int k, uint8_t *o) { pstore(o, ge_mul_by_pow_2(pload(a), k)); }
int h_ge_eq(const uint8_t *a, const uint8_t *b) { return ge_eq(pload(a), pload(b)); }
int h_ge_is_identity(const uint8_t *a) { return ge_is_identity(pload(a)); }
int h_ge_decompress(const uint8_t *in, uint8_t *o) { u32 w[8]; memcpy(w, in, 32); ge_p3 p; bool ok = ge_decompress(p, w); pstore(o, p); return ok; }
void h_ge_compress(const uint8_t *a, uint8_t *o) {
    ge_p3 p = pload(a); feT zi = fe_invert(p.Z);
    u32 w[8]; ge_affine_compress(fe_mul(p.X, zi), fe_mul(p.Y, zi), w); memcpy(o, w, 32);
}
// p +/- q where q is given as an extended point with Z = 1 (converted to affine Niels here)
void h_ge_madd(const uint8_t *a, const uint8_t *q, int neg, uint8_t *o) {
    ge_p3 Q = pload(q);
    u32 w[24];
    fe_to_words(fe_add(Q.Y, Q.X), w); fe_to_words(fe_sub(Q.Y, Q.X), w + 8); fe_to_words(fe_mul(Q.T, fe_d2()), w + 16);
    aniels_words_cneg(w, neg != 0);
    pstore(o, ge_p1p1_to_p3(ge_madd(pload(a), aniels_from_words(w))));
}
// the same through ge_madd_signed_p3 (table entry kept as limbs; used by k_mul_base_wide)
void h_ge_madd_signed(const uint8_t *a, const uint8_t *q, int neg, uint8_t *o) {
    ge_p3 Q = pload(q);
    ge_aniels A;
    A.ypx = fe_carry(fe_add(Q.Y, Q.X)); A.ymx = fe_carry(fe_sub(Q.Y, Q.X)); A.xy2d = fe_mul(Q.T, fe_d2());
    pstore(o, ge_madd_signed_p3(pload(a), A, neg != 0));
}
// +/- q as an extended point straight from its affine Niels form (first addition of a chain)
void h_ge_from_aniels_signed(const uint8_t *q, int neg, uint8_t *o) {
    ge_p3 Q = pload(q);
    ge_aniels A;
    A.ypx = fe_carry(fe_add(Q.Y, Q.X)); A.ymx = fe_carry(fe_sub(Q.Y, Q.X)); A.xy2d = fe_mul(Q.T, fe_d2());
    pstore(o, ge_from_aniels_signed(A, neg != 0));
}
// p +/- q through the cached (ProjectiveNiels) path with the per-lane conditional negation
void h_ge_add_cached_signed(const uint8_t *a, const uint8_t *q, int neg, uint8_t *o) {
    pstore(o, ge_p1p1_to_p3(ge_add_cached(pload(a), ge_cached_cneg(ge_p3_to_cached(pload(q)), neg != 0))));
}
// X25519 ladder exactly as the kernel runs it (montgomery.rs:183-211), s = already-clamped scalar
void h_x25519_ladder(const uint8_t *s, const uint8_t *u, uint8_t *o) {
    feT au = load(u); mont_pp x0, x1; x0.U = fe_one(); x0.W = fe_zero(); x1.U = au; x1.W = fe_one();
    u32 prev = 0;
    for (int i = 254; i >= 0; i--) {
        u32 cur = (s[i >> 3] >> (i & 7)) & 1;
        u32 sw = prev ^ cur; fe_cswap(x0.U, x1.U, sw); fe_cswap(x0.W, x1.W, sw);
        mont_diff_add_and_double(x0, x1, au); prev = cur;
    }
    fe_cswap(x0.U, x1.U, prev); fe_cswap(x0.W, x1.W, prev);
    store(o, fe_mul(x0.U, fe_invert(x0.W)));
}
}

// ---- scalar arithmetic, SHA-512 and the host transcript of the verify_batch pipeline ----------------
#include "../../curve25519-dalek_amd/csrc/sc_sha.h"
#include "../../curve25519-dalek_amd/csrc/transcript_host.h"
extern "C" {
void h_sc_mul(const uint8_t *a, const uint8_t *b, uint8_t *o) { u32 x[8], y[8], r[8]; memcpy(x, a, 32); memcpy(y, b, 32); sc_to_words(sc_mul(sc_from_words(x), sc_from_words(y)), r); memcpy(o, r, 32); }
void h_sc_add(const uint8_t *a, const uint8_t *b, uint8_t *o) { u32 x[8], y[8], r[8]; memcpy(x, a, 32); memcpy(y, b, 32); sc_to_words(sc_add(sc_from_words(x), sc_from_words(y)), r); memcpy(o, r, 32); }
void h_sc_neg(const uint8_t *a, uint8_t *o) { u32 x[8], r[8]; memcpy(x, a, 32); sc_to_words(sc_neg(sc_from_words(x)), r); memcpy(o, r, 32); }
void h_sc_from_wide(const uint8_t *a, uint8_t *o) { u32 x[16], r[8]; memcpy(x, a, 64); sc_to_words(sc_from_wide(x), r); memcpy(o, r, 32); }
int h_sc_is_canonical(const uint8_t *a) { u32 x[8]; memcpy(x, a, 32); return sc_is_canonical(x); }
void h_sha512(const uint8_t *m, size_t n, uint8_t *o) {
    sha512_stream st; st.init();
    size_t i = 0;
    for (; i + 8 <= n && i < 64; i += 8) { u64 v; memcpy(&v, m + i, 8); st.put_be64(bswap64(v)); }   // word path, like the kernels' prefix
    for (; i < n; i++) st.put_byte(m[i]);
    st.finish();
    u32 w[16]; sha512_digest_words(st.h, w); memcpy(o, w, 64);
}
// the kernels' message path: `pre` bytes one by one (any block position), then put_bytes over the rest (word-wise where it can)
void h_sha512_put_bytes(const uint8_t *m, size_t n, size_t pre, uint8_t *o) {
    sha512_stream st; st.init();
    size_t i = 0;
    for (; i < pre && i < n; i++) st.put_byte(m[i]);
    st.put_bytes(m + i, n - i);
    st.finish();
    u32 w[16]; sha512_digest_words(st.h, w); memcpy(o, w, 64);
}
void h_transcript_zs(const uint8_t *hrams, const uint8_t *sigs, uint64_t n, uint8_t *zs) { c25519_transcript_zs(hrams, sigs, n, zs); }
// every form of Keccak-f[1600] this host can run (generic; BMI2; AVX-512VL lane pairs -- whichever keccak_pick() would take among them is one of these): `which` = 0 / 1 / 2,
// returns 0 when the host lacks the instructions (o untouched)
int h_keccak_form(const uint8_t *st, int which, uint8_t *o) {
    uint64_t a[25]; memcpy(a, st, 200);
    __builtin_cpu_init();
    if (which == 0) c25519_tr::keccak_f_generic(a);
    else if (which == 1) { if (!(__builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2"))) return 0; c25519_tr::keccak_f_bmi2(a); }
    else { if (!(__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl"))) return 0; c25519_tr::keccak_f_pairs(a); }
    memcpy(o, a, 200); return 1;
}
const char *h_keccak_impl() { return c25519_tr::keccak_impl(); }
}
// ---- BLAKE2b (csrc/blake2b.h: the compression function of the device z-mode's hash tree) as the plain unkeyed hash of RFC 7693 section 3.3 ----
#include "../../curve25519-dalek_amd/csrc/blake2b.h"
extern "C" {
void h_blake2b(const uint8_t *msg, uint64_t len, uint32_t outlen, uint8_t *o) {
    u64 h[8], m[16];
    blake2b_init(h, outlen);
    uint64_t off = 0;
    while (len - off > 128) { memcpy(m, msg + off, 128); off += 128; blake2b_compress(h, m, off, false); }      // (the last block is never empty unless the message is)
    uint8_t blk[128] = {0};
    memcpy(blk, msg + off, len - off);
    memcpy(m, blk, 128);
    blake2b_compress(h, m, len, true);
    memcpy(o, h, outlen);
}
}

// ---- device scalar arithmetic on 28-bit limbs (csrc/sc28.h) --------------------------------------------------------------
#include "../../curve25519-dalek_amd/csrc/sc28.h"
extern "C" {
void h_sc28_from_wide(const uint8_t *a, uint8_t *o) { u32 x[16], r[8]; memcpy(x, a, 64); sc28_to_words(sc28_from_wide(x), r); memcpy(o, r, 32); }
void h_sc28_mul_5x10(const uint8_t *z16, const uint8_t *b, uint8_t *o) {
    u32 zw[4], bw[8], zl[5], r[8]; memcpy(zw, z16, 16); memcpy(bw, b, 32);
    sc28_limbs_from_words<4, 5>(zw, zl);
    sc28_to_words(sc28_mul_5x10(zl, sc28_from_words(bw).v), r); memcpy(o, r, 32);
}
void h_sc28_mul(const uint8_t *a, const uint8_t *b, uint8_t *o) {
    u32 aw[8], bw[8], r[8]; memcpy(aw, a, 32); memcpy(bw, b, 32);
    sc28_to_words(sc28_mul(sc28_from_words(aw).v, sc28_from_words(bw).v), r); memcpy(o, r, 32);
}
void h_sc28_add(const uint8_t *a, const uint8_t *b, uint8_t *o) { u32 aw[8], bw[8], r[8]; memcpy(aw, a, 32); memcpy(bw, b, 32); sc28_to_words(sc28_add(sc28_from_words(aw), sc28_from_words(bw)), r); memcpy(o, r, 32); }
void h_sc28_neg(const uint8_t *a, uint8_t *o) { u32 aw[8], r[8]; memcpy(aw, a, 32); sc28_to_words(sc28_neg(sc28_from_words(aw)), r); memcpy(o, r, 32); }
int h_sc28_canonical(const uint8_t *a) { u32 x[8]; memcpy(x, a, 32); return sc28_words_canonical(x); }
void h_sc28_roundtrip(const uint8_t *a, uint8_t *o) { u32 x[8], r[8]; memcpy(x, a, 32); sc28_to_words(sc28_from_words(x), r); memcpy(o, r, 32); }
}

// ---- the host fold's doubling chain: scalar (64 x 64 -> 128) against the AVX-512 IFMA form (csrc/host51.h) -----------------------------------------------------
#include "../../curve25519-dalek_amd/csrc/host51.h"
extern "C" {
int h_has_ifma() { return host_has_ifma() ? 1 : 0; }
// in: X, Y, Z as 3 x 5 u64 limbs (any values below 2^52); k doublings; out: affine x, y (canonical 32-byte encodings) and x * y against T / Z from each path
static void h_affine(const hp3 &r, uint8_t *o) {
    const h51 zi = h51_invert(r.Z);
    u32 w[8];
    fe_to_words(h51_to_fe(h51_mul(r.X, zi)), w); memcpy(o, w, 32);
    fe_to_words(h51_to_fe(h51_mul(r.Y, zi)), w); memcpy(o + 32, w, 32);
    fe_to_words(h51_to_fe(h51_mul(r.T, zi)), w); memcpy(o + 64, w, 32);
}
void h_pow2_scalar(const uint64_t *xyz, int k, uint8_t *o) {
    hp3 p; for (int i = 0; i < 5; i++) { p.X.v[i] = xyz[i]; p.Y.v[i] = xyz[5 + i]; p.Z.v[i] = xyz[10 + i]; p.T.v[i] = 0; }
    h_affine(hp3_mul_by_pow_2(p, k), o);
}
// the whole fold: n columns (X, Y, Z, T as 4 x 5 u64 limbs each, top window first), shift[k] doublings in front of column k; ifma = 1: the lane form (where the CPU has it)
void h_horner(const uint64_t *cols, const int *shift, int n, int ifma, uint8_t *o) {
    std::vector<hp3> c((size_t)n);
    for (int k = 0; k < n; k++) for (int i = 0; i < 5; i++) { c[k].X.v[i] = cols[20 * k + i]; c[k].Y.v[i] = cols[20 * k + 5 + i]; c[k].Z.v[i] = cols[20 * k + 10 + i]; c[k].T.v[i] = cols[20 * k + 15 + i]; }
    if (ifma) { h_affine(hp3_horner(c.data(), shift, n), o); return; }
    hp3 total = hp3_identity();
    for (int k = 0; k < n; k++) { if (k) total = hp3_mul_by_pow_2(total, shift[k]); total = hp3_add(total, c[k]); }
    h_affine(total, o);
}
void h_pow2_ifma(const uint64_t *xyz, int k, uint8_t *o) {
    hp3 p; for (int i = 0; i < 5; i++) { p.X.v[i] = xyz[i]; p.Y.v[i] = xyz[5 + i]; p.Z.v[i] = xyz[10 + i]; p.T.v[i] = 0; }
#if defined(__x86_64__) && !defined(C25519_NO_IFMA)
    if (host_has_ifma()) { h_affine(hp3_mul_by_pow_2_ifma(p, k), o); return; }
#endif
    h_affine(hp3_mul_by_pow_2(p, k), o);
}
}
