/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (curve25519-dalek_amd/) never does.
 *
 * Exported C API (ctypes) over the CPU restatement of the reference's serial u64 path.
 * PARITY PINNING: the reference is Rust and cannot be built here (no rustc/cargo), so there is no
 * oracle/_ref.  The restatement is pinned against the reference's own golden vectors
 * (tests/golden/, extracted by tests/golden/extract_vectors.py from /root/reference) -- see
 * tests/test_oracle_*.py.  The only unpinned piece is the z_i derivation of verify_batch
 * (STROBE transcript): the reference holds no byte-level vector for it ("parity unpinned" for z).
 */
#include <pthread.h>
#include <stdio.h>
#include "ge.h"
#include "hashes.h"

#define EXPORT __attribute__((visibility("default")))

/* raw 160-byte point <-> ge_p3 : {X,Y,Z,T} x 5 x u64, the ABI's point_fmt 2 */
static ge_p3 p3_load(const uint64_t *p) { ge_p3 r; memcpy(&r, p, 160); return r; }
static void p3_store(uint64_t *p, ge_p3 r) { memcpy(p, &r, 160); }

/* ---------------- field ---------------- */
EXPORT void orc_fe_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { fe_to_bytes(out, fe_mul(fe_from_bytes(a), fe_from_bytes(b))); }
EXPORT void orc_fe_sq(const uint8_t a[32], uint8_t out[32]) { fe_to_bytes(out, fe_sq(fe_from_bytes(a))); }
EXPORT void orc_fe_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { fe_to_bytes(out, fe_add(fe_from_bytes(a), fe_from_bytes(b))); }
EXPORT void orc_fe_sub(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { fe_to_bytes(out, fe_sub(fe_from_bytes(a), fe_from_bytes(b))); }
EXPORT void orc_fe_neg(const uint8_t a[32], uint8_t out[32]) { fe_to_bytes(out, fe_neg(fe_from_bytes(a))); }
EXPORT void orc_fe_invert(const uint8_t a[32], uint8_t out[32]) { fe_to_bytes(out, fe_invert(fe_from_bytes(a))); }
EXPORT void orc_fe_pow_p58(const uint8_t a[32], uint8_t out[32]) { fe_to_bytes(out, fe_pow_p58(fe_from_bytes(a))); }
EXPORT void orc_fe_canon(const uint8_t a[32], uint8_t out[32]) { fe_to_bytes(out, fe_from_bytes(a)); }
EXPORT int orc_fe_sqrt_ratio_i(const uint8_t u[32], const uint8_t v[32], uint8_t out[32]) {
    fe r; int ok = fe_sqrt_ratio_i(&r, fe_from_bytes(u), fe_from_bytes(v)); fe_to_bytes(out, r); return ok;
}
EXPORT void orc_fe_invert_batch(uint8_t *io, size_t n) {
    fe *x = malloc((n ? n : 1) * sizeof(fe)), *s = malloc((n ? n : 1) * sizeof(fe));
    for (size_t i = 0; i < n; i++) x[i] = fe_from_bytes(io + 32 * i);
    fe_invert_batch(x, s, n);
    for (size_t i = 0; i < n; i++) fe_to_bytes(io + 32 * i, x[i]);
    free(x); free(s);
}
/* raw-limb entry points: let the tests feed unreduced limbs (< 2^54) like the reference's own
   overflow tests do (u64/field.rs:162-166) */
EXPORT void orc_fe_mul_limbs(const uint64_t a[5], const uint64_t b[5], uint64_t out[5]) { fe r = fe_mul(fe_from_limbs(a), fe_from_limbs(b)); memcpy(out, r.v, 40); }
EXPORT void orc_fe_to_bytes_limbs(const uint64_t a[5], uint8_t out[32]) { fe_to_bytes(out, fe_from_limbs(a)); }

/* ---------------- scalars ---------------- */
EXPORT void orc_sc_from_bytes_mod_order(const uint8_t in[32], uint8_t out[32]) { sc_to_bytes(out, sc_from_bytes_mod_order(in)); }
EXPORT void orc_sc_from_bytes_mod_order_wide(const uint8_t in[64], uint8_t out[32]) { sc_to_bytes(out, sc_from_bytes_wide(in)); }
EXPORT int orc_sc_is_canonical(const uint8_t in[32]) { return sc_is_canonical_bytes(in); }
EXPORT void orc_sc_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { sc_to_bytes(out, sc_mul(sc_from_bytes(a), sc_from_bytes(b))); }
EXPORT void orc_sc_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { sc_to_bytes(out, sc_add(sc_from_bytes(a), sc_from_bytes(b))); }
EXPORT void orc_sc_sub(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { sc_to_bytes(out, sc_sub(sc_from_bytes(a), sc_from_bytes(b))); }
EXPORT void orc_sc_neg(const uint8_t a[32], uint8_t out[32]) { sc_to_bytes(out, sc_sub(SC_ZERO, sc_from_bytes(a))); }
EXPORT void orc_sc_naf(const uint8_t s[32], unsigned w, int8_t out[256]) { sc_non_adjacent_form(out, s, w); }
EXPORT void orc_sc_radix16(const uint8_t s[32], int8_t out[64]) { sc_as_radix_16(out, s); }
EXPORT void orc_sc_radix2w(const uint8_t s[32], unsigned w, int8_t out[64]) { sc_as_radix_2w(out, s, w); }
EXPORT void orc_sc_clamp(uint8_t b[32]) { sc_clamp_integer(b); }

/* ---------------- Edwards ---------------- */
static ge_basepoint_table *g_btab;
static ge_aniels *g_naf8B;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_tables(void) {
    g_btab = malloc(sizeof *g_btab); ge_basepoint_table_create(g_btab, ge_basepoint());   /* edwards.rs:1131-1141 */
    g_naf8B = malloc(64 * sizeof(ge_aniels)); ge_naf8_table_aniels(g_naf8B, ge_basepoint()); /* window.rs:266-276 */
}
static void ensure_tables(void) { pthread_once(&g_once, init_tables); }

EXPORT void orc_ed_basepoint(uint64_t out[20]) { p3_store(out, ge_basepoint()); }
EXPORT void orc_ed_identity(uint64_t out[20]) { p3_store(out, ge_identity()); }
EXPORT int orc_ed_decompress(const uint8_t in[32], uint64_t out[20]) { ge_p3 p; if (!ge_decompress(&p, in)) return 0; p3_store(out, p); return 1; }
EXPORT void orc_ed_compress(const uint64_t p[20], uint8_t out[32]) { ge_compress(out, p3_load(p)); }
/* edwards.rs:634-647 compress_batch_alloc */
EXPORT void orc_ed_compress_batch(const uint64_t *pts, size_t n, uint8_t *out) {
    fe *zs = malloc((n ? n : 1) * sizeof(fe)), *sc = malloc((n ? n : 1) * sizeof(fe));
    for (size_t i = 0; i < n; i++) zs[i] = p3_load(pts + 20 * i).Z;
    fe_invert_batch(zs, sc, n);
    for (size_t i = 0; i < n; i++) { ge_p3 p = p3_load(pts + 20 * i); ge_affine_compress(out + 32 * i, fe_mul(p.X, zs[i]), fe_mul(p.Y, zs[i])); }
    free(zs); free(sc);
}
EXPORT void orc_ed_add(const uint64_t a[20], const uint64_t b[20], uint64_t out[20]) { p3_store(out, ge_add(p3_load(a), p3_load(b))); }
EXPORT void orc_ed_sub(const uint64_t a[20], const uint64_t b[20], uint64_t out[20]) { p3_store(out, ge_sub(p3_load(a), p3_load(b))); }
EXPORT void orc_ed_neg(const uint64_t a[20], uint64_t out[20]) { p3_store(out, ge_neg(p3_load(a))); }
EXPORT void orc_ed_double(const uint64_t a[20], uint64_t out[20]) { p3_store(out, ge_dbl(p3_load(a))); }
EXPORT void orc_ed_mul_by_pow_2(const uint64_t a[20], unsigned k, uint64_t out[20]) { p3_store(out, ge_mul_by_pow_2(p3_load(a), k)); }
EXPORT int orc_ed_eq(const uint64_t a[20], const uint64_t b[20]) { return ge_eq(p3_load(a), p3_load(b)); }
EXPORT int orc_ed_is_identity(const uint64_t a[20]) { return ge_is_identity(p3_load(a)); }
EXPORT int orc_ed_is_small_order(const uint64_t a[20]) { return ge_is_small_order(p3_load(a)); }
EXPORT int orc_ed_is_torsion_free(const uint64_t a[20]) { /* edwards.rs:1435 */
    uint8_t l[32]; sc_to_bytes(l, sc_const(ORC_SC_L));
    return ge_is_identity(ge_variable_base_mul(p3_load(a), l));
}
EXPORT void orc_ed_mul(const uint64_t p[20], const uint8_t s[32], uint64_t out[20]) { p3_store(out, ge_variable_base_mul(p3_load(p), s)); }
EXPORT void orc_ed_mul_base(const uint8_t s[32], uint64_t out[20]) { ensure_tables(); p3_store(out, ge_mul_base_table(g_btab, s)); }
/* first entries of the generated tables, for the spot-check against u64/constants.rs:349- */
EXPORT void orc_ed_basepoint_table_entry(unsigned i, unsigned j, uint64_t out[15]) { ensure_tables(); memcpy(out, &g_btab->t[i][j], 120); }
EXPORT void orc_ed_naf8_basepoint_entry(unsigned i, uint64_t out[15]) { ensure_tables(); memcpy(out, &g_naf8B[i], 120); }
EXPORT void orc_ed_double_scalar_mul_basepoint(const uint8_t a[32], const uint64_t A[20], const uint8_t b[32], uint64_t out[20]) {
    ensure_tables(); p3_store(out, ge_vartime_double_base_mul(a, p3_load(A), b, g_naf8B));
}
/* which: 0 = size dispatch (edwards.rs:1025), 1 = Straus, 2 = Pippenger */
EXPORT void orc_ed_msm_vartime(const uint8_t *scalars, const uint64_t *points, size_t n, int which, uint64_t out[20]) {
    ge_p3 *pts = malloc((n ? n : 1) * sizeof(ge_p3));
    for (size_t i = 0; i < n; i++) pts[i] = p3_load(points + 20 * i);
    ge_p3 r = which == 1 ? ge_straus_vartime(scalars, pts, n) : which == 2 ? ge_pippenger_vartime(scalars, pts, n) : ge_multiscalar_mul_vartime(scalars, pts, n);
    p3_store(out, r); free(pts);
}
/* edwards.rs:574-590 */
EXPORT void orc_ed_to_montgomery(const uint64_t p[20], uint8_t out[32]) {
    ge_p3 P = p3_load(p);
    fe U = fe_add(P.Z, P.Y), W = fe_sub(P.Z, P.Y);
    fe_to_bytes(out, fe_mul(U, fe_invert(W)));
}

/* ---------------- Ristretto (ristretto.rs:266-345, :500-533, :822-829) ---------------- */
EXPORT int orc_ris_decompress(const uint8_t in[32], uint64_t out[20]) {
    fe s = fe_from_bytes(in);
    uint8_t chk[32]; fe_to_bytes(chk, s);
    if (memcmp(chk, in, 32) != 0 || fe_is_negative(s)) return 0;
    fe one = FE_ONE, ss = fe_sq(s);
    fe u1 = fe_sub(one, ss), u2 = fe_add(one, ss), u2_sqr = fe_sq(u2);
    fe v = fe_sub(fe_mul(fe_neg(fe_from_limbs(ORC_EDWARDS_D)), fe_sq(u1)), u2_sqr);
    fe I; int ok = fe_invsqrt(&I, fe_mul(v, u2_sqr));
    fe Dx = fe_mul(I, u2), Dy = fe_mul(I, fe_mul(Dx, v));
    fe x = fe_mul(fe_add(s, s), Dx);
    x = fe_cneg(x, fe_is_negative(x));
    fe y = fe_mul(u1, Dy), t = fe_mul(x, y);
    if (!ok || fe_is_negative(t) || fe_is_zero(y)) return 0;
    ge_p3 r = {x, y, one, t}; p3_store(out, r); return 1;
}
EXPORT void orc_ris_compress(const uint64_t p[20], uint8_t out[32]) {
    ge_p3 P = p3_load(p);
    fe X = P.X, Y = P.Y, Z = P.Z, T = P.T;
    fe u1 = fe_mul(fe_add(Z, Y), fe_sub(Z, Y)), u2 = fe_mul(X, Y);
    fe invsqrt; fe_invsqrt(&invsqrt, fe_mul(u1, fe_sq(u2)));
    fe i1 = fe_mul(invsqrt, u1), i2 = fe_mul(invsqrt, u2);
    fe z_inv = fe_mul(i1, fe_mul(i2, T)), den_inv = i2;
    fe sqrt_m1 = fe_from_limbs(ORC_SQRT_M1);
    fe iX = fe_mul(X, sqrt_m1), iY = fe_mul(Y, sqrt_m1);
    fe ench = fe_mul(i1, fe_from_limbs(ORC_INVSQRT_A_MINUS_D));
    int rotate = fe_is_negative(fe_mul(T, z_inv));
    X = fe_select(X, iY, rotate); Y = fe_select(Y, iX, rotate); den_inv = fe_select(den_inv, ench, rotate);
    Y = fe_cneg(Y, fe_is_negative(fe_mul(X, z_inv)));
    fe s = fe_mul(den_inv, fe_sub(Z, Y));
    s = fe_cneg(s, fe_is_negative(s));
    fe_to_bytes(out, s);
}
EXPORT int orc_ris_eq(const uint64_t a[20], const uint64_t b[20]) {
    ge_p3 A = p3_load(a), B = p3_load(b);
    return fe_eq(fe_mul(A.X, B.Y), fe_mul(A.Y, B.X)) | fe_eq(fe_mul(A.X, B.X), fe_mul(A.Y, B.Y));
}

/* ---------------- Montgomery / X25519 (montgomery.rs:183-211, :409-468) ---------------- */
typedef struct { fe U, W; } mont_pp;
static void mont_diff_add_and_double(mont_pp *P, mont_pp *Q, fe affine_PmQ) {
    fe t0 = fe_add(P->U, P->W), t1 = fe_sub(P->U, P->W), t2 = fe_add(Q->U, Q->W), t3 = fe_sub(Q->U, Q->W);
    fe t4 = fe_sq(t0), t5 = fe_sq(t1), t6 = fe_sub(t4, t5);
    fe t7 = fe_mul(t0, t3), t8 = fe_mul(t1, t2);
    fe t9 = fe_add(t7, t8), t10 = fe_sub(t7, t8);
    fe t11 = fe_sq(t9), t12 = fe_sq(t10);
    fe a24 = {{121666, 0, 0, 0, 0}};
    fe t13 = fe_mul(a24, t6);
    fe t14 = fe_mul(t4, t5), t15 = fe_add(t13, t5), t16 = fe_mul(t6, t15);
    fe t17 = fe_mul(affine_PmQ, t12);
    P->U = t14; P->W = t16; Q->U = t11; Q->W = t17;
}
/* s*u for a Scalar s with bit 255 clear: bits 254..0, MSB first (montgomery.rs:488-496) */
static void mont_mul(uint8_t out[32], const uint8_t u[32], const uint8_t s[32]) {
    fe affine_u = fe_from_bytes(u);
    mont_pp x0 = {FE_ONE, FE_ZERO}, x1 = {affine_u, FE_ONE};
    int prev = 0;
    for (int i = 254; i >= 0; i--) {
        int cur = (s[i >> 3] >> (i & 7)) & 1;
        if (prev ^ cur) { mont_pp t = x0; x0 = x1; x1 = t; }
        mont_diff_add_and_double(&x0, &x1, affine_u);
        prev = cur;
    }
    if (prev) { mont_pp t = x0; x0 = x1; x1 = t; }
    fe_to_bytes(out, fe_mul(x0.U, fe_invert(x0.W)));
}
EXPORT void orc_mont_mul(const uint8_t u[32], const uint8_t s[32], uint8_t out[32]) { mont_mul(out, u, s); }
/* x25519-dalek/src/x25519.rs:390 = MontgomeryPoint(u).mul_clamped(k) (montgomery.rs:150) */
EXPORT void orc_x25519(const uint8_t k[32], const uint8_t u[32], uint8_t out[32]) {
    uint8_t s[32]; memcpy(s, k, 32); sc_clamp_integer(s); mont_mul(out, u, s);
}

/* ---------------- hashes ---------------- */
EXPORT void orc_sha512(const uint8_t *m, size_t n, uint8_t out[64]) { sha512_ctx c; sha512_init(&c); sha512_update(&c, m, n); sha512_final(&c, out); }
EXPORT void orc_sha3_256(const uint8_t *m, size_t n, uint8_t out[32]) { orc_sha3_256_impl(m, n, out); }
/* Merlin conformance shape: new(label); append_message(l1, m); challenge_bytes(l2, out) */
EXPORT void orc_merlin_simple(const char *proto, const char *l1, const uint8_t *m, size_t n, const char *l2, uint8_t *out, size_t outlen) {
    merlin_transcript t; merlin_new(&t, proto); merlin_append_message(&t, l1, m, n); merlin_challenge_bytes(&t, l2, out, outlen);
}

/* A STROBE-128 op script: pins every operation Merlin uses (incl. KEY with arbitrary data, which the batch transcript only
   reaches with 32 zero bytes) against the independent reference in tests/pyref.py.  ops: nops x 6 bytes
   {op (0 meta_ad, 1 ad, 2 prf, 3 key), more, LE32 length}; data: the absorbed bytes, concatenated; out: the PRF outputs. */
EXPORT void orc_strobe_script(const uint8_t *proto, size_t proto_len, const uint8_t *ops, size_t nops, const uint8_t *data, uint8_t *out) {
    strobe128 s; strobe_new(&s, proto, proto_len);
    for (size_t i = 0; i < nops; i++) {
        const uint8_t *o = ops + 6 * i;
        size_t n = (size_t)o[2] | ((size_t)o[3] << 8) | ((size_t)o[4] << 16) | ((size_t)o[5] << 24);
        switch (o[0]) {
        case 0: strobe_meta_ad(&s, data, n, o[1]); data += n; break;
        case 1: strobe_ad(&s, data, n, o[1]); data += n; break;
        case 2: strobe_prf(&s, out, n, o[1]); out += n; break;
        default: strobe_key(&s, data, n, o[1]); data += n; break;
        }
    }
}

/* ---------------- Ed25519 ---------------- */
enum { ST_OK = 0, ST_NONE = 1, ST_SCALAR_FORMAT = 2, ST_VERIFY = 3, ST_ARRAY_LENGTH = 4 };

static void hram_hash(uint8_t out[64], const uint8_t R[32], const uint8_t A[32], const uint8_t *m, size_t n) {
    sha512_ctx c; sha512_init(&c); sha512_update(&c, R, 32); sha512_update(&c, A, 32); sha512_update(&c, m, n); sha512_final(&c, out);
}
/* signing.rs:878-905 (ExpandedSecretKey::raw_sign) + hazmat.rs expanded key; RFC 8032 5.1.5/5.1.6 */
static void expand_secret(const uint8_t sk[32], uint8_t a[32], uint8_t prefix[32]) {
    uint8_t h[64]; orc_sha512(sk, 32, h);
    memcpy(a, h, 32); sc_clamp_integer(a); memcpy(prefix, h + 32, 32);
}
EXPORT void orc_ed25519_pubkey(const uint8_t sk[32], uint8_t pk[32]) {
    ensure_tables();
    uint8_t a[32], prefix[32]; expand_secret(sk, a, prefix);
    ge_compress(pk, ge_mul_base_table(g_btab, a));   /* verifying.rs:97-101 mul_base_clamped */
}
EXPORT void orc_ed25519_sign(const uint8_t sk[32], const uint8_t *m, size_t n, uint8_t sig[64]) {
    ensure_tables();
    uint8_t a[32], prefix[32], pk[32], h[64], rb[32], kb[32];
    expand_secret(sk, a, prefix);
    ge_compress(pk, ge_mul_base_table(g_btab, a));
    sha512_ctx c; sha512_init(&c); sha512_update(&c, prefix, 32); sha512_update(&c, m, n); sha512_final(&c, h);
    sc52 r = sc_from_bytes_wide(h); sc_to_bytes(rb, r);
    ge_compress(sig, ge_mul_base_table(g_btab, rb));
    hram_hash(h, sig, pk, m, n);
    sc52 k = sc_from_bytes_wide(h); sc_to_bytes(kb, k);
    sc52 as = sc_from_bytes_mod_order(a);           /* scalar = from_bytes_mod_order(clamped) */
    sc_to_bytes(sig + 32, sc_add(sc_mul(k, as), r));
}
/* verifying.rs:203-214 raw_verify / :549-556 RCompute::finish */
EXPORT int orc_ed25519_verify(const uint8_t pk[32], const uint8_t *m, size_t n, const uint8_t sig[64]) {
    ensure_tables();
    ge_p3 A; if (!ge_decompress(&A, pk)) return ST_NONE;          /* VerifyingKey::from_bytes */
    if (!sc_is_canonical_bytes(sig + 32)) return ST_SCALAR_FORMAT;
    uint8_t h[64], kb[32], Rcheck[32];
    hram_hash(h, sig, pk, m, n); sc_to_bytes(kb, sc_from_bytes_wide(h));
    ge_compress(Rcheck, ge_vartime_double_base_mul(kb, ge_neg(A), sig + 32, g_naf8B));
    return memcmp(Rcheck, sig, 32) == 0 ? ST_OK : ST_VERIFY;
}
/* verifying.rs:359-382 */
EXPORT int orc_ed25519_verify_strict(const uint8_t pk[32], const uint8_t *m, size_t n, const uint8_t sig[64]) {
    ensure_tables();
    ge_p3 A, R; if (!ge_decompress(&A, pk)) return ST_NONE;
    if (!sc_is_canonical_bytes(sig + 32)) return ST_SCALAR_FORMAT;
    if (!ge_decompress(&R, sig)) return ST_VERIFY;
    if (ge_is_small_order(R) || ge_is_small_order(A)) return ST_VERIFY;
    uint8_t h[64], kb[32], Rcheck[32];
    hram_hash(h, sig, pk, m, n); sc_to_bytes(kb, sc_from_bytes_wide(h));
    ge_compress(Rcheck, ge_vartime_double_base_mul(kb, ge_neg(A), sig + 32, g_naf8B));
    return memcmp(Rcheck, sig, 32) == 0 ? ST_OK : ST_VERIFY;
}
/* ---- Ed25519ph / Ed25519ctx: the prehashed variants (RFC 8032 5.1 with dom2; verifying.rs:230-257 raw_verify_prehashed, :424-461
 * verify_prehashed_strict, :520-534 RCompute::new with prehash_ctx = Some(ctx); signing.rs:917-976 raw_sign_prehashed).
 * dom2 = "SigEd25519 no Ed25519 collisions" || 0x01 || len(ctx) || ctx is hashed BEFORE R || A (resp. before the hash prefix of the nonce);
 * the message is the 64-byte SHA-512 of the original message (the reference takes the digest state and finalises it itself). */
static void dom2_update(sha512_ctx *c, const uint8_t *ctx, size_t ctxlen) {
    const uint8_t two[2] = {1, (uint8_t)ctxlen};
    sha512_update(c, (const uint8_t *)"SigEd25519 no Ed25519 collisions", 32); sha512_update(c, two, 2); sha512_update(c, ctx, ctxlen);
}
static void hram_hash_ph(uint8_t out[64], const uint8_t *ctx, size_t ctxlen, const uint8_t R[32], const uint8_t A[32], const uint8_t ph[64]) {
    sha512_ctx c; sha512_init(&c); dom2_update(&c, ctx, ctxlen); sha512_update(&c, R, 32); sha512_update(&c, A, 32); sha512_update(&c, ph, 64); sha512_final(&c, out);
}
#define ST_PREHASHED_CONTEXT_LENGTH 5      /* errors.rs InternalError::PrehashedContextLength */
EXPORT int orc_ed25519_sign_prehashed(const uint8_t sk[32], const uint8_t ph[64], const uint8_t *ctx, size_t ctxlen, uint8_t sig[64]) {
    ensure_tables();
    if (ctxlen > 255) return ST_PREHASHED_CONTEXT_LENGTH;      /* signing.rs:931-933 */
    uint8_t a[32], prefix[32], pk[32], h[64], rb[32], kb[32];
    expand_secret(sk, a, prefix);
    ge_compress(pk, ge_mul_base_table(g_btab, a));
    sha512_ctx c; sha512_init(&c); dom2_update(&c, ctx, ctxlen); sha512_update(&c, prefix, 32); sha512_update(&c, ph, 64); sha512_final(&c, h);     /* :952-958 */
    sc52 r = sc_from_bytes_wide(h); sc_to_bytes(rb, r);
    ge_compress(sig, ge_mul_base_table(g_btab, rb));
    hram_hash_ph(h, ctx, ctxlen, sig, pk, ph);                                                                                                    /* :963-970 */
    sc52 k = sc_from_bytes_wide(h); sc_to_bytes(kb, k);
    sc52 as = sc_from_bytes_mod_order(a);
    sc_to_bytes(sig + 32, sc_add(sc_mul(k, as), r));
    return ST_OK;
}
EXPORT int orc_ed25519_verify_prehashed(const uint8_t pk[32], const uint8_t ph[64], const uint8_t *ctx, size_t ctxlen, const uint8_t sig[64], int strict) {
    ensure_tables();
    if (ctxlen > 255) return ST_PREHASHED_CONTEXT_LENGTH;      /* (the reference debug_asserts; a release build would truncate the length byte) */
    ge_p3 A, R; if (!ge_decompress(&A, pk)) return ST_NONE;
    if (!sc_is_canonical_bytes(sig + 32)) return ST_SCALAR_FORMAT;
    if (strict) {                                              /* verifying.rs:443-451 */
        if (!ge_decompress(&R, sig)) return ST_VERIFY;
        if (ge_is_small_order(R) || ge_is_small_order(A)) return ST_VERIFY;
    }
    uint8_t h[64], kb[32], Rcheck[32];
    hram_hash_ph(h, ctx, ctxlen, sig, pk, ph); sc_to_bytes(kb, sc_from_bytes_wide(h));
    ge_compress(Rcheck, ge_vartime_double_base_mul(kb, ge_neg(A), sig + 32, g_naf8B));
    return memcmp(Rcheck, sig, 32) == 0 ? ST_OK : ST_VERIFY;
}

/* batch.rs:168-222: the transcript-derived 128-bit z_i.  hrams: n x 64, ss: n x 32, zs out: n x 16 */
EXPORT void orc_batch_transcript_zs(const uint8_t *hrams, const uint8_t *ss, size_t n, uint8_t *zs) {
    merlin_transcript t; merlin_new(&t, "ed25519 batch verification");
    for (size_t i = 0; i < n; i++) merlin_append_message(&t, "hram", hrams + 64 * i, 64);
    for (size_t i = 0; i < n; i++) merlin_append_message(&t, "sig.s", ss + 32 * i, 32);
    merlin_rng_finalize_zero(&t);
    for (size_t i = 0; i < n; i++) merlin_rng_fill(&t, zs + 16 * i, 16);
}

/* batch.rs:146-251.  msgs concatenated, msg_off[n+1]; pks n x 32 (compressed; the reference's
   VerifyingKey carries the decompressed point, built by from_bytes -> failure there is reported
   as ST_NONE before verify_batch could have been called).
   zs_override (n x 16) if non-NULL replaces the transcript z_i (the engine's "fast" z-mode). */
EXPORT int orc_ed25519_verify_batch_z(const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks, size_t n, const uint8_t *zs_override) {
    size_t m = 2 * n + 1;
    uint8_t *hrams = malloc(64 * (n ? n : 1)), *ss = malloc(32 * (n ? n : 1)), *zs = malloc(16 * (n ? n : 1));
    uint8_t *scalars = malloc(32 * m); ge_p3 *points = malloc(m * sizeof(ge_p3));
    int status = ST_OK;
    for (size_t i = 0; i < n; i++) if (!ge_decompress(&points[1 + n + i], pks + 32 * i)) { status = ST_NONE; goto done; }
    for (size_t i = 0; i < n; i++) { hram_hash(hrams + 64 * i, sigs + 64 * i, pks + 32 * i, msgs + msg_off[i], msg_off[i + 1] - msg_off[i]); memcpy(ss + 32 * i, sigs + 64 * i + 32, 32); }
    if (zs_override) memcpy(zs, zs_override, 16 * n); else orc_batch_transcript_zs(hrams, ss, n, zs);
    for (size_t i = 0; i < n; i++) if (!sc_is_canonical_bytes(ss + 32 * i)) { status = ST_SCALAR_FORMAT; goto done; }
    sc52 bcoef = SC_ZERO;
    for (size_t i = 0; i < n; i++) {
        uint8_t zb[32] = {0}; memcpy(zb, zs + 16 * i, 16);
        sc52 z = sc_from_bytes(zb), s = sc_from_bytes(ss + 32 * i), h = sc_from_bytes_wide(hrams + 64 * i);
        bcoef = sc_add(bcoef, sc_mul(z, s));
        memcpy(scalars + 32 * (1 + i), zb, 32);
        sc_to_bytes(scalars + 32 * (1 + n + i), sc_mul(h, z));
    }
    sc_to_bytes(scalars, sc_sub(SC_ZERO, bcoef));
    points[0] = ge_basepoint();
    for (size_t i = 0; i < n; i++) if (!ge_decompress(&points[1 + i], sigs + 64 * i)) { status = ST_VERIFY; goto done; }
    status = ge_is_identity(ge_multiscalar_mul_vartime(scalars, points, m)) ? ST_OK : ST_VERIFY;
done:
    free(hrams); free(ss); free(zs); free(scalars); free(points);
    return status;
}
EXPORT int orc_ed25519_verify_batch(const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks, size_t n) {
    return orc_ed25519_verify_batch_z(msgs, msg_off, sigs, pks, n, NULL);
}

/* ---------------- batch drivers (cpu_baseline leg of bench.py, and bulk parity tests) ----------
   Each splits [0,n) into `threads` contiguous slices, one pthread per slice. */
typedef struct { int kind; const uint8_t *a, *b; uint8_t *out; size_t lo, hi; uint8_t *out2; size_t mlen; } job;
static void *job_run(void *arg) {
    job *j = arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        switch (j->kind) {
        case 0: { ge_p3 p = ge_mul_base_table(g_btab, j->a + 32 * i); ge_compress(j->out + 32 * i, p); break; }  /* mul_base + compress */
        case 1: orc_x25519(j->a + 32 * i, j->b + 32 * i, j->out + 32 * i); break;
        case 2: { ge_p3 p; j->out[i] = (uint8_t)ge_decompress(&p, j->a + 32 * i); break; }
        case 3: orc_ed25519_pubkey(j->a + 32 * i, j->out + 32 * i); orc_ed25519_sign(j->a + 32 * i, j->b + j->mlen * i, j->mlen, j->out2 + 64 * i); break;
        }
    }
    return NULL;
}
static uint8_t *g_out2; static size_t g_mlen;   /* extra operands of job kind 3 (set by its only caller) */
static void run_jobs(int kind, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n, int threads) {
    ensure_tables();
    if (threads < 1) threads = 1;
    pthread_t *th = malloc(threads * sizeof *th); job *js = malloc(threads * sizeof *js);
    for (int t = 0; t < threads; t++) {
        js[t] = (job){kind, a, b, out, n * t / threads, n * (t + 1) / threads, g_out2, g_mlen};
        pthread_create(&th[t], NULL, job_run, &js[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th); free(js);
}
/* "All host cores" legs of bench.py's cpu_baseline (one independent slice per thread).  The reference's MSM and
   verify_batch are single-threaded calls; a caller with T cores would cut its terms / signatures into T slices, run the
   reference's own call on each (pippenger.rs / batch.rs:146 unchanged) and add the partial sums / AND the verdicts. */
typedef struct { const uint8_t *scalars; const uint64_t *points; size_t lo, hi; ge_p3 sum; } msm_job;
static void *msm_job_run(void *arg) {
    msm_job *j = arg; size_t n = j->hi - j->lo;
    ge_p3 *pts = malloc((n ? n : 1) * sizeof(ge_p3));
    for (size_t i = 0; i < n; i++) pts[i] = p3_load(j->points + 20 * (j->lo + i));
    j->sum = ge_multiscalar_mul_vartime(j->scalars + 32 * j->lo, pts, n);
    free(pts); return NULL;
}
EXPORT void orc_ed_msm_vartime_mt(const uint8_t *scalars, const uint64_t *points, size_t n, int threads, uint64_t out[20]) {
    ensure_tables();
    if (threads < 1) threads = 1;
    pthread_t *th = malloc(threads * sizeof *th); msm_job *js = malloc(threads * sizeof *js);
    for (int t = 0; t < threads; t++) { js[t] = (msm_job){scalars, points, n * t / threads, n * (t + 1) / threads, ge_identity()}; pthread_create(&th[t], NULL, msm_job_run, &js[t]); }
    ge_p3 r = ge_identity();
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); r = ge_add(r, js[t].sum); }
    p3_store(out, r); free(th); free(js);
}
typedef struct { const uint8_t *msgs; const uint64_t *off; const uint8_t *sigs, *pks; size_t lo, hi; int status; } vb_job;
static void *vb_job_run(void *arg) {
    vb_job *j = arg;
    j->status = orc_ed25519_verify_batch_z(j->msgs, j->off + j->lo, j->sigs + 64 * j->lo, j->pks + 32 * j->lo, j->hi - j->lo, NULL);
    return NULL;
}
/* worst slice verdict (every slice is its own verify_batch call with its own transcript); msg_off are absolute offsets into msgs */
EXPORT int orc_ed25519_verify_batch_mt(const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks, size_t n, int threads) {
    ensure_tables();
    if (threads < 1) threads = 1;
    pthread_t *th = malloc(threads * sizeof *th); vb_job *js = malloc(threads * sizeof *js);
    for (int t = 0; t < threads; t++) { js[t] = (vb_job){msgs, msg_off, sigs, pks, n * t / threads, n * (t + 1) / threads, 0}; pthread_create(&th[t], NULL, vb_job_run, &js[t]); }
    int st = ST_OK;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); if (js[t].status != ST_OK && st == ST_OK) st = js[t].status; }
    free(th); free(js); return st;
}
EXPORT void orc_mul_base_compress_batch(const uint8_t *scalars, size_t n, uint8_t *out, int threads) { run_jobs(0, scalars, NULL, out, n, threads); }
EXPORT void orc_x25519_batch(const uint8_t *k, const uint8_t *u, size_t n, uint8_t *out, int threads) { run_jobs(1, k, u, out, n, threads); }
EXPORT void orc_ed_decompress_ok_batch(const uint8_t *in, size_t n, uint8_t *ok, int threads) { run_jobs(2, in, NULL, ok, n, threads); }
/* keypairs + signatures over fixed-length messages (test-input generator for verify_batch) */
EXPORT void orc_ed25519_keygen_sign_batch(const uint8_t *seeds, const uint8_t *msgs, size_t mlen, size_t n, uint8_t *pks, uint8_t *sigs, int threads) {
    g_out2 = sigs; g_mlen = mlen; run_jobs(3, seeds, msgs, pks, n, threads);
}
