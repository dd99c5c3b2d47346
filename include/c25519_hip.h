/* c25519_hip.h -- C ABI of libc25519hip.so, the MI355X (gfx950) batched Curve25519 engine.
 *
 * Drop-in boundary for the hot path of dalek-cryptography/curve25519-dalek.  The
 * reference has no FFI; the seam these entry points serve is its internal backend switch
 * (curve25519-dalek/src/backend.rs:45-277) plus the three callers that bypass it
 * (edwards.rs:1192 mul_base, montgomery.rs:183 mul_bits_be, ed25519-dalek/src/batch.rs:146
 * verify_batch).  INTEGRATION.md shows the Rust `extern "C"` block and the BackendKind::Hip arm a
 * maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes; all buffers caller-owned, contiguous, little-endian; nothing is
 *     retained after return.
 *   - `_dev` entry points take DEVICE pointers (HBM) and enqueue on the context's HIP stream;
 *     those that return a verdict/point to the host synchronise that stream.  The un-suffixed
 *     twins take HOST pointers and move the data themselves: the batch is cut into chunks (passes), and chunk c+1 travels up
 *     on a copy stream while chunk c computes and chunk c-1 travels down, so a call costs about max(link, kernels), not
 *     their sum.  Any host memory works; pageable memory whose pages have been touched moves at the link rate on this
 *     platform, but an OUTPUT buffer that has never been written (a fresh allocation) pays first-touch page faults at
 *     ~5 GB/s: reuse output buffers, or take them from c25519_host_alloc.
 *   - return value: int32 status.  0 OK; 1 NONE (a point failed to decompress: the Rust side maps
 *     it to Option::None); 2 SCALAR_FORMAT; 3 VERIFY; 4 ARRAY_LENGTH (mirrors
 *     ed25519-dalek/src/errors.rs:21-42 InternalError); negative = -(hipError_t) runtime failure.
 *   - point formats (`fmt`): 0 = 32-byte CompressedEdwardsY (edwards.rs:175),
 *     1 = 32-byte CompressedRistretto (ristretto.rs:223),
 *     2 = 160-byte raw EdwardsPoint {X,Y,Z,T} x 5 x u64 radix-2^51 limbs (u64/field.rs:43-52).  The reference keeps
 *         limbs below 2^52 (its debug_assert!s, field.rs:162-166); INPUT limbs here may be ANY u64 values -- the
 *         element read is sum_i l_i 2^(51 i) mod p exactly (no truncation, no status: there is no limb pattern without a
 *         meaning), so unreduced sums are accepted as what they denote; OUTPUT limbs are canonical (< 2^51, value < p).
 *         (edwards.rs:390-395 is not repr(C); the Rust shim copies limb-by-limb into this layout).
 *   - scalars: 32 bytes little-endian, < 2^255 (Scalar invariant #1, scalar.rs:197-205); they need
 *     NOT be reduced mod l for point multiplication (clamped integers are legal).
 *   - device buffers of 32-, 64- and 160-byte items must be 16-byte aligned (hipMalloc / torch
 *     allocations are); message blobs may have any alignment.
 *   - a context is bound to one GPU and one stream; calls on one context must not overlap.
 *     Results never depend on which GPU ran them.
 */
#ifndef C25519_HIP_H
#define C25519_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C25519_OK 0
#define C25519_NONE 1
#define C25519_SCALAR_FORMAT 2
#define C25519_VERIFY 3
#define C25519_ARRAY_LENGTH 4
#define C25519_PREHASHED_CONTEXT_LENGTH 5   /* Ed25519ph: context longer than 255 octets (errors.rs InternalError::PrehashedContextLength) */

#define C25519_FMT_EDWARDS_Y 0
#define C25519_FMT_RISTRETTO 1
#define C25519_FMT_RAW160 2

/* ed25519_verify_batch z_mode: how the 128-bit batch coefficients z_i are derived.
 * C25519_Z_TRANSCRIPT (0, the default of every host-language wrapper): byte for byte the reference's derivation --
 *   ONE Merlin/STROBE-128 transcript over the whole batch (batch.rs:168-222, batch/transcript.rs), whatever n is and
 *   however many passes, contexts (ed25519_verify_batch_multi) or ranks (multi.verify_batch_sharded) share the batch, and
 *   ONE equation over the whole batch (batch.rs:235-250: the partial sums of passes / contexts / ranks are added).  It is
 *   a sequential sponge (about 1.7 Keccak-f per signature), so it runs on one host core and bounds the call near
 *   2-3 x 10^6 signatures/s; the curve arithmetic still runs on the GPU.
 * C25519_Z_DEVICE (1, explicit opt-in, NOT the reference's derivation and not a reviewed standard construction):
 *   z_i = a 16-byte quarter of BLAKE2b-512(root || LE64(i / 4)), read as sign-magnitude (uniform on the 2^128 - 1 integers
 *   -(2^127 - 1) .. 2^127 - 1), where root is the root of a hash tree over what the reference's transcript
 *   absorbs: H(R_i || A_i || M_i) -- SHA-512 as in Ed25519, taken mod l, 32 bytes: the batch equation only sees that residue
 *   (batch.rs:213-217) -- and the 32-byte s_i of every signature.  Every node is a plain unkeyed BLAKE2b-256 digest (RFC 7693):
 *   node = BLAKE2b-256(TAG || data), TAG = one 128-byte block: the string "c25519-hip/verify_batch/z-tree/v5", zero bytes, then
 *   level, number of inputs of the level and batch size as little-endian u64 (bytes 104 .. 127); level 0: data = (h_i mod l) || s_i of
 *   4 signatures (256 bytes, absent signatures = zero bytes); upper levels: data = 4 child nodes (128 bytes, absent = zero bytes).
 *   (Rounds 2-5 built the same tree from SHA-512's compression function, tag .../v4; the values differ.  BLAKE2b because the tree's
 *   levels are a chain of dependent compressions that one wavefront per SIMD executes, and BLAKE2b's is 12 rounds against 80:
 *   csrc/blake2b.h.)  tests/pyref.py device_zs restates the derivation with hashlib.blake2b.  Every z_i depends on every bit of the batch;
 *   honest batches give the same verdict in both modes; a batch containing an invalid signature passes with
 *   probability <= 2^-127.99 (reference: 2^-128) under the usual assumption that BLAKE2b (and SHA-512 in h_i) is
 *   collision resistant and its output unpredictable.  BINDING MARGIN of the tree leaves (since v4): the z_i bind (R_i, A_i, M_i) only
 *   through h_i mod l, a 252-bit value, where the reference's transcript absorbs the full 64-byte hash (batch.rs:191-199).  Two
 *   different (R, A, M) triples with equal h mod l would produce identical z_i for two different batch equations; finding such a pair
 *   is a birthday search on 252 bits, about 2^126 hash evaluations -- near, not at, the 128-bit level of the rest of the construction
 *   (the same order as the cross-pass birthday bound this library refuses elsewhere by checking sub-batches independently).  Callers
 *   that need the reference's full margin use C25519_Z_TRANSCRIPT.  Batches beyond ~1.5 x 2^20 signatures are checked as
 *   independent sub-batches (own tree, own z_i, own identity check).  The z_i VALUES differ from the reference's. */
#define C25519_Z_TRANSCRIPT 0
#define C25519_Z_DEVICE 1

/* c25519_ctx_create flags, bit 8: VARTIME_TABLES.  By default every entry point that replaces a CONSTANT-TIME function
 * of the reference -- mul_base (edwards.rs:918, :1192-1209), `&EdwardsPoint * &Scalar` (variable_base.rs), sign / keygen,
 * X25519 public keys -- gives what the reference's LookupTable::select gives (window.rs:54-76): no memory ADDRESS and no
 * BRANCH depends on the scalar.
 *   variable base: the 8-entry table of variable_base.rs, every entry read and the wanted one kept by selects.
 *   fixed base (round 5): radix-2^5 tables in LDS, 52 additions, and the digit never reaches an address at all -- lane j of each
 *   32-lane half of a wave reads the table entry of the signed digit value j - 16 (an address made of its LANE INDEX), and every
 *   lane then pulls the entry of ITS digit out of the registers of lane (half base + digit + 16) with ds_bpermute_b32, a
 *   register-to-register transfer through the LDS crossbar that reads no memory.  The digit is the permute's lane selector and
 *   nothing else; the sign is part of the selector (no conditional negation).  MEASURED before it was adopted
 *   (profiles/r05_instruction_rates.txt, c25519_microbench 50-67): throughput and latency of the permute are the same for
 *   identity, all-equal, random-within-the-half, pairs 32 lanes apart and two-source selector patterns (24.2 - 24.4 cycles per
 *   wave-instruction, 71 - 73 cycles of latency); a selector pattern that crosses the two halves at random is 16 % slower --
 *   which is why a lane's sources stay inside its own 32-lane half BY CONSTRUCTION (radix 2^6 over the whole wave was not
 *   built).  Round 6 added the many-to-one patterns -- groups of 2, 4, 8, 16, 32 lanes of a half pulling ONE source lane, what equal
 *   digits in neighbouring lanes make of the selector (c25519_microbench 80-89, profiles/r06_instruction_rates.txt).  This timing
 *   independence is a MEASURED property of gfx950 (the kernels are built for nothing else), not an architectural guarantee: a port to
 *   another GPU must repeat the probe, or use the full-window scan.  Asserted on the compiled code: tests/test_ct_isa.py (exactly 6 + 3 LDS reads and 30 permutes per window, no other
 *   memory access, no exec / vcc branch in the window loop).  The full-window scan of rounds 2-4 remains as k_mul_base<5, CT>
 *   behind the CT_FETCH knob of the tuning build (2.77 against 1.84 ms per 2^20 scalars).
 * With this flag those entry points use the fast tables instead -- fixed base: the radix-2^16 tables in HBM, 16
 * additions, 4x faster -- whose ADDRESSES depend on the scalar: set it only when every scalar handed to this context
 * is public (or call the *_vartime entry points).  The X25519 ladder is constant-time in either mode; MSM /
 * verify_batch / double_base are variable-time by definition, as in the reference. */
#define C25519_FLAG_VARTIME_TABLES 0x100u

typedef struct c25519_ctx c25519_ctx;

/* Create a context on HIP device `device` (>= 0).  Builds the fixed-base table and uploads it:
 * flags & 0x1f selects the fixed-base algorithm (results are identical):
 * 4, 5 or 6: one radix-2^w window table per digit position, the structure of the reference's
 * EdwardsBasepointTable (edwards.rs:1131-1141, :1246-1282) sized for the 160 KiB LDS;
 * 9: signed 9-tooth x 6-table comb in LDS (31 additions + 4 doublings per scalar, 147 KB);
 * 10 .. 20: the EdwardsBasepointTable structure with radix 2^w and the table in HBM, served by L2 / MALL:
 * ceil(256/w) additions with one 128-byte gather each, no doublings (table 1.7 MB at w = 10, 71 MB at w = 16);
 * 0 (default) = 16.  The table is computed on the device when the context is created.
 * flags & C25519_FLAG_VARTIME_TABLES: see above.  Returns NULL if there is no usable GPU: there is NO CPU fallback. */
c25519_ctx *c25519_ctx_create(int device, uint32_t flags);
void c25519_ctx_destroy(c25519_ctx *ctx);
/* Use an existing hipStream_t (e.g. PyTorch's current stream) instead of the context's own. */
int32_t c25519_ctx_set_stream(c25519_ctx *ctx, void *hip_stream);
/* Block until everything enqueued on the context's stream has finished. */
int32_t c25519_ctx_synchronize(c25519_ctx *ctx);
const char *c25519_last_error(const c25519_ctx *ctx);
/* Page-locked host memory (hipHostMalloc) for input / output buffers of the host-pointer entry points: DMA-able as is, never
 * pays a first-touch fault.  c25519_last_ffi_ms: wall-clock milliseconds of the most recent host-pointer call of this
 * context and the bytes it moved each way (either pointer may be NULL); -1 before the first such call. */
void *c25519_host_alloc(size_t bytes);
void c25519_host_free(void *p);
double c25519_last_ffi_ms(const c25519_ctx *ctx, uint64_t *h2d_bytes, uint64_t *d2h_bytes);
/* Release the workspaces earlier calls left allocated (a 2^24-term MSM keeps ~2.6 GB: gather records prepared ahead, the
 * normaliser's prefix products, staging copies of host-pointer calls).  Synchronises the context; the next call
 * re-allocates what it needs.  For long-lived contexts that see an occasional very large call. */
int32_t c25519_ctx_trim(c25519_ctx *ctx);
/* name of the kernel that c25519_phase_ms phase 0 (which = 0) / phase 3 (which = 1) of the latest entry point timed */
const char *c25519_last_kernel_name(const c25519_ctx *ctx, int which);
/* milliseconds the device spent in the most recent entry point's kernels (hipEvent pair on the
 * context's stream); valid after the call returned / the stream was synchronised. */
float c25519_last_kernel_ms(c25519_ctx *ctx);
/* Host clock of the phases of the latest synchronous MSM / verify_batch call on this context, microseconds since the call was entered:
 * out4[0] inputs staged and their upload enqueued (0 for the device-pointer forms), [1] every kernel enqueued, [2] results on the host (the last
 * kernel writes them into page-locked host memory and the host polls a sequence word: no copy engine, no interrupt), [3] folded and encoded. */
int32_t c25519_last_call_host_us(const c25519_ctx *ctx, double *out4);
/* Event counters of a context since its creation (diagnostics; tools/soak_small.py logs them).  Small and mid-size MSM / verify_batch calls (single passes up to 2^18 terms:
 * vartime_multiscalar_mul at the sizes of benches/dalek_benchmarks.rs:16, verify_batch of ed25519-dalek/benches/ed25519_benchmarks.rs:53) end with their last
 * kernel writing the record into page-locked host memory and the host polling a sequence word.  which = 0: calls whose record had not arrived after 2 ms of polling --
 * the host then blocked on the stream (a busy stream, a shared GPU); 1: calls whose record never arrived although the stream drained without error -- each was re-run
 * through the slot + copy path and returned its normal status (none observed: profiles/r06_soak_small.txt); 2: directly published calls.  Like
 * ed25519-dalek/src/batch.rs:146-251, a call returns or errs; it never waits on wall-clock alone. */
uint64_t c25519_ctx_counter(const c25519_ctx *ctx, int32_t which);

/* Per-call phase timing from a ring of hipEvents recorded on the launch streams (the last 64 calls of this context):
 * phase 0 = the dominant kernel of the call made `back` calls ago (0 = most recent) -- k_mul_base_*, k_x25519,
 * k_var_base, or k_accumulate for an MSM / verify_batch pass; phase 1 = the kernels after it (batched compression;
 * bucket reduction); for MSM / verify_batch passes also phase 2 = the whole pass and phase 3 = the decompression of
 * R_i (verify_batch).  Synchronises that call's last event.  Returns -1 if unavailable. */
float c25519_phase_ms(c25519_ctx *ctx, uint32_t back, int phase);
/* The same, summed over every pass of the most recent c25519_msm_* / ed25519_verify_batch* call (large inputs run as
 * several passes that alternate between two stream sets); *passes (may be NULL) receives the number of passes. */
float c25519_last_call_phase_ms(c25519_ctx *ctx, int phase, uint32_t *passes);

/* ---- fixed base: out[i] = scalars[i] * B ------------------------------------------------------
 * replaces EdwardsBasepointTable::mul_base / EdwardsPoint::mul_base (edwards.rs:918, :1192-1209),
 * followed by compress (edwards.rs:615 / :634 batch) when out_fmt is 0 or 1.
 * scalars: n x 32;  out: n x 32 (fmt 0/1) or n x 160 (fmt 2). */
int32_t c25519_mul_base_batch_dev(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, int out_fmt, uint8_t *d_out);
int32_t c25519_mul_base_batch(c25519_ctx *ctx, const uint8_t *scalars, uint64_t n, int out_fmt, uint8_t *out);
/* the same for scalars the caller declares PUBLIC: always the context's fast tables (variable-time table access). */
int32_t c25519_mul_base_batch_vartime_dev(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, int out_fmt, uint8_t *d_out);

/* EdwardsPoint::mul_base_clamped (edwards.rs:948-956): out[i] = clamp_integer(bytes[i]) * B -- the clamped integer is NOT
 * reduced mod l (scalar.rs:1407; legal for point multiplication, scalar.rs:197-222).  Secrets: constant-time tables unless the
 * context was created with C25519_FLAG_VARTIME_TABLES; the clamped copies are wiped. */
int32_t c25519_mul_base_clamped_batch_dev(c25519_ctx *ctx, const uint8_t *d_bytes, uint64_t n, int out_fmt, uint8_t *d_out);
int32_t c25519_mul_base_clamped_batch(c25519_ctx *ctx, const uint8_t *bytes, uint64_t n, int out_fmt, uint8_t *out);

/* ---- constant-time fixed-base tables for a CALLER'S point -----------------------------------------------------------------
 * replaces EdwardsBasepointTable::create(&P) (edwards.rs:1131-1141, generic over the basepoint; radices :1246-1292) and
 * RistrettoBasepointTable::create (ristretto.rs:1080-1110), then `&scalar * &table` = mul_base on that table
 * (edwards.rs:1192-1209).  create: point = 32 bytes (fmt 0 / 1) or 160 bytes (fmt 2), HOST pointer; the table (radix 2^5:
 * 52 windows x 17 affine Niels entries, the layout of the context's own constant-time table) is built once and stays in device
 * memory.  mul_table: out[i] = scalars[i] * P, ALWAYS with the constant-time lookup of the default fixed-base kernel (no address and
 * no branch depends on the scalar: see C25519_FLAG_VARTIME_TABLES above), whatever the context's flags -- these tables exist for secret scalars: Pedersen commitments
 * a*G + b*H, ElGamal / DH keys on a fixed generator.  out_fmt 0 / 1 / 2; fmt 1 gives RistrettoBasepointTable's result. */
typedef struct c25519_basetable c25519_basetable;
c25519_basetable *c25519_basetable_create(c25519_ctx *ctx, const uint8_t *point, int in_fmt);
void c25519_basetable_destroy(c25519_ctx *ctx, c25519_basetable *t);
int32_t c25519_mul_table_batch_dev(c25519_ctx *ctx, const c25519_basetable *t, const uint8_t *d_scalars, uint64_t n, int out_fmt, uint8_t *d_out);
int32_t c25519_mul_table_batch(c25519_ctx *ctx, const c25519_basetable *t, const uint8_t *scalars, uint64_t n, int out_fmt, uint8_t *out);

/* ---- X25519: out[i] = x25519(k[i], u[i]) -------------------------------------------------------
 * replaces x25519-dalek/src/x25519.rs:390 = MontgomeryPoint(u).mul_clamped(k) (montgomery.rs:150,
 * :183-211): k is clamped inside, bit 255 of u ignored, u >= p reduced, low-order u -> all-zero. */
int32_t c25519_x25519_batch_dev(c25519_ctx *ctx, const uint8_t *d_k, const uint8_t *d_u, uint64_t n, uint8_t *d_out);
int32_t c25519_x25519_batch(c25519_ctx *ctx, const uint8_t *k, const uint8_t *u, uint64_t n, uint8_t *out);
/* The same with SharedSecret::was_contributory (x25519-dalek/src/x25519.rs:335) as a batched flag: contributory[i] = 1 iff
 * out[i] is not all-zero (a low-order u[i] gives the all-zero shared secret, montgomery.rs:403-412).  MontgomeryPoint::
 * mul_clamped (montgomery.rs:150-162) is exactly this function: k is clamped inside. */
int32_t c25519_x25519_contributory_batch_dev(c25519_ctx *ctx, const uint8_t *d_k, const uint8_t *d_u, uint64_t n, uint8_t *d_out, uint8_t *d_contributory);
int32_t c25519_x25519_contributory_batch(c25519_ctx *ctx, const uint8_t *k, const uint8_t *u, uint64_t n, uint8_t *out, uint8_t *contributory);
/* X25519 public keys: out[i] = x25519(k[i], 9), computed the way x25519-dalek does it -- PublicKey::from(&secret) =
 * EdwardsPoint::mul_base_clamped(secret).to_montgomery() (x25519.rs:105-109, :255-259; edwards.rs:948, :574-590) --
 * i.e. through the fixed-base tables and the birational map (Z+Y)/(Z-Y), 12x cheaper than the ladder. */
int32_t c25519_x25519_base_batch_dev(c25519_ctx *ctx, const uint8_t *d_k, uint64_t n, uint8_t *d_out);
int32_t c25519_x25519_base_batch(c25519_ctx *ctx, const uint8_t *k, uint64_t n, uint8_t *out);

/* ---- (de)compression ------------------------------------------------------------------------------
 * decompress: CompressedEdwardsY::decompress (edwards.rs:211-258, ZIP-215 rules) for in_fmt 0,
 * CompressedRistretto::decompress (ristretto.rs:266-345) for in_fmt 1.
 * in: n x 32; out: n x 160 raw points (unspecified where ok[i] == 0); ok: n bytes (1 = Some).
 * Returns C25519_NONE if any ok[i] == 0, else C25519_OK (ok[] tells which). */
int32_t c25519_decompress_batch_dev(c25519_ctx *ctx, const uint8_t *d_in, uint64_t n, int in_fmt, uint8_t *d_out, uint8_t *d_ok);
int32_t c25519_decompress_batch(c25519_ctx *ctx, const uint8_t *in, uint64_t n, int in_fmt, uint8_t *out, uint8_t *ok);
/* compress: EdwardsPoint::compress_batch (edwards.rs:621-647) for out_fmt 0,
 * RistrettoPoint::compress (ristretto.rs:500-533) for out_fmt 1.  in: n x 160 raw; out: n x 32. */
int32_t c25519_compress_batch_dev(c25519_ctx *ctx, const uint8_t *d_in, uint64_t n, int out_fmt, uint8_t *d_out);
int32_t c25519_compress_batch(c25519_ctx *ctx, const uint8_t *in, uint64_t n, int out_fmt, uint8_t *out);

/* Edwards -> Montgomery u-coordinate, batched: EdwardsPoint::to_montgomery_batch (edwards.rs:595-612),
 * u = (Z+Y)/(Z-Y) with one shared inversion per lane-chunk; the identity maps to u = 0 (edwards.rs:574-590).
 * in: n x 160 raw points; out: n x 32 MontgomeryPoint bytes. */
int32_t c25519_to_montgomery_batch_dev(c25519_ctx *ctx, const uint8_t *d_in, uint64_t n, uint8_t *d_out);
int32_t c25519_to_montgomery_batch(c25519_ctx *ctx, const uint8_t *in, uint64_t n, uint8_t *out);

/* ---- variable-time multiscalar multiplication: out = sum scalars[i] * points[i] -----------------
 * replaces backend::pippenger_optional_multiscalar_mul / straus_optional_multiscalar_mul
 * (backend.rs:79, :224; edwards.rs:1002-1031; ristretto.rs:984).  points: n x 32 (fmt 0/1) or
 * n x 160 (fmt 2).  `out` is a HOST pointer (32 or 160 bytes) in both variants; the call
 * synchronises.  Returns C25519_NONE iff some point fails to decompress (Option::None of the
 * reference); n == 0 gives the identity. */
int32_t c25519_msm_vartime_dev(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out);
int32_t c25519_msm_vartime(c25519_ctx *ctx, const uint8_t *scalars, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out);
/* Multi-GPU building block: this rank's partial sum as a raw 160-byte point (no final compress),
 * and the fold of `count` partial points gathered from all ranks.  Both are
 * deterministic functions of their inputs.  c25519_fold_partials is host arithmetic over `count`
 * points (count = number of GPUs) and accepts ctx == NULL. */
int32_t c25519_msm_partial_dev(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, uint8_t *out160);
int32_t c25519_fold_partials(c25519_ctx *ctx, const uint8_t *partials160, uint64_t count, int out_fmt, uint8_t *out);

/* The same decomposition WITHOUT the partial sum ever visiting the host (what curve25519-dalek_amd/multi.py runs, one
 * process per GPU): c25519_msm_partial_record_dev only ENQUEUES on the context's stream and leaves a fixed-size RECORD in
 * device memory -- the window column sums of this rank's terms (the reference's `columns`, pippenger.rs:146-151, before
 * the Horner fold :159), its counters (a point that does not decode, a scalar with bit 255 set) and the number of terms
 * and the window width the window layout was derived from.  The exchange step is ONE all_gather of C25519_PARTIAL_RECORD_BYTES per rank
 * (RCCL, device to device), one copy to the host, and c25519_fold_partial_records: records with the same layout are
 * added column by column and folded once; the others are folded one by one (host arithmetic over count x <= 56 points;
 * ctx may be NULL).  Returns C25519_NONE iff any rank saw a point that does not decode.  The result is bit-identical to
 * c25519_msm_vartime over all terms on one context.  d_record: device pointer, 16-byte aligned. */
#define C25519_PARTIAL_RECORD_BYTES 9024
int32_t c25519_msm_partial_record_dev(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, uint8_t *d_record);
int32_t c25519_fold_partial_records(c25519_ctx *ctx, const uint8_t *records, uint64_t count, int out_fmt, uint8_t *out);
/* A record that holds a given 160-byte point: lets a participant whose partial sum was computed elsewhere (the host-pointer
 * entry points, another library) join the same fold.  status: C25519_OK or C25519_NONE; counters8 (may be NULL): eight
 * u32 counters in the order of the record's flags.  Host arithmetic, no context. */
int32_t c25519_partial_record_pack(const uint8_t *point160, int32_t status, const uint32_t *counters8, uint8_t *record);

/* One process driving several GPUs (SURVEY.md 8e for a host without torch / RCCL, e.g. the Rust shim): the terms are cut
 * into nctx contiguous shards, shard r runs the whole single-GPU path on ctxs[r] (contexts on different devices, or
 * several on one) from its own host thread, and the nctx partial sums are folded on the host -- the one exchange step of
 * the path, 160 bytes per context.  HOST pointers; the result is identical to c25519_msm_vartime on one context.
 * (With one process PER GPU the same decomposition is c25519_msm_partial_dev + an all_gather of the partials over
 * RCCL + c25519_fold_partials: curve25519-dalek_amd/multi.py.) */
int32_t c25519_msm_vartime_multi(c25519_ctx **ctxs, int32_t nctx, const uint8_t *scalars, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out);

/* ---- ed25519_dalek::verify_batch (ed25519-dalek/src/batch.rs:146-251) ---------------------------
 * msgs: concatenated messages; msg_off: n+1 offsets into msgs (u64); sigs: n x 64; pks: n x 32.
 * Error precedence as in the reference: a public key that does not decompress -> C25519_NONE (the
 * reference fails earlier, at VerifyingKey::from_bytes, verifying.rs:167); any non-canonical s ->
 * SCALAR_FORMAT; any R that does not decompress, or a non-identity result -> VERIFY.
 * (ARRAY_LENGTH is raised by the host-language wrapper, which owns the three lengths.)
 * msg_off must be non-decreasing with msg_off[n] <= msgs_len (the bytes readable at msgs); otherwise the call returns
 * -(hipErrorInvalidValue) and no byte outside [msgs, msgs + msgs_len) is read. */
int32_t ed25519_verify_batch_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                 const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, uint32_t z_mode);
int32_t ed25519_verify_batch(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off,
                             const uint8_t *sigs, const uint8_t *pks, uint64_t n, uint32_t z_mode);
/* verify_batch over several contexts / GPUs from one process, contiguous shards of the signatures.  C25519_Z_TRANSCRIPT: the
 * reference's ONE transcript over the whole batch and its single equation -- every context hashes its shard, the host runs
 * the transcript once over all H(R||A||M) and s, every context evaluates its share of the equation with its z_i, and the
 * partial sums are folded into one identity check: z_i, equation and verdict are those of the reference whatever nctx is.
 * C25519_Z_DEVICE: each shard is its own random linear combination; the verdict is the worst shard verdict in the
 * reference's precedence. */
int32_t ed25519_verify_batch_multi(c25519_ctx **ctxs, int32_t nctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks,
                                   uint64_t n, uint32_t z_mode);
/* The same check for callers that hold VerifyingKey values: the reference's VerifyingKey keeps the decompressed
 * point beside the 32 key bytes (verifying.rs:64-71, built once by from_bytes :167-175), so its verify_batch
 * never decompresses A_i (batch.rs:236 uses pk.point directly).  pk_points: n x 160 raw
 * EdwardsPoints matching pks (e.g. from c25519_decompress_batch), or NULL = decompress the key bytes here.
 * CONTRACT (the reference's VerifyingKey type invariant, verifying.rs:64-71): pk_points[i] MUST be the decompression of
 * pks[i].  The hash uses the bytes and the group equation the point; a mismatched pair verifies against a key that
 * was not hashed.  The library does not re-check it (that would be the decompression this entry point exists to skip);
 * host-language wrappers must only build the pair through from_bytes (dalek.VerifyingKey does). */
int32_t ed25519_verify_batch_keys_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                      const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, uint64_t n, uint32_t z_mode);
int32_t ed25519_verify_batch_keys(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off,
                                  const uint8_t *sigs, const uint8_t *pks, const uint8_t *pk_points, uint64_t n, uint32_t z_mode);

/* ---- verify_batch across ranks with the reference's ONE transcript (C25519_Z_TRANSCRIPT; SURVEY.md 8e: "the sequential
 * transcript runs once on the host and the z_i are scattered") -- the building blocks multi.verify_batch_sharded and
 * ed25519_verify_batch_multi are made of:
 *   1. ed25519_batch_hram_dev      this rank's H(R_i || A_i || M_i) (batch.rs:179-191) to device memory: n x 64 bytes followed
 *                                  by a 64-byte trailer of counters (u32 [0] signatures with a non-canonical s, [1] bad
 *                                  message offsets); enqueue only.  d_hram needs room for n x 64 + 64 bytes.
 *   2. (exchange)                  every hram and every s reaches whoever runs the transcript -- one all_gather
 *   3. ed25519_batch_transcript_zs the reference's Merlin/STROBE transcript over the WHOLE batch (batch.rs:168-222), HOST
 *                                  pointers, no context: hram n x 64, sigs n x 64 (s = bytes 32..63) -> z16 n x 16
 *   4. ed25519_verify_batch_record_dev   this rank's share of the batch equation (batch.rs:213-244) with ITS z_i (d_z16) and
 *                                  its hram buffer from step 1 (trailer included), as a partial-result record in device
 *                                  memory (layout and size as c25519_msm_partial_record_dev); enqueue only
 *   5. (exchange) + ed25519_fold_verify_records   the records of all ranks -> the reference's single identity check
 *                                  (batch.rs:246-250) and error precedence; HOST pointer, ctx may be NULL.
 * The z_i, the equation and the verdict are those of the reference on the whole batch, whatever the number of ranks. */
int32_t ed25519_batch_hram_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len, const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n,
                               uint8_t *d_hram);
int32_t ed25519_batch_transcript_zs(const uint8_t *hram, const uint8_t *sigs, uint64_t n, uint8_t *z16);
int32_t ed25519_verify_batch_record_dev(c25519_ctx *ctx, const uint8_t *d_sigs, const uint8_t *d_pks, const uint8_t *d_pk_points, const uint8_t *d_hram, const uint8_t *d_z16,
                                        uint64_t n, uint8_t *d_record);
int32_t ed25519_fold_verify_records(c25519_ctx *ctx, const uint8_t *records, uint64_t count);

/* diagnostics for the tests that pin the z derivation: the z_i a batch of n <= 1.5 x 2^20 signatures gets, 16 bytes each
 * to the HOST buffer out_z16 (HOST pointers throughout).  z_mode 0: little-endian u128, the reference's values;
 * z_mode 1: bit 127 = sign, bits 0..126 = magnitude; z_mode 2 (this entry point only): the values of z_mode 1 computed by the host
 * restatement of the derivation that batches of at most 128 signatures use (no kernel runs) -- equal to z_mode 1 byte for byte. */
int32_t c25519_debug_batch_zs(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks, uint64_t n,
                              uint32_t z_mode, uint8_t *out_z16);

/* diagnostics for kernel traces: the SORT of one MSM pass alone (n scalars on the device, 4096 <= n <= 2^22; the window layout of
 * `layout_terms` terms, 0 = of n), `reps` times back to back on the context's stream, then a synchronisation.  No result:
 * tools/sort_only.py runs it under rocprofv3 to time the sort kernels without an accumulation beside them. */
int32_t c25519_debug_sort(c25519_ctx *ctx, const uint8_t *d_scalars, uint64_t n, uint64_t layout_terms, int32_t reps);

/* ---- variable base: out[i] = scalars[i] * points[i] ------------------------------------------------
 * replaces backend::variable_base_mul (backend.rs:253 -> scalar_mul/variable_base.rs:11-47;
 * `&EdwardsPoint * &Scalar`, edwards.rs:890-911), radix-16 fixed windows, one pair per lane.
 * points: n x 32 (fmt 0) or n x 160 (fmt 2); out: n x 32 (fmt 0) or n x 160 (fmt 2);
 * ok (may be NULL): n bytes, 0 where a compressed point did not decode (that output is unspecified). */
int32_t c25519_mul_batch_dev(c25519_ctx *ctx, const uint8_t *d_scalars, const uint8_t *d_points, uint64_t n, int in_fmt, int out_fmt, uint8_t *d_out, uint8_t *d_ok);
int32_t c25519_mul_batch(c25519_ctx *ctx, const uint8_t *scalars, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out, uint8_t *ok);

/* EdwardsPoint::mul_clamped (edwards.rs:932-946): out[i] = clamp_integer(bytes[i]) * points[i] (not reduced mod l). */
int32_t c25519_mul_clamped_batch_dev(c25519_ctx *ctx, const uint8_t *d_bytes, const uint8_t *d_points, uint64_t n, int in_fmt, int out_fmt, uint8_t *d_out, uint8_t *d_ok);
int32_t c25519_mul_clamped_batch(c25519_ctx *ctx, const uint8_t *bytes, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out, uint8_t *ok);

/* ---- order checks: flags[i] of point i ----------------------------------------------------------------------
 * replaces EdwardsPoint::is_small_order (edwards.rs:1405-1407: mul_by_cofactor().is_identity()), EdwardsPoint::is_torsion_free
 * (edwards.rs:1435-1437: (self * BASEPOINT_ORDER).is_identity()) and, through them, VerifyingKey::is_weak
 * (ed25519-dalek/src/verifying.rs:192-194), batched.  in_fmt: C25519_FMT_EDWARDS_Y or C25519_FMT_RAW160.
 * flags[i]: C25519_POINT_DECODES (always set for raw points) | C25519_POINT_SMALL_ORDER | C25519_POINT_TORSION_FREE;
 * a point that does not decode has flags[i] = 0.  `which` selects the checks to run (C25519_POINT_SMALL_ORDER, C25519_POINT_TORSION_FREE
 * or both; the torsion check is a full variable-base multiplication by l).  Variable time: the points are public keys / signature parts. */
#define C25519_POINT_DECODES 1
#define C25519_POINT_SMALL_ORDER 2
#define C25519_POINT_TORSION_FREE 4
int32_t c25519_point_order_checks_batch_dev(c25519_ctx *ctx, const uint8_t *d_points, uint64_t n, int in_fmt, int which, uint8_t *d_flags);
int32_t c25519_point_order_checks_batch(c25519_ctx *ctx, const uint8_t *points, uint64_t n, int in_fmt, int which, uint8_t *flags);

/* ---- double base: out[i] = a[i] * A[i] + b[i] * B ---------------------------------------------------------
 * replaces backend::vartime_double_base_mul (backend.rs:267 -> scalar_mul/vartime_double_base.rs:23-72;
 * EdwardsPoint::vartime_double_scalar_mul_basepoint, edwards.rs:1099-1106), the single-signature kernel,
 * batched: a radix-16 ladder on A_i plus the fixed-base table on B.  Formats and `ok` as c25519_mul_batch. */
int32_t c25519_double_base_batch_dev(c25519_ctx *ctx, const uint8_t *d_a, const uint8_t *d_A, const uint8_t *d_b, uint64_t n, int in_fmt, int out_fmt, uint8_t *d_out, uint8_t *d_ok);
int32_t c25519_double_base_batch(c25519_ctx *ctx, const uint8_t *a, const uint8_t *A, const uint8_t *b, uint64_t n, int in_fmt, int out_fmt, uint8_t *out, uint8_t *ok);

/* ---- per-signature verification: status[i] for every signature ----------------------------------------
 * replaces VerifyingKey::verify (verifying.rs:565 -> raw_verify :203 -> RCompute::finish :549-556 over
 * vartime_double_base::mul, scalar_mul/vartime_double_base.rs:23-72) and, with strict != 0,
 * VerifyingKey::verify_strict (verifying.rs:359-382: R must decode, R and A must not be of small
 * order).  status[i]: 0 OK, 1 key does not decode (VerifyingKey::from_bytes), 2 SCALAR_FORMAT, 3 VERIFY.
 * This is what locates the bad signature after a failed batch. */
int32_t ed25519_verify_each_dev(c25519_ctx *ctx, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                                const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, int strict, uint8_t *d_status);
int32_t ed25519_verify_each(c25519_ctx *ctx, const uint8_t *msgs, const uint64_t *msg_off, const uint8_t *sigs, const uint8_t *pks,
                            uint64_t n, int strict, uint8_t *status);

/* Ed25519ph / Ed25519ctx, per signature (RFC 8032 5.1; replaces VerifyingKey::verify_prehashed ed25519-dalek/src/verifying.rs:284 ->
 * raw_verify_prehashed :230-257, verify_prehashed_strict :424-461, challenge by RCompute::new :520-534 with prehash_ctx = Some(ctx)):
 * hram_i = SHA-512("SigEd25519 no Ed25519 collisions" || 0x01 || len(ctx) || ctx || R_i || A_i || PH(M_i)).
 * prehashes: n x 64 bytes, PH(M_i) = SHA-512(M_i) (the reference takes the digest state and finalises it); context: HOST pointer in both
 * forms, 0 .. 255 bytes (NULL allowed when context_len = 0 -- the reference's `None`), ONE context for the batch; a longer context returns
 * C25519_PREHASHED_CONTEXT_LENGTH.  status per signature as ed25519_verify_each. */
int32_t ed25519_verify_each_prehashed_dev(c25519_ctx *ctx, const uint8_t *d_prehashes, const uint8_t *context, uint32_t context_len,
                                          const uint8_t *d_sigs, const uint8_t *d_pks, uint64_t n, int strict, uint8_t *d_status);
int32_t ed25519_verify_each_prehashed(c25519_ctx *ctx, const uint8_t *prehashes, const uint8_t *context, uint32_t context_len, const uint8_t *sigs, const uint8_t *pks,
                                      uint64_t n, int strict, uint8_t *status);

/* ---- batched key generation and signing (consumers of the fixed-base kernel) ---------------------------
 * keygen: pk_i = compress(clamp(SHA-512(seed_i)[0..32]) * B)   (verifying.rs:97-101, RFC 8032 5.1.5)
 * sign:   RFC 8032 5.1.6 as in signing.rs:878-905; also returns the public keys.  seeds: n x 32. */
int32_t ed25519_keygen_batch_dev(c25519_ctx *ctx, const uint8_t *d_seeds, uint64_t n, uint8_t *d_pks);
int32_t ed25519_sign_batch_dev(c25519_ctx *ctx, const uint8_t *d_seeds, const uint8_t *d_msgs, const uint64_t *d_msg_off, uint64_t msgs_len,
                               uint64_t n, uint8_t *d_pks, uint8_t *d_sigs);
int32_t ed25519_sign_batch(c25519_ctx *ctx, const uint8_t *seeds, const uint8_t *msgs, const uint64_t *msg_off, uint64_t n, uint8_t *pks, uint8_t *sigs);
/* Ed25519ph signing (SigningKey::sign_prehashed signing.rs:312 -> raw_sign_prehashed :917-976): r = H(dom2 || prefix || PH(M)), k = H(dom2 || R || A ||
 * PH(M)); prehashes n x 64 bytes, context a HOST pointer of 0 .. 255 bytes (as ed25519_verify_each_prehashed). */
int32_t ed25519_sign_batch_prehashed_dev(c25519_ctx *ctx, const uint8_t *d_seeds, const uint8_t *d_prehashes, const uint8_t *context, uint32_t context_len,
                                         uint64_t n, uint8_t *d_pks, uint8_t *d_sigs);
int32_t ed25519_sign_batch_prehashed(c25519_ctx *ctx, const uint8_t *seeds, const uint8_t *prehashes, const uint8_t *context, uint32_t context_len, uint64_t n,
                                     uint8_t *pks, uint8_t *sigs);

/* ---- precomputed static points: VartimePrecomputedMultiscalarMul (traits.rs:304-419) ---------------------
 * replaces backend::VartimePrecomputedStraus::{new, len, optional_mixed_multiscalar_mul}
 * (backend.rs:100-192 -> scalar_mul/precomputed_straus.rs:29-127; edwards.rs:1037-1076).
 * create: for static points S_i (HOST pointer; fmt 0/1/2) the multiples 2^(c k) S_i of every window k are computed
 * once and stay resident in HBM as affine Niels records (17 x 128 bytes per point at c = 16): the bucket-method
 * counterpart of the reference's per-point NAF tables.  A call then needs no point preparation, no per-window passes and
 * no doubling chain -- one bucket accumulation and one bucket reduction over all digits of all scalars.  Returns NULL
 * if a point does not decode.  msm: sum static_scalars[i]*S_i (the first n_static_scalars static points;
 * more scalars than points is an error, precomputed_straus.rs:86) + sum dyn_scalars[j]*dyn_points[j] (the dynamic terms
 * run through the ordinary MSM and are added).
 * All pointers of the msm call are HOST pointers; returns C25519_NONE iff a dynamic point does not decode. */
typedef struct c25519_precomp c25519_precomp;
c25519_precomp *c25519_precomp_create(c25519_ctx *ctx, const uint8_t *static_points, uint64_t n, int in_fmt);
void c25519_precomp_destroy(c25519_ctx *ctx, c25519_precomp *p);
uint64_t c25519_precomp_len(const c25519_precomp *p);
int32_t c25519_precomp_msm_vartime(c25519_ctx *ctx, const c25519_precomp *p, const uint8_t *static_scalars, uint64_t n_static_scalars,
                                   const uint8_t *dyn_scalars, const uint8_t *dyn_points, uint64_t n_dyn, int in_fmt, int out_fmt, uint8_t *out);

/* ---- regular-schedule multiscalar multiplication: MultiscalarMul::multiscalar_mul ------------------------
 * replaces backend::straus_multiscalar_mul (backend.rs:196 -> scalar_mul/straus.rs:103-144;
 * edwards.rs:966-1000).  One radix-16 fixed-window ladder per term (the schedule of variable_base.rs: identical
 * instruction stream for every input, every table lookup a full scan with selects, whatever the context's flags) and a
 * tree sum; staged scalars and per-lane tables are wiped.  HOST pointers; fmt as c25519_mul_batch. */
int32_t c25519_msm_consttime(c25519_ctx *ctx, const uint8_t *scalars, const uint8_t *points, uint64_t n, int in_fmt, int out_fmt, uint8_t *out);

/* ---- RistrettoPoint::double_and_compress_batch (ristretto.rs:564-648): out[i] = compress(2 * P_i) with one
 * shared inversion per lane-chunk.  in: n x 160 raw points; out: n x 32 CompressedRistretto. */
int32_t c25519_double_and_compress_batch_dev(c25519_ctx *ctx, const uint8_t *d_in, uint64_t n, uint8_t *d_out);
int32_t c25519_double_and_compress_batch(c25519_ctx *ctx, const uint8_t *in, uint64_t n, uint8_t *out);

/* ---- Scalar::invert_batch_alloc (scalar.rs:802-856): io[i] <- 1/io[i] mod l in place (HOST pointer; all inputs
 * must be canonical and non-zero, as in the reference); prod_inv (32 bytes, may be NULL) receives the product of
 * all inverses, the reference's return value. */
int32_t c25519_scalar_invert_batch(c25519_ctx *ctx, uint8_t *io, uint64_t n, uint8_t *prod_inv);

/* ---- diagnostics ----------------------------------------------------------------------------------
 * Integer-multiplier roofline probes: runs a dependent-free chain microbenchmark and
 * returns giga-operations per second.  which: 0 v_mad_u64_u32, 1 fe_mul (radix 2^25.5, this
 * engine), 2 fe_sq, 3 fe_mul written on 5 x u64 limbs with unsigned __int128 products (the
 * reference's literal layout, for the A/B in DESIGN.md), 4 v_add_u32, 5 v_mul_lo_u32. */
double c25519_microbench(c25519_ctx *ctx, int which, int iters);
/* Device field self-test: runs ONE field operation per element on the GPU, in the translation unit built with the
 * chained-carry multiplication (chain = 1: kernels.hip) or the ten-column one (chain = 0), and returns canonical bytes
 * (field.rs:368-450 to_bytes).  a_limbs / b_limbs: n x 10 u32 limbs at bit positions 0,26,51,...,230 (HOST pointers;
 * any magnitudes the operation's bound class admits, csrc/fe26.h); out: n x 32.  op: 0 a*b (a wide, b loose), 1 a^2
 * (loose), 2 1/a, 3 canonical encoding of a (wide), 4 a^((p-5)/8), 5 a-b (both loose), 6 weak reduction of a,
 * 7 (a-b)*(a+b) (both tight); 8-11 the lockstep multiplier of the bucket accumulation (csrc/fe26x.h; a wide, b loose):
 * 8 a*b and 9 b^2 out of one group of three products, 10 a*b and 11 b^2 out of one group of four.  Pins the device code
 * generation against big integers (field.rs:552-642). */
int32_t c25519_selftest_field(c25519_ctx *ctx, int op, int chain, const uint32_t *a_limbs, const uint32_t *b_limbs, uint64_t n, uint8_t *out);
/* The same for the device SCALAR arithmetic mod l (csrc/sc28.h: ten 28-bit limbs, reduction by folding with l = 2^252 + c),
 * which replaces Scalar52 (u64/scalar.rs:66-320) inside the verify_batch / sign / per-signature-verify kernels.
 * a_words / b_words: n x 16 u32 per operand (HOST pointers; b may be NULL for the unary ops); out: n x 32 bytes.
 * op 0 from_bytes_mod_order_wide(a: 16 words) (scalar.rs:248, u64/scalar.rs:89-118); 1 a*b mod l, a = 5 and b = 10 raw 28-bit
 * limbs with a*b < 2^393 (z_i * s_i, z_i * h_i: batch.rs:225-233); 2 a*b mod l, 10 x 10 raw limbs with a*b < 2^512 (u64/scalar.rs:302;
 * signing.rs:899 k*a with the clamped a); 3 a+b and 4 -a on canonical operands (8 words each; u64/scalar.rs:161-207);
 * 5 out[0] = (a < l), the word-wise test behind from_canonical_bytes (scalar.rs:259-263); 6 words -> limbs -> words of a < 2^256;
 * 7 r + k*a as the signer chains them: k = a mod l (16 words), a = b[0..8] (unreduced, < 2^256), r = b[8..16] (canonical). */
int32_t c25519_selftest_scalar(c25519_ctx *ctx, int op, const uint32_t *a_words, const uint32_t *b_words, uint64_t n, uint8_t *out);
/* The window layout the MSM uses for n terms of RAW points (host arithmetic, no GPU needed; encoded inputs of 4096 .. 6143 terms and verify_batch choose widths of
 * their own -- every record carries the width it was made with): window k covers bits
 * [pos[k], pos[k] + wid[k]) of s' = s + addk (addk as 8 little-endian 32-bit words); all windows but the last two are
 * signed (digit = slice - 2^(wid-1)).  pos / wid need room for 56 entries.  Used by the CPU tests to check that the
 * digits always recompose the scalar. */
int32_t c25519_msm_geometry(uint64_t n, int32_t *c, int32_t *nwin, uint8_t *pos, uint8_t *wid, uint32_t *addk);

#ifdef __cplusplus
}
#endif
#endif
