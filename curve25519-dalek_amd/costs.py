"""Field-operation counts of the hot path AS IMPLEMENTED, derived from the kernels' parameters -- the ONE table that
bench.py's `valu` objects and DESIGN.md section 4 both read (round 1 kept hand-typed constants in bench.py that drifted
from the code).

M = fe_mul (csrc/fe26.h: 100 v_mad_u64_u32 products + 1 for the 19*carry fold), S = fe_sq (55 + 1).  Every function
returns a dict {"M": .., "S": .., "what": ..} per UNIT of the workload (scalar, ladder, term, signature), and
`mac(cost)` turns it into 32x32->64 multiply-accumulates, the unit of the binding roof.
`reference_mac` holds SURVEY.md 8d's counts for the REFERENCE algorithm (5 x 51-bit schoolbook: M = 100, S = 60).

Formula sources (file:function):
  mixed addition                7 M      ge26.h:ge_madd_signed_p3
  first window by conversion    1 M      ge26.h:ge_from_aniels_signed
  complete addition             9 M      ge26.h:ge_add  (1 M to_cached + 4 M + 4 M)
  doubling to extended          4 S+4 M  ge26.h:ge_dbl + ge_p1p1_to_p3   (3 M when another doubling follows)
  inversion / pow_p58           254 S + 11 M / 251 S + 12 M              fe26.h:fe_invert, fe_pow_p58
  decompression                 255 S + 21 M                             ge26.h:ge_decompress (T = X*Y is dead code in the MSM prep)
  batched compression           5 M + inversion / 16                     finish.hip:k_compress_p32<16>
  affine Niels record           2 M (x*y, *2d)                           devio.h:pts_store
"""
import math

MAC_PER_M = 100
MAC_PER_S = 55
INV = {"M": 11, "S": 254}          # fe_invert
CHUNK = 16                         # points per lane sharing one inversion (Montgomery's trick): compression, ladder output
PREP_CHUNK = 64                    # the same in the MSM's normaliser (msm.hip:k_prep_raw2<64, 4>)

reference_mac = {"fixed_base": 47100, "x25519": 231000, "msm": 26500, "verify": 57000, "verify_bytes": 74400}


def mac(c):
    return MAC_PER_M * c["M"] + MAC_PER_S * c["S"]


def _add(*cs):
    return {"M": sum(c["M"] for c in cs), "S": sum(c["S"] for c in cs)}


def _scaled(c, f):
    return {"M": c["M"] * f, "S": c["S"] * f}


def compress_batch():
    """finish.hip:k_compress_p32<16>: prefix product 1 M, zi = inv*pre 1 M, inv*Z 1 M, X*zi, Y*zi 2 M, + 1/16 inversion."""
    return _add({"M": 5, "S": 0}, _scaled(INV, 1.0 / CHUNK))


def fixed_base_wide(C=16):
    """kernels.hip:k_mul_base_wide: ceil(256/C) windows, the first by conversion, no doublings; + batched compression."""
    nw = -(-256 // C)
    c = _add({"M": 1 + 7 * (nw - 1), "S": 0}, compress_batch())
    c["what"] = "1 M (first window) + %d madd x 7 M (radix-2^%d tables in HBM) + 5 M compress + 1/16 inversion" % (nw - 1, C)
    return c


def fixed_base_comb():
    """kernels.hip:k_mul_base_comb: 31 mixed additions (6 tables x 5 rows + the parity fix-up) + 4 doublings."""
    c = _add({"M": 31 * 7 + 4 * 4, "S": 4 * 4}, compress_batch())
    c["what"] = "31 madd x 7 M + 4 dbl x (4 S + 4 M) (LDS comb) + 5 M compress + 1/16 inversion"
    return c


def fixed_base_ct(W=5):
    """kernels.hip:k_mul_base_ctp<W> (round 5; rounds 2-4: k_mul_base<W, CT>): ceil(256/W) mixed additions; the table entry reaches the lane by a
    cross-lane fetch (LDS reads at lane-index addresses + ds_bpermute), which is not multiplier work and therefore not in this count."""
    nw = -(-256 // W)
    c = _add({"M": 7 * nw, "S": 0}, compress_batch())
    c["what"] = "%d madd x 7 M (radix-2^%d LDS tables, constant-time cross-lane fetch per lookup) + 5 M compress + 1/16 inversion" % (nw, W)
    return c


def x25519():
    """kernels.hip:k_x25519: 255 ladder steps of 5 M + 4 S + one 10-product small multiplication (0.1 M), then
    finish.hip:k_ratio_p32 (3 M + 1/16 inversion)."""
    c = _add({"M": 255 * 5.1, "S": 255 * 4}, {"M": 3, "S": 0}, _scaled(INV, 1.0 / CHUNK))
    c["what"] = "255 x (5 M + 4 S + 10-product a24 mul) + 3 M + 1/16 inversion"
    return c


def _reduce_per_bucket():
    """Bucket reduction sum_b (b+1) B_b, priced at its ALGORITHMIC minimum -- the running-sum method of
    pippenger.rs:146-151, two complete additions (9 M each) per bucket.  msm.hip:k_reduce_a / k_reduce_b issue more
    (per 8 buckets: 13 serial additions, then 13 additions + 3 doublings of the wave-wide combine, i.e. 3.25 additions
    + 0.375 doublings per bucket) to cut the dependent chain from 2 x 32768 additions to 40; the surplus is overhead,
    not achieved work."""
    return {"M": 2 * 9, "S": 0}


def msm(n, nwin, half, raw_points=True, filled_windows=None):
    """One bucket-method pass over n terms with `nwin` windows of `half` buckets (c25519_msm_geometry).
    filled_windows: windows that hold a digit for a reduced (< 2^253) scalar: all but the overflow window."""
    fw = (nwin - 1) if filled_windows is None else filled_windows
    acc = {"M": 7 * fw, "S": 0}
    if raw_points:       # msm.hip:k_prep_raw2<64>: 1 M prefix, 2 M unwind, 2 M (x, y), 2 M record, + 1/64 inversion
        prep = _add({"M": 7, "S": 0}, _scaled(INV, 1.0 / PREP_CHUNK))
    else:                # kernels.hip:k_prep_compressed: decompression + record
        prep = {"M": 21 + 2, "S": 255}
    red = _scaled(_reduce_per_bucket(), float(nwin) * half / n)
    c = _add(acc, prep, red)
    c["what"] = "%d windows x 7 M bucket adds + %s + bucket reduction %.1f M-equivalents per term (%d x %d buckets)" % (
        fw, "7.2 M + 4 S normalise (1/64 inversion)" if raw_points else "decompress (255 S + 23 M)", (mac(red)) / 100.0, nwin, half)
    return c


def verify(n, nwin, half, keys_as_bytes=False, r_windows=8):
    """One verify_batch pass: R decompressed (255 S + 23 M incl. record), A from the cached point (Z = 1, as
    VerifyingKey::from_bytes leaves it: 1 M prefix product + 2 M record, msm.hip:k_prep_raw's affine path) or decompressed, R-terms in `r_windows` windows (|z| < 2^127), A-terms in nwin - 1; SHA-512 and the scalar
    arithmetic mod l are integer work outside this count."""
    a_prep = {"M": 23, "S": 255} if keys_as_bytes else {"M": 3, "S": 0}
    red = _scaled(_reduce_per_bucket(), float(nwin) * half / n)
    c = _add({"M": 23, "S": 255}, a_prep, {"M": 7 * (r_windows + nwin - 1), "S": 0}, red)
    c["what"] = "decompress R (255 S + 23 M) + %s + (%d + %d) windows x 7 M + bucket reduction; SHA-512 and scalar muls not counted" % (
        "decompress A (255 S + 23 M)" if keys_as_bytes else "A from VerifyingKey.point (Z = 1: 3 M)", r_windows, nwin - 1)
    return c


def table():
    """Rows for DESIGN.md (python -m curve25519_dalek_amd.costs)."""
    rows = [("fixed base, radix 2^16 tables", fixed_base_wide(16)), ("fixed base, LDS comb", fixed_base_comb()),
            ("fixed base, constant-time scan W=5", fixed_base_ct(5)), ("X25519 ladder", x25519()),
            ("MSM 2^21 terms, raw points", msm(1 << 21, 17, 32768)), ("verify_batch 2^20, VerifyingKey", verify(1 << 20, 17, 32768)),
            ("verify_batch 2^20, keys as bytes", verify(1 << 20, 17, 32768, True))]
    return [(name, round(c["M"], 1), round(c["S"], 1), int(round(mac(c))), c["what"]) for name, c in rows]


if __name__ == "__main__":
    for r in table():
        print("| %s | %s M + %s S | %d | %s |" % r)
