"""Constant-time posture, checked on the COMPILED code (no GPU needed: hipcc -S for gfx950).

The secret-scalar kernels promise what the reference's LookupTable::select promises (window.rs:54-76): every table
entry of a window is read and the wanted one kept by a select -- no address and no branch depends on the scalar.  The
source says so, but the compiler is free to turn `hit ? table[j] : acc` into a read that only the lanes with `hit` perform,
behind a branch on "does any lane of the wave have this digit" (it did: round 2 found s_cbranch_execz around the LDS reads
of k_mul_base<5, CT>).  So the property is asserted on the instruction stream:
  * the window loop of k_mul_base_ctp<5, *, *, *> (round 5, the default: cross-lane fetch): six LDS reads at lane-index addresses, 24 / 30
    ds_bpermute_b32, the addition -- no other LDS or memory read, no exec-mask branch;
  * the scan loop of k_mul_base<5, 1024, *, CT> (rounds 2-4, kept as the A/B arm): 6 ds_read_b128 + 24 v_cndmask per entry, no exec-mask branch;
  * the scan loop of k_var_base<*, *, CT>: only unconditional loads and selects, no exec-mask branch;
  * the ladder loop of k_x25519: no exec-mask branch, and conditional swaps as v_cndmask."""
import os
import re
import shutil
import subprocess
from collections import Counter

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curve25519-dalek_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


def _asm(tmp_path_factory, name):
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, name + ".hip")],
                   check=True, capture_output=True, timeout=900)
    return open(out).read().split("\n")


def _functions(lines, pattern):
    """-> {mangled name: body lines} for every function whose label matches"""
    out = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(" + pattern + r"\S*):", l)
        if m:
            end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
            out[m.group(1)] = lines[i:end]
    return out


def _loops(body):
    """-> [(label, Counter of opcodes)] for every backward branch (innermost loops come first in the text)"""
    labels = {}
    for a, x in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", x)
        if m:
            labels[m.group(1)] = a
    res = []
    for a, x in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", x)
        if m and m.group(1) in labels and labels[m.group(1)] < a:
            ops = Counter(y.split()[0] for y in body[labels[m.group(1)]:a + 1] if re.match(r"^\s+[a-z]", y))
            res.append((m.group(1), ops))
    return res


def _exec_branches(ops):
    return sum(v for k, v in ops.items() if k.startswith("s_cbranch_exec") or k.startswith("s_cbranch_vcc") or "saveexec" in k)


@pytest.fixture(scope="module")
def kernels_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "kernels")


@pytest.fixture(scope="module")
def single_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "single")


def test_fixed_base_cross_lane_fetch_loop(kernels_asm):
    """k_mul_base_ctp (kernels.hip): per window a lane reads the table entry of ITS LANE INDEX (6 ds_read_b128) and pulls the entry of its digit
    from another lane's registers (24 ds_bpermute_b32: the digit is the lane selector of a register-to-register transfer).  Asserted on every
    instantiation (block sizes 256 / 512 / 1024, three output formats, split and unsplit): exactly these LDS operations in the window loop, no
    other LDS / global / scratch read there (nothing a digit could address), no branch on exec or vcc (nothing a digit could steer)."""
    fns = _functions(kernels_asm, r"_ZN6c2551914k_mul_base_ctpILi5ELi\d+ELi\dELb[01]ELb[01]E")
    assert len(fns) >= 16, sorted(fns)
    for name, body in fns.items():
        loops = [ops for _, ops in _loops(body) if ops.get("ds_bpermute_b32", 0) and ops.get("v_mad_u64_u32", 0) > 500]
        assert loops, name
        inner = min(loops, key=lambda o: sum(o.values()))                      # the window loop (the scalar loop around it contains it)
        limbs = name.endswith("ELb1EEEvPKhmPK15HIP_vector_typeIjLj4EEPjPh")       # the limb-table form: 9 reads (3 x 3 pieces) and 30 permutes; packed tables: 6 and 24
        assert inner["ds_bpermute_b32"] == (30 if limbs else 24), (name, dict(inner))
        # (the third 16-byte piece of a limb element holds two limbs: the compiler reads it as ds_read_b64)
        assert inner["ds_read_b128"] == 6 and inner.get("ds_read_b64", 0) == (3 if limbs else 0) and sum(v for k, v in inner.items() if k.startswith("ds_read")) == (9 if limbs else 6), (name, dict(inner))
        # (a scratch_load is a register spill at a compile-time offset of the lane's private stack -- the raw / P40 output forms of the 1024-thread
        #  kernel spill six words at the 128-register budget -- not a table access; global / buffer / flat loads would be)
        assert sum(v for k, v in inner.items() if k.startswith(("global_load", "buffer_load", "flat_load"))) == 0, (name, dict(inner))
        assert sum(v for k, v in inner.items() if k.startswith("scratch_load")) <= 2, (name, dict(inner))
        assert sum(v for k, v in inner.items() if k.startswith("ds_") and not k.startswith(("ds_read_b128", "ds_read_b64", "ds_bpermute_b32"))) == 0, (name, dict(inner))
        assert _exec_branches(inner) == 0, (name, dict(inner))
        # all 64 lanes take part in every permute: no exec manipulation anywhere inside the window loop
        assert not any("exec" in k for k in inner), (name, [k for k in inner if "exec" in k])


def test_fixed_base_scan_reads_every_entry_without_a_branch(kernels_asm):
    fns = _functions(kernels_asm, r"_ZN6c2551910k_mul_baseILi5ELi1024ELi\dELb1E")
    assert len(fns) >= 2                                     # the output-format instantiations of the constant-time kernel
    for name, body in fns.items():
        scans = [ops for _, ops in _loops(body) if ops.get("ds_read_b128", 0) and not ops.get("v_mad_u64_u32", 0)]
        assert len(scans) == 1, (name, [dict(o) for _, o in _loops(body)])
        ops = scans[0]
        assert ops["ds_read_b128"] == 6, (name, dict(ops))                      # one 96-byte entry per trip, always
        assert ops.get("v_cndmask_b32_e64", 0) + ops.get("v_cndmask_b32_e32", 0) >= 24, (name, dict(ops))
        assert _exec_branches(ops) == 0, (name, dict(ops))                      # the only branch is the scalar trip counter


def test_fixed_base_split_kernel_scan_is_the_same(kernels_asm):
    """k_mul_base_ct_split (small batches: a scalar's windows split between two threads): the same scan -- 6 LDS reads and 24
    selects per entry, no exec-mask branch -- in every block-size / output-format instantiation; the part a thread works on is
    uniform per wave, so the table addresses stay wave-uniform"""
    fns = _functions(kernels_asm, r"_ZN6c2551919k_mul_base_ct_splitILi\d+ELi\dE")
    assert len(fns) >= 6, sorted(fns)
    for name, body in fns.items():
        scans = [ops for _, ops in _loops(body) if ops.get("ds_read_b128", 0) == 6 and not ops.get("v_mad_u64_u32", 0)]
        assert len(scans) == 1, (name, [dict(o) for _, o in _loops(body)])
        ops = scans[0]
        assert ops.get("v_cndmask_b32_e64", 0) + ops.get("v_cndmask_b32_e32", 0) >= 24, (name, dict(ops))
        assert _exec_branches(ops) == 0, (name, dict(ops))


def test_variable_base_scan_has_no_data_dependent_branch(single_asm):
    fns = _functions(single_asm, r"_ZN6c2551910k_var_baseILi\dELb\dELb1E")
    assert len(fns) >= 2
    for name, body in fns.items():
        scans = [ops for _, ops in _loops(body) if sum(v for k, v in ops.items() if k.startswith("global_load")) and not ops.get("v_mad_u64_u32", 0)]
        assert scans, name
        for ops in scans:
            assert _exec_branches(ops) == 0, (name, dict(ops))
            assert ops.get("v_cndmask_b32_e64", 0) + ops.get("v_cndmask_b32_e32", 0) >= 40, (name, dict(ops))   # 4 x 10 limbs per entry


def test_ladder_loop_has_no_data_dependent_branch(kernels_asm):
    fns = _functions(kernels_asm, r"_ZN6c255198k_x25519E")
    assert len(fns) == 1
    body = next(iter(fns.values()))
    ladders = [ops for _, ops in _loops(body) if ops.get("v_mad_u64_u32", 0) > 500]
    assert len(ladders) == 1
    ops = ladders[0]
    assert _exec_branches(ops) == 0, dict(ops)
    assert ops.get("v_cndmask_b32_e64", 0) + ops.get("v_cndmask_b32_e32", 0) >= 40, dict(ops)                   # cswap of (U, W) pairs


# ---- (r6, last) what DESIGN.md section 3.7 found in the compiled loops must not come back ------------------------------------------------------------------
@pytest.fixture(scope="module")
def accum_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "accum")


def test_chained_products_carry_no_s_nop(kernels_asm):
    """The chained-carry products pin every partial sum with an empty asm statement (fe26.h C25519_PIN).  As a statement that DEFINES the register it made the
    compiler's hazard recogniser put an s_nop behind 45 - 75 % of the v_mad_u64_u32 of every chained-form kernel (334 per ladder step of k_x25519: a quarter of a lone
    wave's time, profiles/r06_ab_pin_input.txt); as an INPUT of a volatile statement it does not.  Asserted on the ladder loop and on the window loop of the
    constant-time fixed base: a few s_nop per several hundred products, and still exactly the products of the arithmetic (739 = 5 M + 4 S + one small product)."""
    lad = _functions(kernels_asm, r"_ZN6c255198k_x25519E")
    assert len(lad) == 1
    loops = [ops for _, ops in _loops(next(iter(lad.values()))) if ops["v_mad_u64_u32"] > 500]
    assert loops, "ladder loop not found"
    for ops in loops:
        assert ops["v_mad_u64_u32"] == 739, ops["v_mad_u64_u32"]
        assert ops["s_nop"] <= 20, ops["s_nop"]
    fns = _functions(kernels_asm, r"_ZN6c2551914k_mul_base_ctpILi5ELi1024ELi0ELb1ELb1E")
    assert fns
    for body in fns.values():
        win = [ops for _, ops in _loops(body) if ops["v_mad_u64_u32"] > 500]
        assert win
        for ops in win:
            assert ops["s_nop"] <= 20, ops["s_nop"]


def test_accumulate_gathers_use_scalar_bases(accum_asm):
    """k_accumulate's eight gathers per addition: the record base and the LDS destination are SCALAR (global_load_lds_dwordx4 v, s[..] with M0 from scalar registers), the
    per-lane part is one 32-bit offset -- no v_readfirstlane and no 64-bit address arithmetic in the loop (they were 65 of 1 253 vector instructions per addition:
    profiles/r06_ab_accumulate_gather.txt), no scratch, and the seven products of the mixed addition."""
    fns = _functions(accum_asm, r"_ZN6c2551912k_accumulateE")
    assert len(fns) == 1
    body = next(iter(fns.values()))
    assert not any(re.match(r"^\s+scratch_", l) for l in body)
    main = [ops for _, ops in _loops(body) if ops["v_mad_u64_u32"] > 500]
    assert len(main) >= 1
    for ops in main:
        assert ops["v_mad_u64_u32"] == 708, ops["v_mad_u64_u32"]
        assert ops["v_readfirstlane_b32"] == 0 and ops["v_lshlrev_b64"] == 0, (ops["v_readfirstlane_b32"], ops["v_lshlrev_b64"])
        assert ops["global_load_lds_dwordx4"] in (8, 16)      # (the text between the loop label and the backward branch also holds the issue of the first gather)
        assert ops["v_cndmask_b32_e64"] + ops["v_cndmask_b32_e32"] == 0      # (the digit's sign is applied lazily to the accumulator: v_xad_u32, no select -- profiles/r06_ab_lazy_sign.txt)
        assert ops["v_xad_u32"] >= 20
        assert sum(v for k, v in ops.items() if k.startswith("v_")) <= 1260      # (loop + the issue of the first gather + the sign fix-up behind the loop, which the text range includes)
    loads = [l for l in body if "global_load_lds_dwordx4" in l]
    assert loads and all(re.search(r"global_load_lds_dwordx4 v\d+, s\[\d+:\d+\]", l) for l in loads), loads[:3]
