#!/usr/bin/env python3
"""Per-loop instruction counts of one kernel in a hipcc -S listing: for every backward branch, the number of
v_mad_u64_u32, moves and 64-bit adds between the loop label and the branch.  Used to check that a field
multiplication inside a loop still costs 100 (fe_mul) / 55 (fe_sq) multiplier instructions.
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o k.s csrc/kernels.hip
    python tools/isa_loops.py k.s k_mul_base_wide"""
import re
import sys


def main(path, pat):
    txt = open(path).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
        if pat not in m.group(1):
            continue
        body = [l for l in m.group(2).split("\n") if not l.strip().startswith(";")]
        labels = {}
        for i, l in enumerate(body):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm:
                labels[mm.group(1)] = i
        print(m.group(1)[:90])
        for i, l in enumerate(body):
            mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                seg = body[labels[mm.group(1)]:i]
                c = lambda s: sum(s in x for x in seg)
                print("  loop %-10s %5d instr: mad_u64 %4d  v_mov %4d  lshl_add_u64 %3d  mul_lo %3d  global_load %2d  scratch %2d" % (
                    mm.group(1), len(seg), c("v_mad_u64_u32"), c("v_mov_b32"), c("v_lshl_add_u64"), c("v_mul_lo_u32"), c("global_load"), c("scratch_")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
