#!/usr/bin/env python3
"""Per-kernel instruction statistics of a gfx950 assembly file (hipcc -S / -save-temps): registers, scratch, and opcode counts of the whole body and of
every loop.   python tools/isa_stats.py file.s <mangled-name-regex> [loops]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
show_loops = len(sys.argv) > 3
Q = ('v_mad_u64_u32', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_lshrrev_b64', 'v_lshlrev_b64', 'v_cndmask_b32_e64', 'v_cndmask_b32_e32', 'v_alignbit_b32', 'v_lshlrev_b32_e32', 'v_lshl_add_u32', 'v_lshl_or_b32', 'v_and_or_b32', 'v_add3_u32', 'v_bfe_u32', 'v_perm_b32', 'v_lshl_add_u64')
def summ(ops):
    quarter = sum(ops[k] for k in Q)
    valu = sum(v for k, v in ops.items() if k.startswith('v_'))
    return 'valu %d (mad %d, other 4-cycle %d, cndmask %d) ds_read %d bpermute %d vmem %d salu %d' % (
        valu, ops['v_mad_u64_u32'], quarter - ops['v_mad_u64_u32'], ops['v_cndmask_b32_e64'] + ops['v_cndmask_b32_e32'],
        sum(v for k, v in ops.items() if k.startswith('ds_read')), ops['ds_bpermute_b32'],
        sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_'))), sum(v for k, v in ops.items() if k.startswith('s_')))
for i, l in enumerate(lines):
    m = re.match(r'^(' + pat + r'\S*):', l)
    if not m:
        continue
    end = next(j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end'))
    body = lines[i:end]
    meta = {}
    for x in lines[end:end + 120]:
        mm = re.match(r'\s*;\s*(NumVgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs): (\d+)', x)
        if mm:
            meta.setdefault(mm.group(1), mm.group(2))
    ops = Counter(y.split()[0] for y in body if re.match(r'^\s+[a-z]', y))
    print(m.group(1)[:90])
    print('   ', meta, summ(ops))
    if show_loops:
        labels = {}
        for a, x in enumerate(body):
            mm = re.match(r'^(\.LBB\d+_\d+):', x)
            if mm:
                labels[mm.group(1)] = a
        for a, x in enumerate(body):
            mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', x)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < a:
                lo = Counter(y.split()[0] for y in body[labels[mm.group(1)]:a + 1] if re.match(r'^\s+[a-z]', y))
                if sum(lo.values()) > 40:
                    print('      loop %s: %s' % (mm.group(1), summ(lo)))
