#!/usr/bin/env python
"""Static VALU instruction / issue-cycle estimate of the per-chain loop of a
fused-kernel instantiation (reads the -save-temps .s).  Cost classes from
tools/instr_bench.hip on MI355X: simple VALU 2 cycles, v_pk_* / integer
multiply / cvt / alignbit / f64 4, transcendental 8 (wave64, SIMD-32).
Usage: python tools/asm_count.py file.s 'Li64ELi4ELb1ELb0' [L]"""
import collections
import re
import sys

path, inst = sys.argv[1], sys.argv[2]
L = int(sys.argv[3]) if len(sys.argv) > 3 else 10
src = open(path).read().split('\n')
beg = next(i for i, l in enumerate(src)
           if l.startswith('_ZN5zshmc22hmc_diag_normal_kernel') and inst in l
           and (inst + 'EEEvNS_9FusedArgsE:') in l)
end = next(i for i in range(beg, len(src)) if 's_endpgm' in src[i])
body = src[beg:end]
hdr = [i for i, l in enumerate(body) if 'Loop Header: Depth=1' in l and
       i + 1 < len(body) and 'Child Loop' in body[i + 1]]
h = hdr[0]
label = body[h].split(':')[0].strip()
name = label.replace('.L', '')
last = max(i for i, l in enumerate(body) if 'Header=' + name in l)
close = last
for i in range(last, min(last + 300, len(body))):
    if re.search(r's_cbranch\w+\s+' + re.escape(label) + r'\b', body[i]) or \
            re.search(r's_branch\s+' + re.escape(label) + r'\b', body[i]):
        close = i
        break
inner_s = next(i for i in range(h, close) if 'Inner Loop Header: Depth=2' in body[i])
ilabel = None
for i in range(inner_s, inner_s - 5, -1):
    if body[i].startswith('.LBB'):
        ilabel = body[i].split(':')[0]
        break
inner_e = next(i for i in range(inner_s, close)
               if re.search(r's_cbranch\w+\s+' + re.escape(ilabel) + r'\b', body[i]))


def cost(op):
    if re.match(r'v_(log|sin|cos|sqrt|exp|rcp|rsq)_f32', op):
        return 8
    if re.match(r'v_(mad_u64|mul_lo|mul_hi|cvt_|alignbit|pk_|lshl_add_u64|'
                r'mul_u32_u24|mad_u32)', op) or 'f64' in op:
        return 4
    return 2 if op.startswith('v_') else 0


tot, cyc, other = (collections.Counter() for _ in range(3))
n = c = 0
for i in range(h, close + 1):
    l = body[i].split(';')[0].strip()
    if not l or l.endswith(':') or l.startswith('.'):
        continue
    op = l.split()[0]
    w = L if inner_s <= i <= inner_e else 1
    if op.startswith('v_'):
        tot[op] += w
        cyc[op] += w * cost(op)
        n += w
        c += w * cost(op)
    else:
        other[op] += w
print('chain loop lines %d..%d, leapfrog loop %d..%d' % (h, close, inner_s, inner_e))
print('VALU instructions per chain (L=%d): %d   estimated issue cycles: %d' % (L, n, c))
for op, v in cyc.most_common(18):
    print('  %-24s n=%4d cyc=%5d' % (op, tot[op], v))
print('  non-VALU:', other.most_common(10))
