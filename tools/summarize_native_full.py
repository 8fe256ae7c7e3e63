#!/usr/bin/env python
"""Summary of tools/profile_native_full.sh: durations and PMC counters per
launch of the MFMA likelihood kernel at the full configs[2] / configs[4]
shapes; HBM traffic with the guide's gfx950 correction (FETCH_SIZE counts
64-byte units at half rate on gfx950: x2; both counters are in KiB)."""
import csv
import glob
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
lines = []


KERNELS = ('linear_bernoulli_kernel', 'linear_b3_kernel')


def is_lik(name):
    return any(k in name for k in KERNELS)


def mode_of(name):
    """'<family> <call form>' from the kernel's template arguments
    <D, GRAD, OP, LL> (demangled or mangled); the bf16x3 kernel
    (linear_b3_kernel<D, OP, LL, NACC>) is tagged."""
    if 'linear_b3_kernel' in name:
        tail = name.split('linear_b3_kernel')[1][:80]
        m = re.search(r'<\s*(\d+),\s*(\d),\s*(true|false)', tail)
        if m:
            op, ll = int(m.group(2)), m.group(3) == 'true'
        else:
            m = re.search(r'ILi(\d+)ELi(\d)ELb([01])E', tail)
            op, ll = int(m.group(2)), m.group(3) == '1'
        fam = {0: 'bernoulli', 1: 'multinomial'}[op]
        # round 6: the packed-rows instantiations (6th template argument)
        args = re.search(r'<([^>]*)>', tail)
        packed = (args is not None and
                  args.group(1).replace(' ', '').split(',')[5:6] == ['true']) \
            or re.search(r'ELi\dELi\dELb1EEE', tail) is not None
        own = (args is not None and
               args.group(1).replace(' ', '').split(',')[6:7] == ['true']) \
            or re.search(r'ELb[01]ELb1EEE', tail) is not None
        if own:
            packed = False
        return '%s bf16x3%s %s' % (fam, ' packed-rows' if packed else
                                   (' own-vocabulary' if own else ''),
                                   'll+grad' if ll else 'grad-only')
    tail = name.split('linear_bernoulli_kernel')[1][:80]
    m = re.search(r'<\s*(\d+),\s*(true|false),\s*(\d)(?:,\s*(true|false))?\s*>',
                  tail)
    if m:
        op, ll = int(m.group(3)), (m.group(4) or 'true') == 'true'
    else:
        m = re.search(r'ILi(\d+)ELb([01])ELi(\d)E(?:Lb([01])E)?', tail)
        op, ll = int(m.group(3)), (m.group(4) or '1') == '1'
    fam = {0: 'bernoulli', 1: 'multinomial', 2: 'categorical'}[op]
    return '%s %s' % (fam, 'll+grad' if ll else 'grad-only')


for f in glob.glob(os.path.join(out, tag + '_nativefull_trace', '**',
                                '*kernel_trace.csv'), recursive=True):
    dur = {}
    for r in csv.DictReader(open(f)):
        if is_lik(r['Kernel_Name']):
            dur.setdefault(mode_of(r['Kernel_Name']), []).append(
                (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
    for m, v in sorted(dur.items()):
        lines.append('kernel trace  %-30s launches %d  ms each: %s' % (
            m, len(v), ' '.join('%.2f' % x for x in v)))
vals = {}
for kind in ('fetch', 'write', 'mfma'):
    for f in glob.glob(os.path.join(out, '%s_nativefull_%s' % (tag, kind),
                                    '**', '*counter_collection.csv'),
                       recursive=True):
        agg = {}
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '')
            if not is_lik(k):
                continue
            key = (mode_of(k), row['Counter_Name'])
            agg.setdefault(key, {}).setdefault(row['Dispatch_Id'], 0.0)
            agg[key][row['Dispatch_Id']] += float(row['Counter_Value'])
        for (m, c), d in sorted(agg.items()):
            v = list(d.values())
            vals[(m, c)] = sum(v) / len(v)
            lines.append('pmc  %-30s %-28s mean per launch %.6g (n=%d)' % (
                m, c, vals[(m, c)], len(v)))
for m in sorted({k[0] for k in vals}):
    if (m, 'FETCH_SIZE') in vals and (m, 'WRITE_SIZE') in vals:
        rd = vals[(m, 'FETCH_SIZE')] * 1024 * 2     # gfx950: half-count
        wr = vals[(m, 'WRITE_SIZE')] * 1024
        lines.append('HBM traffic  %-30s read %.4g B (FETCH_SIZE x 2 KiB) + '
                     'write %.4g B = %.4g B per launch' % (m, rd, wr, rd + wr))
# derived: TFLOP/s of a launch (4 N D C flop, shapes of tools/native_kernel_pmc.py)
# and the matrix cores' busy fraction, SQ_VALU_MFMA_BUSY_CYCLES over
# 1 024 SIMDs x GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs' clocks)
FLOP = {'bernoulli': 4.0 * 1e6 * 256 * 32768}
n5 = None
tlog = os.path.join(out, tag + '_nativefull_trace.log')
if os.path.exists(tlog):
    m5 = re.search(r'config5 shape: rows=(\d+) K=(\d+) V=(\d+)',
                   open(tlog).read())
    if m5:
        FLOP['multinomial'] = 4.0 * int(m5.group(1)) * int(m5.group(2)) * \
            int(m5.group(3))
for f in glob.glob(os.path.join(out, tag + '_nativefull_trace', '**',
                                '*kernel_trace.csv'), recursive=True):
    dur = {}
    for r in csv.DictReader(open(f)):
        if is_lik(r['Kernel_Name']):
            dur.setdefault(mode_of(r['Kernel_Name']), []).append(
                (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
    for m, v in sorted(dur.items()):
        fl = FLOP.get(m.split()[0])
        if fl:
            ms = sum(v) / len(v)
            lines.append('derived  %-30s %.2f ms per launch = %.1f TFLOP/s = '
                         '%.3f of 157.3' % (m, ms, fl / ms / 1e9,
                                            fl / ms / 1e9 / 157.3))
for m in sorted({k[0] for k in vals}):
    if (m, 'SQ_VALU_MFMA_BUSY_CYCLES') in vals and (m, 'GRBM_GUI_ACTIVE') in vals:
        lines.append('derived  %-30s MFMA busy %.3f of the SIMD-cycles of the '
                     'launch' % (m, vals[(m, 'SQ_VALU_MFMA_BUSY_CYCLES')] /
                                 (1024.0 * vals[(m, 'GRBM_GUI_ACTIVE')] / 8.0)))
log = os.path.join(out, tag + '_nativefull_trace.log')
if os.path.exists(log):
    lines += [l for l in open(log).read().split('\n') if l.startswith('config')]
txt = '\n'.join(lines)
print(txt)
open(os.path.join(out, tag + '_nativefull_summary.txt'), 'w').write(txt + '\n')
