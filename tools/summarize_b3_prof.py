#!/usr/bin/env python
"""Summary of tools/profile_b3.sh: per instantiation of the likelihood kernels
(bf16x3 and fp32), launches, mean duration from the rocprofv3 kernel trace and
the matrix cores' busy fraction from the PMC pass (SQ_VALU_MFMA_BUSY_CYCLES
over 1 024 SIMDs x GRBM_GUI_ACTIVE / 8: the counter sums the 8 XCDs' clocks;
= 32 x the MFMA count for 32x32x16 bf16, MI355X_MICROARCH.md), and the
effective shader clock = GRBM_GUI_ACTIVE / 8 / kernel duration."""
import collections
import csv
import glob
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r'linear_b3_kernel<\s*(\d+),\s*(\d),\s*(true|false)', name)
    if m:
        fam = {'0': 'bernoulli', '1': 'multinomial'}[m.group(2)]
        return 'linear_b3_kernel (bf16x3)     D=%-4s %-11s %s' % (
            m.group(1), fam, 'll+grad' if m.group(3) == 'true' else 'grad only')
    m = re.search(r'(linear_bernoulli_kernel)<\s*(\d+),\s*(true|false),\s*(\d)'
                  r'(?:,\s*(true|false))?\s*>', name)
    if m and m.group(3) == 'true':
        fam = {'0': 'bernoulli', '1': 'multinomial', '2': 'categorical'}[m.group(4)]
        ll = (m.group(5) or 'true') == 'true'
        return 'linear_bernoulli_kernel (f32) D=%-4s %-11s %s' % (
            m.group(2), fam, 'll+grad' if ll else 'grad only')
    if 'b3_split_kernel' in name:
        return 'b3_split_kernel (image)'
    return None


dur = collections.OrderedDict()
for f in glob.glob(os.path.join(out, tag + '_b3_trace', '**',
                                '*kernel_trace.csv'), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    for r in rows:
        k = short(r['Kernel_Name'])
        if k:
            dur.setdefault(k, []).append(
                (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
pmc = {}
for f in glob.glob(os.path.join(out, tag + '_b3_pmc', '**',
                                '*counter_collection.csv'), recursive=True):
    agg = {}
    for row in csv.DictReader(open(f)):
        k = short(row.get('Kernel_Name', ''))
        if not k:
            continue
        d = agg.setdefault((k, row['Counter_Name']), {})
        d[row['Dispatch_Id']] = d.get(row['Dispatch_Id'], 0.0) + \
            float(row['Counter_Value'])
    for (k, c), d in agg.items():
        pmc[(k, c)] = sum(d.values()) / len(d)
lines = ['%-66s %8s %10s %10s %10s' % ('kernel  width  family  call form',
                                      'launches', 'mean ms', 'MFMA busy',
                                      'clock GHz')]
for k, v in dur.items():
    busy = clk = ''
    ms = sum(v) / len(v)
    if (k, 'SQ_VALU_MFMA_BUSY_CYCLES') in pmc and (k, 'GRBM_GUI_ACTIVE') in pmc:
        act = pmc[(k, 'GRBM_GUI_ACTIVE')] / 8.0
        busy = '%.3f' % (pmc[(k, 'SQ_VALU_MFMA_BUSY_CYCLES')] / (1024.0 * act))
        clk = '%.2f' % (act / (ms * 1e-3) / 1e9)
    lines.append('%-66s %8d %10.3f %10s %10s' % (k, len(v), ms, busy, clk))
lines.append('')
lines.append('(clock: GRBM_GUI_ACTIVE of the PMC pass over the duration of the '
             'trace pass -- two runs, indicative)')
log = os.path.join(out, tag + '_b3_trace.log')
if os.path.exists(log):
    lines.append('')
    lines += [l.rstrip() for l in open(log) if 'bf16x3' in l or 'fp32 ' in l
              or l.startswith('#')]
text = '\n'.join(lines)
print(text)
open(os.path.join(out, tag + '_b3_rocprofv3_summary.txt'), 'w').write(text + '\n')
