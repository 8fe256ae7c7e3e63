#!/usr/bin/env python
"""Summary of tools/profile_lb_modes.sh: per instantiation of the likelihood
kernels, launches, mean duration from the rocprofv3 kernel trace and the matrix
cores' busy fraction from the PMC pass (SQ_VALU_MFMA_BUSY_CYCLES over 1 024
SIMDs x GRBM_GUI_ACTIVE / 8: the counter sums the 8 XCDs' clocks)."""
import collections
import csv
import glob
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r'(linear_bernoulli(?:_mid|_wide)?_kernel)<\s*(\d+),\s*(true|'
                  r'false),\s*(\d)(?:,\s*(true|false))?\s*>', name)
    if not m:
        return None
    fam = {'0': 'bernoulli', '1': 'multinomial', '2': 'categorical'}[m.group(4)]
    grad, ll = m.group(3) == 'true', (m.group(5) or 'true') == 'true'
    form = 'll+grad' if grad and ll else 'grad only' if grad else 'll only'
    return '%-28s D=%-5s %-11s %s' % (m.group(1), m.group(2), fam, form)


dur = collections.OrderedDict()
for f in glob.glob(os.path.join(out, tag + '_lbmodes_trace', '**',
                                '*kernel_trace.csv'), recursive=True):
    rows = sorted(csv.DictReader(open(f)),
                  key=lambda r: int(r['Start_Timestamp']))
    for r in rows:
        k = short(r['Kernel_Name'])
        if k:
            dur.setdefault(k, []).append(
                (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
pmc = {}
for f in glob.glob(os.path.join(out, tag + '_lbmodes_pmc', '**',
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
lines = ['%-62s %8s %10s %10s' % ('kernel  width  family  call form', 'launches',
                                  'mean ms', 'MFMA busy')]
for k, v in dur.items():
    busy = ''
    if (k, 'SQ_VALU_MFMA_BUSY_CYCLES') in pmc and (k, 'GRBM_GUI_ACTIVE') in pmc:
        busy = '%.3f' % (pmc[(k, 'SQ_VALU_MFMA_BUSY_CYCLES')] /
                         (1024.0 * pmc[(k, 'GRBM_GUI_ACTIVE')] / 8.0))
    lines.append('%-62s %8d %10.3f %10s' % (k, len(v), sum(v) / len(v), busy))
log = os.path.join(out, tag + '_lbmodes_trace.log')
if os.path.exists(log):
    lines.append('')
    lines += [l for l in open(log).read().split('\n')
              if ' TF = ' in l]
txt = '\n'.join(lines)
print(txt)
open(os.path.join(out, tag + '_lbmodes_summary.txt'), 'w').write(txt + '\n')
