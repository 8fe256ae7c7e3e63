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


def mode_of(name):
    tail = name.split('linear_bernoulli_kernel')[1][:60]
    return 'multinomial' if re.search(r',\s*1>|Li1E', tail) else 'bernoulli'


for f in glob.glob(os.path.join(out, tag + '_nativefull_trace', '**',
                                '*kernel_trace.csv'), recursive=True):
    dur = {}
    for r in csv.DictReader(open(f)):
        if 'linear_bernoulli_kernel' in r['Kernel_Name']:
            dur.setdefault(mode_of(r['Kernel_Name']), []).append(
                (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
    for m, v in sorted(dur.items()):
        lines.append('kernel trace  %-12s launches %d  ms each: %s' % (
            m, len(v), ' '.join('%.2f' % x for x in v)))
vals = {}
for kind in ('fetch', 'write', 'mfma'):
    for f in glob.glob(os.path.join(out, '%s_nativefull_%s' % (tag, kind),
                                    '**', '*counter_collection.csv'),
                       recursive=True):
        agg = {}
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '')
            if 'linear_bernoulli_kernel' not in k:
                continue
            key = (mode_of(k), row['Counter_Name'])
            agg.setdefault(key, {}).setdefault(row['Dispatch_Id'], 0.0)
            agg[key][row['Dispatch_Id']] += float(row['Counter_Value'])
        for (m, c), d in sorted(agg.items()):
            v = list(d.values())
            vals[(m, c)] = sum(v) / len(v)
            lines.append('pmc  %-12s %-28s mean per launch %.6g (n=%d)' % (
                m, c, vals[(m, c)], len(v)))
for m in ('bernoulli', 'multinomial'):
    if (m, 'FETCH_SIZE') in vals and (m, 'WRITE_SIZE') in vals:
        rd = vals[(m, 'FETCH_SIZE')] * 1024 * 2     # gfx950: half-count
        wr = vals[(m, 'WRITE_SIZE')] * 1024
        lines.append('HBM traffic  %-12s read %.4g B (FETCH_SIZE x 2 KiB) + '
                     'write %.4g B = %.4g B per launch' % (m, rd, wr, rd + wr))
log = os.path.join(out, tag + '_nativefull_trace.log')
if os.path.exists(log):
    lines += [l for l in open(log).read().split('\n') if l.startswith('config')]
txt = '\n'.join(lines)
print(txt)
open(os.path.join(out, tag + '_nativefull_summary.txt'), 'w').write(txt + '\n')
