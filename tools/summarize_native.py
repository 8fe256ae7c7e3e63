#!/usr/bin/env python
"""Summary of tools/profile_native.sh: per-kernel statistics, the kernels
launched between the marker pairs (= inside the native plans' transitions),
and the MFMA counters per launch of the likelihood kernel."""
import collections
import csv
import glob
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
lines = []
for f in glob.glob(os.path.join(out, tag + '_native_trace', '**', '*kernel_stats.csv'), recursive=True):
    lines.append('== kernel stats, whole run (%s)' % os.path.relpath(f, out))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 14:
            lines.append('  ' + ', '.join(row)[:230])
for f in glob.glob(os.path.join(out, tag + '_native_trace', '**', '*kernel_trace.csv'), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    inside, section, sections = False, collections.OrderedDict(), []
    # a boundary is TWO marker launches back to back (the ESS estimator
    # launches the same kernel, one at a time)
    is_mark = ['min_positive_rows_kernel' in r['Kernel_Name'] for r in rows]
    skip = False
    for i, r in enumerate(rows):
        name = r['Kernel_Name']
        if skip:
            skip = False
            continue
        if is_mark[i] and i + 1 < len(rows) and is_mark[i + 1]:
            if inside:
                sections.append(section)
                section = collections.OrderedDict()
            inside = not inside
            skip = True
            continue
        if inside:
            # 'void (anonymous namespace)::softmax_warp_forward<...>(...)':
            # cut at the argument list, not at the first parenthesis
            key = re.sub(r'\(anonymous namespace\)', '{anon}', name)
            key = key.split('(')[0][:110]
            d = section.setdefault(key, [0, 0.0])
            d[0] += 1
            d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
    names = ['config 3: dense-logit Bernoulli', 'config 5: mixture multinomial',
             'softmax regression: dense-logit Categorical',
             'pmf_hmc.py: gathered-dot rating model']
    for i, sec in enumerate(sections):
        lines.append('== kernels launched inside 3 transitions of native plan %d '
                     '(%s): name, launches, total us' % (
                         i, names[i] if i < len(names) else '?'))
        for k, (n, us) in sorted(sec.items(), key=lambda kv: -kv[1][1]):
            lines.append('  %-112s %5d %12.1f' % (k, n, us))
        # anything that is neither this library's nor the runtime's copy /
        # fill blit is somebody else's kernel (ATen, ...)
        foreign = [k for k in sec if 'zshmc::' not in k and
                   '__amd_rocclr' not in k]
        lines.append('  kernels that are neither zshmc:: nor runtime blits '
                     '(ATen etc.) inside the transitions: %d%s' % (
                         len(foreign), ''.join('\n    ' + k for k in foreign)))
for f in glob.glob(os.path.join(out, tag + '_native_pmc', '**', '*counter_collection.csv'), recursive=True):
    agg = {}
    for row in csv.DictReader(open(f)):
        k = row.get('Kernel_Name', '')
        if 'linear_bernoulli_kernel' not in k:
            continue
        tail = k.split('linear_bernoulli_kernel')[1][:48]
        mode = 'multinomial' if re.search(r',\s*1>|Li1E', tail) else 'bernoulli'
        key = (mode, row['Counter_Name'])
        agg.setdefault(key, {}).setdefault(row['Dispatch_Id'], 0.0)
        agg[key][row['Dispatch_Id']] += float(row['Counter_Value'])
    lines.append('== pmc, mean per launch of the likelihood kernel (%s)' % os.path.relpath(f, out))
    for (mode, c), d in sorted(agg.items()):
        v = list(d.values())
        lines.append('  %-12s %-34s %.6g (n=%d)' % (mode, c, sum(v) / len(v), len(v)))
log = os.path.join(out, tag + '_native_trace.log')
if os.path.exists(log):
    lines += [l for l in open(log).read().split('\n') if l.startswith('config')]
txt = '\n'.join(lines)
print(txt)
open(os.path.join(out, tag + '_native_summary.txt'), 'w').write(txt + '\n')
