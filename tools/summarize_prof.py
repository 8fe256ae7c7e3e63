#!/usr/bin/env python
"""Condense rocprofv3 CSV output (tools/profile.sh) into one text summary:
per-kernel stats and, for the fused HMC kernel, mean PMC values per launch."""
import csv
import glob
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
lines = []


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


for f in find('%s_trace/**/*kernel_stats.csv' % tag):
    lines.append('== kernel stats (%s)' % os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 12:
                lines.append('  ' + ', '.join(row))

for f in find('%s_trace/**/*kernel_trace.csv' % tag):
    rows = [r for r in csv.DictReader(open(f))
            if 'hmc_diag_normal' in r.get('Kernel_Name', '')]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
         for r in rows]
    if d:
        half = sorted(d[len(d) // 2:])
        lines.append('== fused kernel, per-launch durations from the trace (us): '
                     'all %d launches mean %.2f (burn-in, step-size-search dry '
                     'runs and clock ramp included); second half (steady '
                     'state) mean %.2f median %.2f min %.2f max %.2f' % (
                         len(d), sum(d) / len(d), sum(half) / len(half),
                         half[len(half) // 2], half[0], half[-1]))

# what bench.py itself measured inside the traced run (HIP events around the
# launches of its timed region): must agree with the trace
log = os.path.join(out, '%s_trace.log' % tag)
if os.path.exists(log):
    import json
    for line in open(log):
        if line.startswith('{') and '"roofline"' in line:
            try:
                b = json.loads(line)
            except ValueError:
                continue
            r = b['roofline']
            # the same launches in the trace, by position: bench.py issues
            # [burn-in incl. the step-size search's single-leapfrog dry runs]
            # [settle] [warmup] [TIMED: steps] [back-to-back loop] [other mode]
            try:
                sys.path.insert(0, os.path.dirname(os.path.dirname(
                    os.path.abspath(__file__))))
                import bench as _bench
                med = sorted(d)[len(d) // 2]
                n_search = sum(1 for v in d[:60] if v < 0.8 * med)
                lo = (n_search + _bench.BURN_IN_ADAPT + _bench.SETTLE +
                      b['warmup'])
                win = d[lo:lo + b['steps']]
                if len(win) == b['steps']:
                    w = sorted(win)
                    lines.append(
                        '== the %d launches of bench.py\'s timed region in the '
                        'trace (launches %d..%d by start time; %d search dry '
                        'runs before them): mean %.2f median %.2f us' % (
                            len(win), lo, lo + len(win) - 1, n_search,
                            sum(win) / len(win), w[len(w) // 2]))
            except Exception as e:                      # noqa: BLE001
                lines.append('(timed-region window not located: %r)' % (e,))
            lines.append(
                '== bench.py line of the SAME traced run: roofline.kernel_ms '
                '%.4f (HIP events, %d launches of the timed region), frac '
                '%.3f, ms_per_step %.4f' % (
                    r.get('kernel_ms', float('nan')),
                    r.get('kernel_launches_timed', -1), r['frac'],
                    b['ms_per_step']))

for sub in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_sq2', 'pmc_sq3'):
    for f in find('%s_%s/**/*counter_collection.csv' % (tag, sub)):
        agg = {}
        with open(f) as fh:
            rd = csv.DictReader(fh)
            for row in rd:
                k = row.get('Kernel_Name', '')
                if 'hmc_diag_normal' not in k:
                    continue
                c = row.get('Counter_Name')
                v = float(row.get('Counter_Value', 0))
                key = (c, row.get('Dispatch_Id'))
                agg.setdefault(c, {}).setdefault(row.get('Dispatch_Id'), 0.0)
                agg[c][row.get('Dispatch_Id')] += v
        if agg:
            lines.append('== %s: fused kernel, mean per launch (%s)' % (
                sub, os.path.relpath(f, out)))
            for c, d in sorted(agg.items()):
                vals = list(d.values())
                # skip the first launches (search / burn-in have other L)
                tail = vals[len(vals) // 2:]
                lines.append('  %-24s mean %.6g  (n=%d, all-launch mean %.6g)' % (
                    c, sum(tail) / len(tail), len(tail), sum(vals) / len(vals)))

txt = '\n'.join(lines)
print(txt)
with open(os.path.join(out, '%s_summary.txt' % tag), 'w') as fh:
    fh.write(txt + '\n')
