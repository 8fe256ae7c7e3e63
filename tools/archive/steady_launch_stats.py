#!/usr/bin/env python
"""Per-launch durations of the fused kernel from a rocprofv3 kernel trace,
split into the adaptive burn-in (step size and acceptance still moving, the
first 50 transitions + step-size-search launches) and the steady region that
bench.py times.  Usage:
  python tools/archive/steady_launch_stats.py gpurun_out/prof/<tag>_trace/trace_kernel_trace.csv [first_steady]"""
import csv
import sys

import numpy as np

rows = [r for r in csv.DictReader(open(sys.argv[1]))
        if 'hmc_diag_normal' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = np.array([(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
              for r in rows])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 70
print('== fused kernel, per-launch durations from the kernel trace (us)')
print('  all %d launches: mean %.1f' % (len(d), d.mean()))
print('  launches 0..%d (adaptive burn-in, step-size search; acceptance 0 -> '
      '0.9, so the write traffic varies): mean %.1f min %.1f max %.1f' % (
          k - 1, d[:k].mean(), d[:k].min(), d[:k].max()))
print('  launches %d.. (steady, what bench.py times): mean %.1f median %.1f '
      'min %.1f max %.1f' % (k, d[k:].mean(), np.median(d[k:]), d[k:].min(),
                             d[k:].max()))
