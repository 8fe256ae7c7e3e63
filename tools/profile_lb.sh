#!/bin/bash
# rocprofv3 evidence for the fused linear-Bernoulli (fp32 MFMA) kernel at a
# slice of BASELINE config 3 (tools/lb_bench.py).  Output: gpurun_out/prof/<tag>_lb_*
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
CMD="env PYTHONPATH=$REPO python $REPO/tools/lb_bench.py 32768 50000"
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_lb_trace -o trace --output-format csv -- $CMD > $OUT/${TAG}_lb_trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU -d $OUT/${TAG}_lb_pmc -o pmc --output-format csv -- $CMD > $OUT/${TAG}_lb_pmc.log 2>&1
cd $REPO
python - <<PY
import csv, glob, os
out = "$OUT"; tag = "$TAG"
lines = []
for f in glob.glob(os.path.join(out, tag + '_lb_trace', '**', '*kernel_stats.csv'), recursive=True):
    lines.append('== kernel stats (%s)' % os.path.relpath(f, out))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 4: lines.append('  ' + ', '.join(row)[:260])
for f in glob.glob(os.path.join(out, tag + '_lb_pmc', '**', '*counter_collection.csv'), recursive=True):
    agg = {}
    for row in csv.DictReader(open(f)):
        if 'linear_bernoulli' not in row.get('Kernel_Name', ''): continue
        agg.setdefault(row['Counter_Name'], {}).setdefault(row['Dispatch_Id'], 0.0)
        agg[row['Counter_Name']][row['Dispatch_Id']] += float(row['Counter_Value'])
    lines.append('== pmc, mean per launch (%s)' % os.path.relpath(f, out))
    for c, d in sorted(agg.items()):
        v = list(d.values()); lines.append('  %-32s %.6g (n=%d)' % (c, sum(v) / len(v), len(v)))
lines.append(open(os.path.join(out, tag + '_lb_trace.log')).read().strip().split('\n')[-1] if False else '')
txt = '\n'.join(lines)
print(txt)
open(os.path.join(out, tag + '_lb_summary.txt'), 'w').write(txt + '\n')
PY
grep "TFLOP" $OUT/${TAG}_lb_trace.log | tee -a $OUT/${TAG}_lb_summary.txt
