#!/bin/bash
# Kernel durations of the fused transition kernel's instantiations under
# tools/mass_adapt_probe.py for each library given (rocprofv3 kernel trace):
#   tools/ring_variant_stats.sh "" build/variants/libzshmc_X.so ...
cd /tmp && export TMPDIR=/tmp PROBE_BIG_ONLY=1
for l in "$@"; do
  rm -rf /tmp/prof
  [ -n "$l" ] && l=/root/repo/$l
  PROBE_LIB=$l timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ring --output-format csv -- python /root/repo/tools/mass_adapt_probe.py > /tmp/rp.log 2>&1
  echo "== ${l:-default}"
  python - <<PY
import csv,glob
f=glob.glob("/tmp/prof/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    if "ring_kernel" in row["Name"] or "mass_" in row["Name"]:
        print("%-100s %5s %10.1f us" % (row["Name"][12:112], row["Calls"], float(row["AverageNs"])/1e3))
PY
done
