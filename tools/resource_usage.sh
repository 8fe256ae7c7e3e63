#!/bin/bash
# Per-kernel VGPR / spill / occupancy table of one HIP source.
# Usage: tools/resource_usage.sh zhusuan_amd/csrc/hmc_fused_ring.hip [extra hipcc flags]
src=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -c "$src" -Rpass-analysis=kernel-resource-usage -o /dev/null "$@" 2>&1 |
  python3 -c '
import re,sys
name=None; row={}
for l in sys.stdin:
    if "error" in l: print(l.strip())
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        if name: print(name,row)
        name=m.group(1); row={}
    for k in ("VGPRs","AGPRs","VGPRs Spill","SGPRs Spill","Occupancy [waves/SIMD]","ScratchSize [bytes/lane]"):
        m=re.search(r"    "+re.escape(k)+r": (\d+)",l)
        if m: row[k]=int(m.group(1))
if name: print(name,row)
'
