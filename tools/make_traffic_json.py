#!/usr/bin/env python
"""profiles/pmc_traffic.json from a tools/profile.sh summary: HBM bytes per
launch of the fused kernel from the FETCH_SIZE / WRITE_SIZE PMC passes, with
the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section).
Usage: python tools/make_traffic_json.py gpurun_out/prof/<tag>_summary.txt <tag>"""
import json
import re
import sys

path, tag = sys.argv[1], sys.argv[2]
txt = open(path).read()
fetch = float(re.search(r'FETCH_SIZE\s+mean ([0-9.e+]+)', txt).group(1))
write = float(re.search(r'WRITE_SIZE\s+mean ([0-9.e+]+)', txt).group(1))
rd = fetch * 1024.0 * 2.0
wr = write * 1024.0
out = {
    'round': tag,
    'source': 'profiles/%s_rocprofv3_summary.txt (rocprofv3 --pmc FETCH_SIZE / '
              '--pmc WRITE_SIZE, separate passes, bench.py config 2)' % tag,
    'fetch_size_kb_raw': fetch,
    'write_size_kb_raw': write,
    'correction': 'gfx950: FETCH_SIZE counts 128-B requests at 64 B for 16 B/lane '
                  'coalesced streams (global_load and LDS-DMA alike) -> x2 '
                  '(MI355X_MICROARCH.md HBM section); WRITE_SIZE used as '
                  'reported; both in KiB',
    'hbm_read_bytes_per_launch': rd,
    'hbm_write_bytes_per_launch': wr,
    'hbm_bytes_per_launch': rd + wr,
    'algorithmic_bytes_per_launch': 8.0 * 65536 * 1024,
    'note': 'measured < algorithmic because rejected chains are not written '
            'back (mean over the second half of the traced run)',
}
json.dump(out, open('profiles/pmc_traffic.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
