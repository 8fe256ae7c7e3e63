#!/usr/bin/env python
"""configs[2] and the softmax-regression extra of bench.py alone (what the
start-evaluation carry changes: ms per transition), without the headline.
    python tools/archive/carry_probe.py [config3|softmax|wide|all]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import zhusuan_amd as zs  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'config3'
dev = torch.device('cuda', 0)
todo = {'config3': [(bench.extra_config3, {})],
        'softmax': [(bench.extra_softmax_regression, {})],
        'wide': [(bench.extra_wide_regression, {'n_feat': 299,
                                                'n_chains': 16384})]}
todo['all'] = todo['config3'] + todo['softmax'] + todo['wide']
for fn, kw in todo[which]:
    e = fn(torch, zs, dev, **kw)
    r = e.get('roofline', {})
    print(json.dumps({
        'workload': e.get('workload', '')[:60], 'ms_per_step': e.get('ms_per_step'),
        'value': e.get('value'), 'mean_acceptance': e.get('mean_acceptance'),
        'frac': r.get('frac'), 'sustained_frac': r.get('sustained_frac'),
        'launches': r.get('launches_per_transition')}))
