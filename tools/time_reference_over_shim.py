#!/usr/bin/env python
"""Time the reference's OWN zhusuan/hmc.py (+ its own model layer), executed
unmodified over oracle/tf_shim.py on torch-CPU with every host thread, on a
slice of BASELINE configs[1] (4 096 chains x 1 024-D diagonal Gaussian,
L = 10).  This is the closest obtainable reading of north_star's "ZhuSuan's own
TF-CPU HMC timed on the host cores" -- TensorFlow is not installable -- and it
can only run where /root/reference exists, i.e. in the BUILD container (the GPU
box has no copy of the reference and the sources may not be vendored):
bench.py therefore carries this number as a recorded value
(profiles/cpu_reference_over_shim.json, `measured_on` says where), next to
the baselines it measures live on the GPU box's own cores.

    python tools/time_reference_over_shim.py [seconds] [C D L out.json]
(e.g. `10 1000 10 5 profiles/cpu_reference_over_shim_config1.json`:
BASELINE configs[0], the gaussian.py shape)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tf_shim  # noqa: E402
from oracle.make_golden_hmc import gaussian_model, load_reference  # noqa: E402

C, D, L = 4096, 1024, 10
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
OUT = os.path.join(ROOT, 'profiles', 'cpu_reference_over_shim.json')
if len(sys.argv) > 5:
    C, D, L = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    OUT = os.path.join(ROOT, sys.argv[5])
threads = os.cpu_count()
torch.set_num_threads(threads)
tf, zs = load_reference()
tf_shim._VARS[:] = []
tf_shim.end_replay()
logstd = np.linspace(-1, 1, D).astype(np.float32)
if D == 10:      # gaussian.py:29
    logstd = np.log(1.0 / (np.arange(D, dtype=np.float32) + 1.0))
model = gaussian_model(np.zeros(D, np.float32), logstd)(tf, zs, C)
x = tf.Variable(np.zeros((C, D), np.float32), name='x')
hmc = zs.hmc.HMC(step_size=0.14 if D != 10 else 0.05, n_leapfrogs=L)
# tf.random_normal / tf.random_uniform: torch's generators (a TF kernel would
# also be a fast C++ Philox; the NumPy Philox of the parity harness would
# dominate the time and is not what is being measured)
tf_shim.set_random_source(lambda s: torch.randn(*s).numpy(),
                          lambda s: torch.rand(*s).numpy())
mark = tf_shim.variable_mark()
n = 0
hmc.sample(model, {}, {'x': x})            # "graph construction" + first run
tf_shim.end_replay()
t0 = time.perf_counter()
while True:
    tf_shim.begin_run(mark)
    _, info = hmc.sample(model, {}, {'x': x})      # = one sess.run(sample_op)
    tf_shim.end_replay()
    n += 1
    el = time.perf_counter() - t0
    if el > budget:
        break
acc = float(info.acceptance_rate.mean())
out = {
    'value': C * L * n / el,
    'unit': 'chain-leapfrog-steps/s',
    'cores': threads,
    'kind': 'reference',
    'what': "the reference's own zhusuan/hmc.py + framework/bn.py + "
            "distributions/univariate.py, unmodified, over oracle/tf_shim.py "
            "(eager float32 torch-CPU stand-in for the TensorFlow ops), "
            "torch.set_num_threads(%d)" % threads,
    'sample': '%d chains x %d latents, L=%d, %d transitions in %.1f s, mean '
              'acceptance %.3f' % (C, D, L, n, el, acc),
    'measured_on': 'build container (%d host threads), NOT the GPU box: the '
                   'reference sources do not travel there' % threads,
    'elem_leapfrog_steps_per_sec': C * D * L * n / el,
    'transitions_per_sec': n / el,
}
json.dump(out, open(OUT, 'w'), indent=1)
print(json.dumps(out, indent=1))
