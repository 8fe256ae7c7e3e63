#!/usr/bin/env python
"""Workload for tools/profile_native.sh: a few transitions of the native
plans -- BASELINE configs[2] and configs[4] as bench.py builds them (the
reference's literal dense spellings, tuned start), and (round 4) the
dense-logit Categorical (softmax regression) and the gathered-dot rating
model of pmf_hmc.py -- reduced so that a traced run takes seconds, bracketed
by marker launches (min_positive_rows_kernel) so that the kernel trace can be
cut to the transitions alone.
  python tools/native_plan_trace.py [n_rows_config3] [n_chains_config5]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402
from zhusuan_amd import _capi  # noqa: E402
import bench  # noqa: E402

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_chains5 = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device('cuda', 0)
marker = torch.ones(1, 4, device=dev)
marker_out = torch.zeros(1, device=dev)


def mark():
    # twice back to back: nothing else in this run launches this kernel
    # twice in a row (the ESS estimator launches it once per call)
    for _ in range(2):
        _capi.call('zshmc_min_positive_rows', marker.data_ptr(), 1, 4,
                   marker_out.data_ptr(),
                   torch.cuda.current_stream().cuda_stream)


orig = bench._time_transitions


def traced(torch_, hmc, op, info, feed, n_warm, n_timed, barrier):
    for _ in range(n_warm):
        op.run(feed_dict=feed, sync=False)
    hmc.check_numerics()
    torch_.cuda.synchronize()
    mark()
    for _ in range(3):
        op.run(feed_dict=feed, sync=False)
    mark()
    torch_.cuda.synchronize()
    return orig(torch_, hmc, op, info, feed, 0, 2, barrier)


bench._time_transitions = traced
r3 = bench.extra_config3(torch, zs, dev, n_rows=n_rows)
print('config3 slice: %.2f ms/transition, kernel %.3f ms = %.1f TFLOP/s (%.1f%%), plan %s' % (
    r3['ms_per_step'], r3['roofline']['kernel_ms'], r3['roofline']['achieved'],
    100 * r3['roofline']['frac'], r3['plan']))
from zhusuan_amd import _ops  # noqa: E402
_ops.clear_caches()
torch.cuda.empty_cache()
r5 = bench.extra_config5(torch, zs, dev, n_chains=n_chains5)
print('config5 at n_chains=%d: %.2f ms/transition, kernel %.3f ms = %.1f TFLOP/s (%.1f%%), '
      'sustained %.1f TFLOP/s, plan %s' % (
          n_chains5, r5['ms_per_step'], r5['roofline']['kernel_ms'],
          r5['roofline']['achieved'], 100 * r5['roofline']['frac'],
          r5['roofline']['sustained_over_transition'], r5['plan']))
_ops.clear_caches()
torch.cuda.empty_cache()
r6 = bench.extra_softmax_regression(torch, zs, dev, n_rows=6000, n_chains=256)
print('config softmax regression slice: %.2f ms/transition, kernel %.3f ms = %.1f TFLOP/s '
      '(%.1f%%), plan %s' % (r6['ms_per_step'], r6['roofline']['kernel_ms'],
                             r6['roofline']['achieved'],
                             100 * r6['roofline']['frac'], r6['plan']))
_ops.clear_caches()
torch.cuda.empty_cache()
r7 = bench.extra_pmf(torch, zs, dev, n_pairs=200000)
print('config pmf slice: %.2f ms/transition, rating likelihood + scatter %.3f ms = '
      '%.0f GB/s gathered, plan %s' % (r7['ms_per_step'],
                                       r7['roofline']['kernel_ms'],
                                       r7['roofline']['achieved'], r7['plan']))
