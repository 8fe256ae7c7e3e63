"""Categorical likelihood + gradient: the exact-fp32 MFMA kernel against the
bf16x3 one (csrc/b3_kernel.h OP 2) on the same operands.
    python tools/cat_b3_bench.py [C K F N]..."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zhusuan_amd import _capi, _ops  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    shapes = [(512, 10, 128, 60000), (1024, 10, 256, 60000),
              (2048, 4, 64, 100000), (256, 32, 256, 60000),
              (512, 16, 192, 60000)]
    for C, K, F, N in shapes:
        G = _ops.class_stride(K)
        D = _ops.likelihood_plan(F, G)[0]
        g = torch.Generator(device=dev).manual_seed(0)
        X = torch.zeros(N, D, device=dev)
        X[:, :F] = torch.randn(N, F, device=dev, generator=g)
        w = torch.zeros(C, G, D, device=dev)
        w[:, :K, :F] = torch.randn(C, K, F, device=dev, generator=g) / F ** .5
        y = torch.randint(0, K, (N,), device=dev, generator=g).float()
        img = _ops.bf16x3_image(X)
        R = C * G
        s = _capi.current_stream()
        out = {}
        for name in ('fp32', 'bf16x3'):
            per_cu = 1 if name == 'fp32' else _ops.resident_per_cu(D, 'bf16x3')
            block = _ops.likelihood_plan(F, G)[1] if name == 'fp32' else 128
            splits = _ops._row_splits(R, N, dev, block, per_cu)
            ws = torch.empty(max(1, splits * R * (D + 1)), device=dev)
            ll = torch.empty(R, device=dev)
            gw = torch.empty(R, D, device=dev)

            def run(want_ll):
                if name == 'fp32':
                    _capi.call('zshmc_linear_categorical_log_lik',
                               w.data_ptr(), X.data_ptr(), y.data_ptr(), R, N,
                               D, K, G, ll.data_ptr() if want_ll else None,
                               gw.data_ptr(), splits, ws.data_ptr(), s)
                else:
                    _capi.call('zshmc_linear_categorical_log_lik_bf16x3',
                               w.data_ptr(), img.data_ptr(), y.data_ptr(), R,
                               N, D, K, G, ll.data_ptr() if want_ll else None,
                               gw.data_ptr(), splits, ws.data_ptr(), s)
            res = []
            for want_ll in (False, True):
                for _ in range(3):
                    run(want_ll)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 10
                for _ in range(n):
                    run(want_ll)
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / n * 1e3)
            out[name] = (res, splits, gw.clone(), ll.clone())
        flops = 4.0 * R * N * D
        e = (out['fp32'][2] - out['bf16x3'][2]).abs().max().item() / \
            out['fp32'][2].abs().max().item()
        print('C=%d K=%d(G=%d) F=%d(D=%d) N=%d: fp32 grad %.3f ms (%.1f TF) '
              'll+grad %.3f ms [splits %d] | bf16x3 grad %.3f ms (%.1f TF) '
              'll+grad %.3f ms [splits %d] | grad rel diff %.1e' % (
                  C, K, G, F, D, N, out['fp32'][0][0],
                  flops / out['fp32'][0][0] / 1e9, out['fp32'][0][1],
                  out['fp32'][1], out['bf16x3'][0][0],
                  flops / out['bf16x3'][0][0] / 1e9, out['bf16x3'][0][1],
                  out['bf16x3'][1], e))


if __name__ == '__main__':
    main()
