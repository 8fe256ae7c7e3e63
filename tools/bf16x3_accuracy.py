#!/usr/bin/env python
"""CPU study for DESIGN 9's next lever: how close does a 3 x bf16 split of both
operands (six bf16 x bf16 products per term, float32 accumulation -- what six
`v_mfma_f32_*_bf16` per fp32 MFMA would compute) come to the float32 kernel,
for the two GEMMs of the logistic-regression likelihood (logits S = X W^T,
gradient G = R^T X with R = y - sigmoid(S))?  Reference: float64.
    python tools/bf16x3_accuracy.py [n_rows] [n_features] [n_chains]
No GPU; NumPy emulation (bf16 = float32 with the low 16 mantissa bits rounded
to nearest even away; products of two bf16 values are exact in float32)."""
import sys

import numpy as np


def bf16(x):
    """Round float32 to bfloat16 (nearest even), returned as float32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    hi = bf16(x)
    mid = bf16(x - hi)
    lo = bf16(x - hi - mid)
    return hi, mid, lo


def matmul_f32(a, b):
    """float32 accumulate in K-blocks of 8 (a stand-in for an MFMA chain:
    pairwise inside a block, sequential across blocks)."""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(0, a.shape[1], 8):
        acc += (a[:, k:k + 8] @ b[k:k + 8]).astype(np.float32)
    return acc


def matmul_bf16x3(a, b, terms=6):
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    pairs = [(ah, bh), (ah, bm), (am, bh), (ah, bl), (al, bh), (am, bm),
             (am, bl), (al, bm), (al, bl)][:terms]
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    # small terms first: what an accumulation order of choice would do
    for x, y in reversed(pairs):
        acc += matmul_f32(x, y)
    return acc


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    c = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    rng = np.random.default_rng(0)
    X = rng.standard_normal((n, d)).astype(np.float32)
    w_true = rng.standard_normal(d) / np.sqrt(d)
    y = (rng.uniform(size=n) < 1 / (1 + np.exp(-X @ w_true))).astype(np.float32)
    W = (w_true + 0.3 * rng.standard_normal((c, d)) / np.sqrt(d)).astype(np.float32)

    S64 = X.astype(np.float64) @ W.astype(np.float64).T
    R64 = y[:, None] - 1 / (1 + np.exp(-S64))
    G64 = R64.T @ X.astype(np.float64)
    ll64 = (S64 * y[:, None] - np.logaddexp(0, S64)).sum(0)

    def report(name, S):
        S = S.astype(np.float64)
        R = (y[:, None] - 1 / (1 + np.exp(-S))).astype(np.float32)
        if name.startswith('bf16'):
            G = matmul_bf16x3(np.ascontiguousarray(R.T), X,
                              int(name.split('/')[1]))
        else:
            G = matmul_f32(np.ascontiguousarray(R.T), X)
        ll = (S * y[:, None] - np.logaddexp(0, S)).sum(0)
        print('%-10s logits max |err| %.2e   gradient max rel err %.2e   '
              'log-lik max |err| %.2e (of %.0f)' % (
                  name, np.abs(S - S64).max(),
                  (np.abs(G - G64) / np.abs(G64).max()).max(),
                  np.abs(ll - ll64).max(), np.abs(ll64).max()))

    print('%d rows x %d features, %d chains' % (n, d, c))
    report('float32', matmul_f32(X, np.ascontiguousarray(W.T)))
    for terms in (3, 6, 9):
        report('bf16x3/%d' % terms,
               matmul_bf16x3(X, np.ascontiguousarray(W.T), terms))


if __name__ == '__main__':
    main()
