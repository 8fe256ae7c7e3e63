#!/usr/bin/env python
"""Bayesian probabilistic matrix factorisation (the workload of the reference's
PMF example): Gibbs sweeps over chunks of 50 users / 50 items, one HMC update
(K = 8 particles, L = 10) per chunk given the current factors of the other
side.  Per-chunk sizes and index lists are fed through placeholders
(`sess.run(sample_u_op, feed_dict={neighbor_v: ..., select_u: ...})`); the
rating logits are written `zs.gathered_dot(u, select_u, v, select_v)` instead
of two tf.gather + multiply + reduce_sum, which keeps the [K, batch, D]
gathers out of memory and makes the scatter gradient deterministic.

MovieLens-1M is not available offline: ratings are synthesised from a
ground-truth factor model of a similar shape (sizes shrink with --small).

    python examples/matrix_factorization_hmc.py [--small] [--epochs N] [--step-size 1e-3]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402


def make_pmf(dev):
    @zs.meta_bayesian_net(scope="pmf", reuse_variables=True)
    def pmf(n, m, D, n_particles, select_u, select_v, alpha_u, alpha_v,
            alpha_pred):
        bn = zs.BayesianNet()
        mu_u = torch.zeros(int(n.value), D, device=dev)
        u = bn.normal("u", mu_u, std=alpha_u, n_samples=n_particles,
                      group_ndims=1)
        mu_v = torch.zeros(int(m.value), D, device=dev)
        v = bn.normal("v", mu_v, std=alpha_v, n_samples=n_particles,
                      group_ndims=1)
        r_logits = zs.gathered_dot(u, select_u.value, v, select_v.value)
        bn.deterministic("r_pred", torch.sigmoid(r_logits))
        bn.normal("r", torch.sigmoid(r_logits), std=alpha_pred)
        return bn
    return pmf


def synthetic_ratings(N, M, D_true, per_user, seed):
    rng = np.random.RandomState(seed)
    U = rng.normal(size=(N, D_true)) * 0.9
    V = rng.normal(size=(M, D_true)) * 0.9
    pop = rng.dirichlet(np.full(M, 0.5))
    rows = []
    for i in range(N):
        k = max(3, rng.poisson(per_user))
        js = rng.choice(M, size=min(k, M), replace=False, p=pop)
        p = 1 / (1 + np.exp(-(U[i] * V[js]).sum(-1)))
        stars = np.clip(np.rint(1 + 4 * p + 0.35 * rng.normal(size=len(js))),
                        1, 5)
        rows += [(i, j, s) for j, s in zip(js, stars)]
    data = np.array(rows, dtype=np.int64)
    rng.shuffle(data)
    n_tr = int(0.9 * len(data))
    return data[:n_tr], data[n_tr:]


class RatingIndex(object):
    """Ratings grouped by one side (users or items) in CSR form, so that the
    pairs of a chunk of consecutive rows are one slice."""

    def __init__(self, data, side, n_rows):
        order = np.argsort(data[:, side], kind='stable')
        self.other = data[order, 1 - side]
        self.score = data[order, 2].astype(np.float32)
        counts = np.bincount(data[:, side], minlength=n_rows)
        self.start = np.concatenate([[0], np.cumsum(counts)])

    def chunk(self, lo, hi):
        """For rows [lo, hi): the distinct partners, the ratings, and for every
        rating its local row index and its index into the partner list."""
        a, b = self.start[lo], self.start[hi]
        partners, local_other = np.unique(self.other[a:b], return_inverse=True)
        rows = np.repeat(np.arange(hi - lo), np.diff(self.start[lo:hi + 1]))
        return partners, self.score[a:b], rows, local_other


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--epochs', type=int, default=None)
    ap.add_argument('--step-size', type=float, default=1e-3,
                    help='the reference uses 1e-3 (and 500 epochs)')
    args = ap.parse_args()
    np.random.seed(1234)
    zs.set_random_seed(1237)
    torch.manual_seed(1237)
    dev = torch.device('cuda', 0)

    if args.small:
        N, M, per_user, n_epochs = 300, 200, 25, args.epochs or 6
    else:
        N, M, per_user, n_epochs = 6040, 3706, 165, args.epochs or 3
    train_data, test_data = synthetic_ratings(N, M, 8, per_user, 0)

    # set configurations and hyper parameters
    D = 30
    K = 8
    chunk_size = 50
    N = (N + chunk_size - 1) // chunk_size * chunk_size
    M = (M + chunk_size - 1) // chunk_size * chunk_size
    by_user = RatingIndex(train_data, 0, N)
    by_item = RatingIndex(train_data, 1, M)

    # Selection
    i64, f32 = torch.int64, torch.float32
    neighbor_u = zs.placeholder(i64, shape=[None], name="neighbor_u")
    neighbor_v = zs.placeholder(i64, shape=[None], name="neighbor_v")
    select_u = zs.placeholder(i64, shape=[None], name="select_u",
                              default=torch.zeros(1, dtype=i64, device=dev))
    select_v = zs.placeholder(i64, shape=[None], name="select_v",
                              default=torch.zeros(1, dtype=i64, device=dev))
    true_rating = zs.placeholder(f32, shape=[None], name='true_rating',
                                 default=torch.ones(1, device=dev))
    n = zs.placeholder(int, shape=[], name='n', default=chunk_size)
    m = zs.placeholder(int, shape=[], name='m', default=chunk_size)
    alpha_u = 1.0
    alpha_v = 1.0
    alpha_pred = 0.2 / 4.0

    # Samples live in two device buffers (the reference's chunked Variables)
    U = 0.1 * torch.randn(K, N, D, device=dev)
    V = 0.1 * torch.randn(K, M, D, device=dev)
    model = make_pmf(dev)(n, m, D, K, select_u, select_v, alpha_u, alpha_v,
                          alpha_pred)

    normalized_rating = zs.deferred(lambda: (true_rating.value - 1.0) / 4.0)
    target_u = zs.deferred(lambda: U[:, neighbor_u.value])
    target_v = zs.deferred(lambda: V[:, neighbor_v.value])

    def rmse_of(su, sv, tr):
        n.feed(N), m.feed(M)
        select_u.feed(su, dev), select_v.feed(sv, dev)
        with torch.no_grad():
            pred = model.observe(u=U, v=V)["r_pred"].mean(0)
        rating = (torch.as_tensor(tr, dtype=f32, device=dev) - 1.0) / 4.0
        return float(torch.sqrt(torch.mean((pred - rating) ** 2)) * 4)

    hmc_u = zs.HMC(step_size=args.step_size, n_leapfrogs=10, adapt_step_size=None,
                   target_acceptance_rate=0.9)
    hmc_v = zs.HMC(step_size=args.step_size, n_leapfrogs=10, adapt_step_size=None,
                   target_acceptance_rate=0.9)
    candidate_sample_u = 0.1 * torch.randn(K, chunk_size, D, device=dev)
    candidate_sample_v = 0.1 * torch.randn(K, chunk_size, D, device=dev)

    def log_joint(bn):
        log_pu, log_pv = bn.cond_log_prob(['u', 'v'])    # [K, N], [K, M]
        log_pr = bn.cond_log_prob('r')                   # [K, batch]
        return log_pu.sum(-1) + log_pv.sum(-1) + log_pr.sum(-1)

    model.log_joint = log_joint

    # shapes for the build-time evaluation of the joint (chain shape [K])
    neighbor_u.feed(np.arange(chunk_size), dev)
    neighbor_v.feed(np.arange(chunk_size), dev)
    sample_u_op, sample_u_info = hmc_u.sample(
        model, {"r": normalized_rating, "v": target_v},
        {"u": candidate_sample_u})
    sample_v_op, sample_v_info = hmc_v.sample(
        model, {"r": normalized_rating, "u": target_u},
        {"v": candidate_sample_v})

    sess = zs.Session()
    for epoch in range(1, n_epochs + 1):
        epoch_time = -time.time()
        accs = []
        for i in range(N // chunk_size):
            sv, tr, ssu, ssv = by_user.chunk(i * chunk_size,
                                             (i + 1) * chunk_size)
            nv = len(sv)
            if not len(tr):
                continue
            sl = slice(i * chunk_size, (i + 1) * chunk_size)
            candidate_sample_u.copy_(U[:, sl])
            sess.run(sample_u_op, feed_dict={neighbor_v: sv, true_rating: tr,
                                             select_u: ssu, select_v: ssv,
                                             n: chunk_size, m: nv})
            U[:, sl] = candidate_sample_u
            accs.append(float(sample_u_info.acceptance_rate.mean()))
        for i in range(M // chunk_size):
            su, tr, ssv, ssu = by_item.chunk(i * chunk_size,
                                             (i + 1) * chunk_size)
            nu = len(su)
            if not len(tr):
                continue
            sl = slice(i * chunk_size, (i + 1) * chunk_size)
            candidate_sample_v.copy_(V[:, sl])
            sess.run(sample_v_op, feed_dict={neighbor_u: su, true_rating: tr,
                                             select_u: ssu, select_v: ssv,
                                             n: nu, m: chunk_size})
            V[:, sl] = candidate_sample_v
            accs.append(float(sample_v_info.acceptance_rate.mean()))
        epoch_time += time.time()
        print("Epoch {}: {:.1f}s, acc = {:.3f}".format(epoch, epoch_time,
                                                       np.mean(accs)))
        print('>>> Train: rmse = {:.4f}'.format(
            rmse_of(train_data[:, 0], train_data[:, 1], train_data[:, 2])))
        print('>>> Test: rmse = {:.4f}'.format(
            rmse_of(test_data[:, 0], test_data[:, 1], test_data[:, 2])))


if __name__ == "__main__":
    main()
