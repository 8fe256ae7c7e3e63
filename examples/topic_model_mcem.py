#!/usr/bin/env python
"""Logistic-normal topic model fitted by Monte-Carlo EM (the workload of the
reference's LNTM example; BASELINE config 5 shape).

  E step   HMC over the document logits eta[n_chains, batch, n_topics] -- the
           hot path.  The word likelihood is written as in lntm_mcem.py:39-46,
           log(softmax(eta) @ phi): under the sampler the latent is symbolic
           (zhusuan_amd/_symbolic.py), the expression is lowered to the fused
           fp32-MFMA kernel and the [rows, n_vocab] mixture never exists in
           memory (`zs.log_mixture(theta, phi)` is the explicit spelling).
  M step   Adam on the topic logits beta (torch.optim, outside the hot path);
           with a gradient flowing into phi the same model takes the dense
           route automatically.
  Eval     annealed importance sampling (zs.AIS) on held-out documents.

No data set is reachable offline: documents are drawn from a ground-truth
topic model shaped like the UCI `nips` bag-of-words corpus (1 500 documents,
12 419 words); --small shrinks everything.

    python examples/topic_model_mcem.py [--small] [--epochs N]
"""
import argparse
import copy
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402

LOG_DELTA = 10.0      # log-std of the prior on beta; larger -> sparser topics


def draw_corpus(n_docs, n_vocab, n_topics, mean_len, seed):
    rng = np.random.RandomState(seed)
    topics = rng.dirichlet(np.full(n_vocab, 0.02), size=n_topics)
    logits = 2.0 * rng.normal(size=(n_docs, n_topics))
    mix = np.exp(logits - logits.max(-1, keepdims=True))
    mix /= mix.sum(-1, keepdims=True)
    word_p = mix @ topics
    return np.stack([rng.multinomial(rng.poisson(mean_len), p / p.sum())
                     for p in word_p]).astype(np.float32)


class TopicModel(object):
    """Holds the device buffers the model reads (the reference's placeholders
    and variables) and builds the two views used below."""

    def __init__(self, n_topics, n_vocab, device):
        self.K, self.V, self.dev = n_topics, n_vocab, device
        self.prior_mean = torch.zeros(n_topics, device=device)
        self.prior_logstd = torch.zeros(n_topics, device=device)
        self.beta = torch.zeros(n_topics, n_vocab, device=device)

    def net(self, n_chains, n_docs):
        K, V, dev = self.K, self.V, self.dev

        @zs.meta_bayesian_net(scope='lntm')
        def build():
            bn = zs.BayesianNet()
            eta = bn.normal('eta', self.prior_mean.expand(n_docs, K),
                            logstd=self.prior_logstd, n_samples=n_chains,
                            group_ndims=1)
            beta = bn.normal('beta', torch.zeros(K, V, device=dev),
                             logstd=LOG_DELTA, group_ndims=1)
            theta = torch.softmax(eta.tensor, -1)          # lntm_mcem.py:39
            phi = torch.softmax(beta.tensor, -1)
            pred = (theta.reshape(-1, K) @ phi).reshape(
                n_chains, n_docs, V)                       # :40-45
            bn.unnormalized_multinomial('x', torch.log(pred),
                                        normalize_logits=False,
                                        dtype=torch.float32)
            return bn

        model = build()
        model.log_joint = lambda bn: (bn.cond_log_prob('eta') +
                                      bn.cond_log_prob('x'))
        return model

    def set_prior(self, eta_all):
        self.prior_mean.copy_(torch.from_numpy(eta_all.mean((0, 1))))
        self.prior_logstd.copy_(torch.from_numpy(
            np.log(eta_all.std((0, 1)) + 1e-6)))


def fit(args, X_train, tm, device):
    batch, n_chains, e_steps = 100, 1, 5
    n_batches = X_train.shape[0] // batch
    model = tm.net(n_chains, batch)
    x = torch.zeros(batch, tm.V, device=device)
    eta = torch.zeros(n_chains, batch, tm.K, device=device)
    hmc = zs.HMC(step_size=1e-3, n_leapfrogs=20, adapt_step_size=True,
                 target_acceptance_rate=0.6)
    e_step, e_info = hmc.sample(model, observed={'x': x, 'beta': tm.beta},
                                latent={'eta': eta})
    beta_param = tm.beta.clone().requires_grad_(True)
    adam = torch.optim.Adam([beta_param], lr=1.0)

    def m_step(lr):
        adam.param_groups[0]['lr'] = lr
        adam.zero_grad()
        bn = model.observe(eta=eta, x=x, beta=beta_param)
        lp_beta, lp_x = bn.cond_log_prob(['beta', 'x'])
        words = lp_x.mean(0).sum()
        (-(lp_beta.sum() + words)).backward()
        adam.step()
        tm.beta.copy_(beta_param.detach())
        return float(words.detach())

    eta_all = np.zeros((n_chains, X_train.shape[0], tm.K), np.float32)
    n_tokens = X_train.sum()
    for epoch in range(1, args.epochs + 1):
        t0 = time.time()
        lr = (10.0 / (10.0 + epoch)) ** 2
        order = np.random.permutation(X_train.shape[0])
        X_train, eta_all = X_train[order], eta_all[:, order]
        word_ll, acc = 0.0, []
        for b in range(n_batches):
            rows = slice(b * batch, (b + 1) * batch)
            x.copy_(torch.from_numpy(X_train[rows]))
            eta.copy_(torch.from_numpy(eta_all[:, rows]))   # persistent chain
            for _ in range(e_steps):
                e_step.run()
                acc.append(e_info.acceptance_rate.mean().item())
            eta_all[:, rows] = eta.cpu().numpy()
            word_ll += m_step(lr)
        tm.set_prior(eta_all)
        print('Epoch {} ({:.1f}s): Perplexity = {:.2f}, acc = {:.3f}, '
              'prior mean = {:.2f}, logstd = {:.2f}'.format(
                  epoch, time.time() - t0, np.exp(-word_ll / n_tokens),
                  np.mean(acc), float(tm.prior_mean.mean()),
                  float(tm.prior_logstd.mean())))


def held_out_perplexity(args, X_test, tm, device):
    n_chains = 25
    model = tm.net(n_chains, X_test.shape[0])
    prior_only = copy.copy(model)
    prior_only.log_joint = lambda bn: bn.cond_log_prob('eta')
    eta = torch.zeros(n_chains, X_test.shape[0], tm.K, device=device)
    hmc = zs.HMC(step_size=0.01, n_leapfrogs=20, adapt_step_size=True,
                 target_acceptance_rate=0.6)
    ais = zs.AIS(model, prior_only, hmc,
                 observed={'x': torch.tensor(X_test, device=device),
                           'beta': tm.beta},
                 latent={'eta': eta}, n_temperatures=args.temperatures)
    t0 = time.time()
    ll = ais.run()
    print('>> Test log likelihood (AIS, {:.1f}s) = {:.3f}, perplexity = {:.2f}'
          .format(time.time() - t0, ll,
                  np.exp(-ll * X_test.shape[0] / X_test.sum())))
    print('   uniform-model perplexity = {}'.format(tm.V))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--epochs', type=int, default=None)
    args = ap.parse_args()
    zs.set_random_seed(1237)
    np.random.seed(1237)
    device = torch.device('cuda', 0)
    if args.small:
        n_docs, n_train, n_vocab, n_topics, doc_len = 260, 200, 1000, 20, 200
        args.epochs, args.temperatures = args.epochs or 8, 100
    else:
        n_docs, n_train, n_vocab, n_topics, doc_len = 1500, 1200, 12419, 100, 1300
        args.epochs, args.temperatures = args.epochs or 10, 1000
    X = draw_corpus(n_docs, n_vocab, max(n_topics // 2, 5), doc_len, 0)
    tm = TopicModel(n_topics, n_vocab, device)
    fit(args, X[:n_train], tm, device)
    held_out_perplexity(args, X[n_train:], tm, device)


if __name__ == '__main__':
    main()
