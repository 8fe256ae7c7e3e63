#!/usr/bin/env python
"""The reference's examples/topic_models/lntm_mcem.py (BASELINE config 5
shape) on zhusuan_amd: logistic-normal topic model by Monte-Carlo EM.

  E step   HMC over eta[n_chains, batch, n_topics] -- the hot path.  The
           likelihood is written `zs.log_mixture(theta, phi)` instead of
           `tf.log(tf.matmul(theta, phi))`, which lets the multinomial term
           and its gradient run in the fused fp32-MFMA kernel without ever
           materialising the [rows, n_vocab] product.
  M step   Adam on beta (torch.optim, outside the hot path); with a gradient
           flowing into phi the same model definition takes the dense route.
  Eval     AIS (zs.AIS) perplexity on held-out documents.

There is no network here, so the corpus is synthetic: documents drawn from a
ground-truth logistic-normal topic model of the same shape as the UCI `nips`
bag-of-words set (1 500 docs, 12 419 words).  Sizes shrink with --small.

    python examples/lntm_mcem.py [--small] [--epochs N]
"""
import argparse
import os
import sys
import time
from copy import copy

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zhusuan_amd as zs  # noqa: E402
from zhusuan_amd.evaluation import AIS  # noqa: E402

# Delta in LNTM corresponds to eta in LDA (Blei et al., 2003); a larger
# log_delta gives sparser topics.
log_delta = 10.0


def synthetic_corpus(n_docs, n_vocab, n_true_topics, doc_len, seed):
    rng = np.random.RandomState(seed)
    phi = rng.dirichlet(np.full(n_vocab, 0.02), size=n_true_topics)
    eta = rng.normal(size=(n_docs, n_true_topics)) * 2.0
    theta = np.exp(eta - eta.max(-1, keepdims=True))
    theta /= theta.sum(-1, keepdims=True)
    p = theta @ phi
    lens = rng.poisson(doc_len, size=n_docs)
    X = np.stack([rng.multinomial(n, pi / pi.sum()) for n, pi in zip(lens, p)])
    return X.astype(np.float32)


def make_lntm(dev):
    @zs.meta_bayesian_net(scope='lntm')
    def lntm(n_chains, n_docs, n_topics, n_vocab, eta_mean, eta_logstd):
        bn = zs.BayesianNet()
        eta_mean = eta_mean.unsqueeze(0).expand(n_docs, -1)
        eta = bn.normal('eta', eta_mean, logstd=eta_logstd,
                        n_samples=n_chains, group_ndims=1)
        theta = torch.softmax(eta.tensor, dim=-1)
        beta = bn.normal('beta', torch.zeros(n_topics, n_vocab, device=dev),
                         logstd=log_delta, group_ndims=1)
        phi = torch.softmax(beta.tensor, dim=-1)
        # doc_word = theta @ phi, kept lazy
        bn.unnormalized_multinomial('x', zs.log_mixture(theta, phi),
                                    normalize_logits=False,
                                    dtype=torch.float32)
        return bn
    return lntm


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--epochs', type=int, default=None)
    args = ap.parse_args()
    zs.set_random_seed(1237)
    np.random.seed(1237)
    dev = torch.device('cuda', 0)

    if args.small:
        n_all, training_size, n_vocab, n_topics = 260, 200, 1000, 20
        epochs, _n_temperatures, doc_len = args.epochs or 8, 100, 200
    else:
        n_all, training_size, n_vocab, n_topics = 1500, 1200, 12419, 100
        epochs, _n_temperatures, doc_len = args.epochs or 10, 1000, 1300
    X = synthetic_corpus(n_all, n_vocab, max(n_topics // 2, 5), doc_len, 0)
    X_train, X_test = X[:training_size], X[training_size:]

    # Define model training parameters
    batch_size = 100
    n_chains = 1
    num_e_steps = 5
    hmc = zs.HMC(step_size=1e-3, n_leapfrogs=20, adapt_step_size=True,
                 target_acceptance_rate=0.6)
    learning_rate_0 = 1.0
    t0 = 10

    iters = X_train.shape[0] // batch_size
    Eta = np.zeros((n_chains, X_train.shape[0], n_topics), dtype=np.float32)
    Eta_mean = np.zeros(n_topics, dtype=np.float32)
    Eta_logstd = np.zeros(n_topics, dtype=np.float32)

    # Device buffers standing in for the placeholders / variables
    x = torch.zeros(batch_size, n_vocab, device=dev)
    eta_mean = torch.zeros(n_topics, device=dev)
    eta_logstd = torch.zeros(n_topics, device=dev)
    eta = torch.zeros(n_chains, batch_size, n_topics, device=dev)
    beta = torch.zeros(n_topics, n_vocab, device=dev)

    def e_obj(bn):
        return bn.cond_log_prob('eta') + bn.cond_log_prob('x')

    # E step: sample eta using HMC
    lntm = make_lntm(dev)
    model = lntm(n_chains, batch_size, n_topics, n_vocab, eta_mean, eta_logstd)
    model.log_joint = e_obj
    sample_op, hmc_info = hmc.sample(model, observed={'x': x, 'beta': beta},
                                     latent={'eta': eta})

    # M step: optimise beta
    beta_param = beta.clone().requires_grad_(True)
    optimizer = torch.optim.Adam([beta_param], lr=learning_rate_0)

    def m_step(lr):
        for g in optimizer.param_groups:
            g['lr'] = lr
        optimizer.zero_grad()
        bn = model.observe(eta=eta, x=x, beta=beta_param)
        log_p_beta, log_px = bn.cond_log_prob(['beta', 'x'])
        log_px = log_px.mean(0).sum()
        (-(log_p_beta.sum() + log_px)).backward()
        optimizer.step()
        beta.copy_(beta_param.detach())
        return float(log_px.detach())

    # Evaluation: AIS on the held-out documents
    n_docs_test = X_test.shape[0]
    _n_chains = 25
    _x = torch.tensor(X_test, device=dev)
    _eta = torch.zeros(_n_chains, n_docs_test, n_topics, device=dev)
    _model = lntm(_n_chains, n_docs_test, n_topics, n_vocab, eta_mean,
                  eta_logstd)
    _model.log_joint = e_obj
    proposal_model = copy(_model)
    proposal_model.log_joint = lambda bn: bn.cond_log_prob('eta')
    _hmc = zs.HMC(step_size=0.01, n_leapfrogs=20, adapt_step_size=True,
                  target_acceptance_rate=0.6)
    ais = AIS(_model, proposal_model, _hmc, observed={'x': _x, 'beta': beta},
              latent={'eta': _eta}, n_temperatures=_n_temperatures)

    for epoch in range(1, epochs + 1):
        time_epoch = -time.time()
        learning_rate = learning_rate_0 * (t0 / (t0 + epoch)) ** 2
        perm = np.random.permutation(X_train.shape[0])
        X_train = X_train[perm, :]
        Eta = Eta[:, perm, :]
        lls, accs = [], []
        for t in range(iters):
            sl = slice(t * batch_size, (t + 1) * batch_size)
            x.copy_(torch.from_numpy(X_train[sl]))
            eta.copy_(torch.from_numpy(Eta[:, sl, :]))
            eta_mean.copy_(torch.from_numpy(Eta_mean))
            eta_logstd.copy_(torch.from_numpy(Eta_logstd))
            # E step
            for j in range(num_e_steps):
                sample_op.run()
                accs.append(hmc_info.acceptance_rate.mean().item())
            Eta[:, sl, :] = eta.cpu().numpy()   # the persistent chain
            # M step
            lls.append(m_step(learning_rate))

        # Update hyper-parameters
        Eta_mean = np.mean(Eta, axis=(0, 1))
        Eta_logstd = np.log(np.std(Eta, axis=(0, 1)) + 1e-6)
        time_epoch += time.time()
        print('Epoch {} ({:.1f}s): Perplexity = {:.2f}, acc = {:.3f}, '
              'eta mean = {:.2f}, logstd = {:.2f}'.format(
                  epoch, time_epoch, np.exp(-np.sum(lls) / np.sum(X_train)),
                  np.mean(accs), np.mean(Eta_mean), np.mean(Eta_logstd)))

    # Run AIS
    eta_mean.copy_(torch.from_numpy(Eta_mean))
    eta_logstd.copy_(torch.from_numpy(Eta_logstd))
    time_ais = -time.time()
    ll_lb = ais.run()
    time_ais += time.time()
    print('>> Test log likelihood (AIS, {:.1f}s) = {:.3f}, perplexity = {:.2f}'
          .format(time_ais, ll_lb, np.exp(-ll_lb * n_docs_test / np.sum(X_test))))
    print('   uniform-model perplexity = {}'.format(n_vocab))
