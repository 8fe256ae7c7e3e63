"""Inputs of the two model-family cases of tests/golden/hmc_reference_traces.npz
(`blr`, `lntm`), shared by the generator that runs the reference's own code
(oracle/make_golden_hmc.py) and the test-side restatements
(tests/helpers_hmc_cases.py).  TEST INFRASTRUCTURE (oracle/__init__.py)."""
import numpy as np


def blr_data():
    """Bayesian logistic regression (BASELINE configs[2] family), small."""
    rng = np.random.RandomState(77)
    n_rows, n_feat, n_chains = 40, 8, 12
    X = rng.normal(size=(n_rows, n_feat)).astype(np.float32)
    w_true = rng.normal(size=n_feat).astype(np.float32)
    y = (rng.uniform(size=n_rows) < 1 / (1 + np.exp(-X @ w_true))).astype(
        np.int32)
    w0 = (0.1 * rng.normal(size=(n_chains, n_feat))).astype(np.float32)
    return X, y, w0


def lntm_data():
    """Logistic-normal topic model (BASELINE configs[4] family), small."""
    rng = np.random.RandomState(78)
    n_chains, n_docs, n_topics, n_vocab = 3, 4, 8, 20
    beta = rng.normal(size=(n_topics, n_vocab)).astype(np.float32)
    x = rng.poisson(1.5, size=(n_docs, n_vocab)).astype(np.float32)
    eta_mean = (0.3 * rng.normal(size=n_topics)).astype(np.float32)
    eta_logstd = (0.2 * rng.normal(size=n_topics)).astype(np.float32)
    eta0 = (0.5 * rng.normal(size=(n_chains, n_docs, n_topics))).astype(
        np.float32)
    return beta, x, eta_mean, eta_logstd, eta0


def softmax_regression_data():
    """Multi-class logistic regression with a Categorical likelihood."""
    rng = np.random.RandomState(79)
    n_rows, n_feat, n_cat, n_chains = 30, 5, 4, 10
    X = rng.normal(size=(n_rows, n_feat)).astype(np.float32)
    w_true = rng.normal(size=(n_cat, n_feat)).astype(np.float32)
    y = np.argmax(X @ w_true.T + rng.gumbel(size=(n_rows, n_cat)),
                  axis=-1).astype(np.int32)
    w0 = (0.1 * rng.normal(size=(n_chains, n_cat, n_feat))).astype(np.float32)
    return X, y, w0


def pmf_data():
    """The rating model of pmf_hmc.py, small: 6 particles, 7 users x 5 items,
    4 factors, 20 observed (user, item) pairs with ratings in [0, 1]."""
    rng = np.random.RandomState(80)
    n_particles, n_users, n_items, n_factors, n_pairs = 6, 7, 5, 4, 20
    su = rng.randint(0, n_users, size=n_pairs).astype(np.int32)
    sv = rng.randint(0, n_items, size=n_pairs).astype(np.int32)
    r = rng.uniform(size=n_pairs).astype(np.float32)
    v_obs = (0.5 * rng.normal(size=(n_particles, n_items, n_factors))).astype(
        np.float32)
    u0 = (0.1 * rng.normal(size=(n_particles, n_users, n_factors))).astype(
        np.float32)
    alphas = (1.0, 1.0, 0.2)        # alpha_u, alpha_v, alpha_pred (std)
    return su, sv, r, v_obs, u0, alphas


def bnn_data():
    """Bayesian-neural-network regression (bnn_sgmcmc.py), small: 6 particles,
    a [5, 10, 1] network, one mini-batch of 16 rows of a 100-row train set."""
    rng = np.random.RandomState(81)
    n_particles, layer_sizes, batch, n_train = 6, [5, 10, 1], 16, 100
    x = rng.normal(size=(batch, layer_sizes[0])).astype(np.float32)
    y = rng.normal(size=batch).astype(np.float32)
    ws0, logstds = [], []
    for n_in, n_out in zip(layer_sizes[:-1], layer_sizes[1:]):
        ws0.append(rng.uniform(-2, 2, size=(n_particles, n_out, n_in + 1))
                   .astype(np.float32))
        logstds.append((0.1 * rng.normal(size=(n_out, n_in + 1))).astype(
            np.float32))
    return x, y, ws0, logstds, layer_sizes, n_train


def blr_bias_data():
    """Round 3: logistic regression with TWO weight blocks (7 and 6 features:
    neither a multiple of 4) and a per-chain scalar intercept -- the model
    family of the native plan's packed state."""
    rng = np.random.RandomState(81)
    n_rows, d1, d2, n_chains = 45, 7, 6, 14
    X1 = rng.normal(size=(n_rows, d1)).astype(np.float32)
    X2 = rng.normal(size=(n_rows, d2)).astype(np.float32)
    logit = X1 @ rng.normal(size=d1) + X2 @ rng.normal(size=d2) + 0.4
    y = (rng.uniform(size=n_rows) < 1 / (1 + np.exp(-logit))).astype(np.int32)
    u0 = (0.1 * rng.normal(size=(n_chains, d1))).astype(np.float32)
    v0 = (0.1 * rng.normal(size=(n_chains, d2))).astype(np.float32)
    b0 = (0.1 * rng.normal(size=n_chains)).astype(np.float32)
    return X1, X2, y, u0, v0, b0


def lntm_ragged_data():
    """Round 3: the topic model with K = 6 topics (not a multiple of 4: the
    native plan pads its rows to 8 and keeps the padding out of the
    softmax)."""
    rng = np.random.RandomState(82)
    n_chains, n_docs, n_topics, n_vocab = 3, 5, 6, 23
    beta = rng.normal(size=(n_topics, n_vocab)).astype(np.float32)
    x = rng.poisson(1.5, size=(n_docs, n_vocab)).astype(np.float32)
    # (per topic: lntm_mcem.py:36 tiles it over the documents itself)
    eta_mean = (0.3 * rng.normal(size=n_topics)).astype(np.float32)
    eta_logstd = (0.2 * rng.normal(size=n_topics)).astype(np.float32)
    eta0 = (0.5 * rng.normal(size=(n_chains, n_docs, n_topics))).astype(
        np.float32)
    return beta, x, eta_mean, eta_logstd, eta0
