"""NumPy float32 restatement of the Bayesian-neural-network regression model
of the reference's examples/bayesian_neural_nets/bnn_sgmcmc.py -- log joint
and its gradient with respect to the weight matrices.  TEST INFRASTRUCTURE
(see oracle/__init__.py).

Follows, in /root/reference:
  examples/bayesian_neural_nets/bnn_sgmcmc.py:19-35   build_bnn: particles of
      w_i[n_out, n_in + 1] ~ N(0, exp(logstd_i)) (group_ndims = 2), a column
      of ones appended to the activations, h <- einsum('imk,ijk->ijm', w, h)
      / sqrt(n_in + 1), ReLU between layers, y ~ N(h, exp(-0.95))
  bnn_sgmcmc.py:71-76   log_joint = sum_i log p(w_i) + mean_batch(log p(y))
      * n_train
  zhusuan/distributions/univariate.py:174-181   Normal._log_prob
Pinned by tests/test_oracle_sgmcmc_reference.py to traces of the reference's
own sgmcmc.py sampling the reference's own build_bnn (imported from the
unmodified example file; oracle/make_golden_sgmcmc.py)."""
import numpy as np

F32 = np.float32
Y_LOGSTD = F32(-0.95)
_C = F32(-0.5 * np.log(2 * np.pi))


def _normal_lp(x, mean, logstd):
    prec = np.exp(F32(-2) * logstd)
    return (_C - logstd - F32(0.5) * prec * np.square(x - mean)).astype(F32)


def forward(ws, x):
    """Activations per layer (inputs with the ones column) and y_mean."""
    n_particles = ws[0].shape[0]
    h = np.broadcast_to(x[None], (n_particles,) + x.shape).astype(F32)
    inputs, pre = [], []
    for i, w in enumerate(ws):
        h = np.concatenate([h, np.ones(h.shape[:-1] + (1,), F32)], -1)
        inputs.append(h)
        z = (np.einsum('imk,ijk->ijm', w, h) /
             np.sqrt(F32(h.shape[2]))).astype(F32)
        pre.append(z)
        h = np.maximum(z, 0) if i < len(ws) - 1 else z
    return inputs, pre, h[..., 0]


def log_joint(ws, x, y, logstds, n_train):
    _, _, y_mean = forward(ws, x)
    lp = sum(_normal_lp(w, F32(0), ls).sum((-2, -1), dtype=F32)
             for w, ls in zip(ws, logstds))
    return (lp + _normal_lp(y, y_mean, Y_LOGSTD).mean(1, dtype=F32) *
            F32(n_train)).astype(F32)


def grad_log_joint(ws, x, y, logstds, n_train):
    inputs, pre, y_mean = forward(ws, x)
    batch = x.shape[0]
    prec_y = np.exp(F32(-2) * Y_LOGSTD)
    g = ((y - y_mean) * prec_y * F32(n_train) / F32(batch))[..., None]
    grads = [None] * len(ws)
    for i in reversed(range(len(ws))):
        h = inputs[i]
        scale = np.sqrt(F32(h.shape[2]))
        if i < len(ws) - 1:
            g = g * (pre[i] > 0)
        grads[i] = (np.einsum('ijm,ijk->imk', g, h) / scale).astype(F32)
        g = (np.einsum('ijm,imk->ijk', g, ws[i]) / scale)[..., :-1]
    return [(gw - np.exp(F32(-2) * ls) * w).astype(F32)
            for gw, w, ls in zip(grads, ws, logstds)]
