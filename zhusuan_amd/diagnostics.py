"""Effective sample size -- the metric definition behind "ESS/s".  Mirrors
reference zhusuan/diagnostics.py:17-64, which is itself pure NumPy on the
host; quirks kept: the rho sum starts at lag 0,
stops at the first negative rho, and `effective_sample_size` returns the
minimum positive ESS over dimensions as a scalar.

`effective_sample_size_batch` is the same estimator vectorised over many
chains/dimensions with FFT autocovariances (identical up to float64
rounding).  `effective_sample_size_device` runs it on the GPU for every
(chain, dimension) series of a block of recorded draws
(csrc/diagnostics.hip); bench.py uses it for ESS/s over the whole chain
population."""
import numpy as np

__all__ = ['effective_sample_size', 'effective_sample_size_1d',
           'effective_sample_size_batch', 'effective_sample_size_device']


def effective_sample_size_1d(samples):
    """ESS of one scalar chain (diagnostics.py:17-40)."""
    samples = np.asarray(samples)
    n = samples.shape[0]
    mu_hat = np.mean(samples)
    var = np.var(samples) * n / (n - 1)
    var_plus = var * (n - 1) / n
    centred = samples - mu_hat
    sum_rho = 0
    for t in range(0, n):
        rho = 1 - (var - np.mean(centred[:n - t] * centred[t:])) / var_plus
        if rho < 0:
            break
        sum_rho += rho
    return n / (1 + 2 * sum_rho)


def effective_sample_size(samples, burn_in=100):
    """ESS of a chain of vector samples [M, D]: minimum positive ESS over
    dimensions after dropping `burn_in` rows (diagnostics.py:43-64)."""
    samples = np.asarray(samples)
    current_ess = np.inf
    for d in range(samples.shape[1]):
        ess = effective_sample_size_1d(np.squeeze(samples[burn_in:, d]))
        assert ess >= 0
        if ess > 0:
            current_ess = min(current_ess, ess)
    return current_ess


def effective_sample_size_batch(samples, burn_in=100):
    """Same estimator for samples [M, ...]: returns an array of ESS with the
    trailing shape.  Autocovariances at all lags come from one FFT."""
    x = np.asarray(samples, dtype=np.float64)[burn_in:]
    n = x.shape[0]
    trailing = x.shape[1:]
    x = x.reshape(n, -1)
    mu = x.mean(axis=0)
    var = x.var(axis=0) * n / (n - 1)
    var_plus = var * (n - 1) / n
    c = x - mu
    nfft = 1 << int(np.ceil(np.log2(2 * n)))
    f = np.fft.rfft(c, n=nfft, axis=0)
    acov = np.fft.irfft(f * np.conj(f), n=nfft, axis=0)[:n]
    acov /= (n - np.arange(n))[:, None]          # mean over the n-t products
    with np.errstate(divide='ignore', invalid='ignore'):
        rho = 1 - (var[None] - acov) / var_plus[None]
    neg = rho < 0
    first_neg = np.where(neg.any(axis=0), neg.argmax(axis=0), n)
    mask = np.arange(n)[:, None] < first_neg[None]
    sum_rho = np.where(mask, rho, 0.0).sum(axis=0)
    return (n / (1 + 2 * sum_rho)).reshape(trailing)


def effective_sample_size_device(samples, burn_in=100, per_dimension=False):
    """The reference estimator on the device.

    :param samples: float32 device tensor ``[M, chain axes..., D]``: M
        recorded draws of the state (what ``effective_sample_size`` takes as
        ``[M, D]`` for ONE chain).
    :param burn_in: rows dropped from the front (diagnostics.py:43 default).
    :param per_dimension: return the ESS of every series (``[chain axes..., D]``)
        instead of the minimum positive ESS over the last axis per chain
        (``[chain axes...]``, diagnostics.py:55-64).
    """
    import torch
    from . import _capi
    if not (torch.is_tensor(samples) and samples.is_cuda):
        raise TypeError('effective_sample_size_device needs a device tensor; '
                        'use effective_sample_size(_batch) on the host')
    if samples.dim() < 2:
        raise ValueError('samples must be [M, ..., D]')
    x = samples[burn_in:].to(torch.float32).contiguous()
    n = int(x.shape[0])
    trailing = tuple(x.shape[1:])
    n_series = 1
    for t in trailing:
        n_series *= int(t)
    ess = torch.empty(n_series, dtype=torch.float32, device=x.device)
    _capi.call('zshmc_ess_series', x.data_ptr(), n, n_series, ess.data_ptr(),
               _capi.current_stream())
    if per_dimension:
        return ess.reshape(trailing)
    cols = int(trailing[-1])
    rows = n_series // cols
    out = torch.empty(rows, dtype=torch.float32, device=x.device)
    _capi.call('zshmc_min_positive_rows', ess.data_ptr(), rows, cols,
               out.data_ptr(), _capi.current_stream())
    return out.reshape(trailing[:-1])
