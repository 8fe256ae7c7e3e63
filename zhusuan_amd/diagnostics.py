"""Effective sample size -- the metric definition behind "ESS/s".  Mirrors
reference zhusuan/diagnostics.py:17-64, which is itself pure NumPy on the
host; quirks kept: the rho sum starts at lag 0,
stops at the first negative rho, and `effective_sample_size` returns the
minimum positive ESS over dimensions as a scalar.

`effective_sample_size_batch` is the same estimator vectorised over many
chains/dimensions with FFT autocovariances (identical up to float64
rounding), used by bench.py for ESS/s."""
import numpy as np

__all__ = ['effective_sample_size', 'effective_sample_size_1d',
           'effective_sample_size_batch']


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
