"""Autocovariance estimator in the FFT form statsmodels documents (oracle/shims/README.md).

acovf(x, demean=False, unbiased=True, fft=True)[k] = sum_t x[t] x[t+k] / (n - k): zero-padded FFT of a
regular length >= 2 n + 1, inverse FFT of |F|^2, first n terms, divided by n - k. Only this call form is
used by the reference (dynamical/residence_time.py:128-130).
"""
import numpy as np
from scipy.fft import next_fast_len


def acovf(x, adjusted=False, demean=True, fft=True, missing="none", nlag=None, unbiased=None):
    if unbiased is not None:
        adjusted = unbiased
    x = np.asarray(x, dtype=np.float64)
    if demean:
        x = x - x.mean()
    n = len(x)
    d = (n - np.arange(n)) if adjusted else np.full(n, n)
    if fft:
        m = next_fast_len(2 * n + 1)
        f = np.fft.fft(x, n=m)
        acov = np.fft.ifft(f * np.conjugate(f))[:n].real / d
    else:
        acov = np.correlate(x, x, "full")[n - 1:] / d
    return acov if nlag is None else acov[: nlag + 1]
