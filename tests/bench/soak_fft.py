#!/usr/bin/env python
"""
tests/bench/soak_fft.py [n_max] [seed] — every series length 1..n_max (default 700) plus random long ones through the
FFT correlation estimator (csrc/fft_pow2.hip) against numpy's FFT of the same zero-padded series: cross- and
autocorrelation, batches of 1-5 pairs, a random number of lags. The sums c[k](n-k) must agree within 1e-13 |a||b|.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from mdproptools_amd import backend as B

    n_max = int(sys.argv[1]) if len(sys.argv) > 1 else 700
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    sizes = list(range(1, n_max + 1)) + [int(x) for x in rng.integers(n_max, 3_000_000, 40)]
    worst = 0.0
    for n in sizes:
        nb = int(rng.integers(1, 6)) if n < 200_000 else 1
        a = rng.standard_normal((nb, n)) * rng.choice([1e-3, 1.0, 1e4], (nb, 1))
        b = rng.standard_normal((nb, n)) + rng.uniform(-2, 2)
        L = 2
        while L < 2 * n:
            L *= 2
        n_lags = int(rng.integers(1, n + 1))
        w = (n - np.arange(n_lags))
        for x, y in ((a, b), (a, a)):
            got = B.xcorr(x, None if y is x else y, method=B.XCORR_FFT, n_lags=n_lags)
            ref = np.fft.irfft(np.fft.rfft(x, L) * np.conj(np.fft.rfft(y, L)), L)[:, :n_lags]
            scale = np.linalg.norm(x, axis=1) * np.linalg.norm(y, axis=1)
            err = float((np.abs(got * w - ref).max(axis=1) / scale).max())
            worst = max(worst, err)
            if not err <= 1e-13:
                print("MISMATCH n=%d batch=%d lags=%d auto=%s: %.3e" % (n, nb, n_lags, y is x, err))
                sys.exit(1)
    print("soak_fft: %d lengths (1..%d and 40 up to 3e6), worst |sum error| / (|a||b|) = %.2e" % (len(sizes), n_max, worst))


if __name__ == "__main__":
    main()
