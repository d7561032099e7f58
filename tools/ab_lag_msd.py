#!/usr/bin/env python
"""tools/ab_lag_msd.py [E] [F] — kernel time of the full-lag MSD difference kernel (lag_variant 1) at C4 shape (default a
fifth of its entities) and its agreement with the FFT path."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd._lib import Context  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rng = np.random.default_rng(0)
r = torch.from_numpy(np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0)).cuda()
ctx = Context(0)
ctx.set_option("lag_variant", 1)
for rep in range(3):
    out = B.lag_msd(r, F - 1, [0, E], ctx=ctx)
    ms = ctx.last_kernel_ms()[0]
    pairs = F * (F - 1) / 2 * E
    print("%s  %.2f ms  %.3g frame-pair-entities/s  (sub+fma pairs: %.1f T/s)" % (ctx.last_kernel_name(), ms, pairs / ms * 1e3, 3 * pairs / ms * 1e-9))
ctx.set_option("lag_variant", 2)
fft = B.lag_msd(r, F - 1, [0, E], ctx=ctx)
print("max rel diff to the FFT path: %.2e" % float(np.max(np.abs(out[1:] - fft[1:]) / np.abs(out[1:]))))
