#!/usr/bin/env python
"""tools/ab_fft.py [n] [n_series] — kernel time of the FFT correlation estimator (mdhip_xcorr, MDHIP_XCORR_FFT) for the
pass-plan options of csrc/fft_pow2.hip: fft_logr (largest radix of a pass) x fft_logc (columns per tile)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd._lib import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = torch.from_numpy(np.random.default_rng(0).standard_normal((ns, n))).cuda()
ref = None
for logr in (6, 7, 8, 9, 10):
    for logc in (2, 3, 4, 5):
        ctx = Context(0)
        ctx.set_option("fft_logr", logr)
        ctx.set_option("fft_logc", logc)
        best = 1e9
        for rep in range(6):
            out = B.xcorr(x, method=B.XCORR_FFT, ctx=ctx)
            best = min(best, ctx.last_kernel_ms()[0])
        if ref is None:
            ref = out
        err = float(np.abs(out - ref).max() / ref[:, 0].max())
        print("fft_logr %2d fft_logc %d  kernels %.3f ms  (max diff vs first config %.1e acf0)" % (logr, logc, best, err), flush=True)
        ctx.close()
