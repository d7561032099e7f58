#!/usr/bin/env python
"""tools/ab_xcorr_direct.py [n] — kernel time of the direct correlation estimator for different numbers of time slabs
(option xcorr_tile; 0 = the library's choice)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd._lib import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = torch.from_numpy(np.random.default_rng(0).standard_normal((1, n))).cuda()
ctx = Context(0)
for slabs in [0] + list(range(3, 14)) + [16, 17, 21, 25, 26, 29, 30, 34, 42, 50, 67, 84]:
    ctx.set_option("xcorr_tile", slabs)
    best = 1e9
    for rep in range(2):
        B.xcorr(x, method=B.XCORR_DIRECT, ctx=ctx)
        best = min(best, ctx.last_kernel_ms()[0])
    print("slabs %3d  %.2f ms  %.1f TFLOP/s" % (slabs, best, n * (n + 1.0) / best * 1e-9), flush=True)
