#!/usr/bin/env python
"""tools/ab_libs_scan.py LIB.so [LIB.so ...] — mdhip_cumtrapz at C5 shape (3 x 1e6 samples, device-resident) and on
27 x 1e4 through several BUILDS of libmdhip.so in one process: kernel time, and the largest deviation from the
sequential float64 sum (numpy) relative to max |I|."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib  # noqa: E402
from mdproptools_amd import backend as B  # noqa: E402

libs = [a for a in sys.argv[1:] if a.endswith(".so")]


def ctx_of(path):
    _lib._lib = None
    _lib.STRICT = False
    _lib.LIB_PATH = os.path.abspath(path)
    return _lib.Context(0)


ctxs = [ctx_of(p) for p in libs]
rng = np.random.default_rng(3)
for shape in ((3, 1_000_000), (27, 10_000), (1, 2049), (5, 4_100_001)):
    yh = np.cumsum(rng.normal(size=shape), axis=1)
    y = torch.from_numpy(yh).cuda()
    ref = np.concatenate([np.zeros((shape[0], 1)), np.cumsum(0.5 * 1e-3 * (yh[:, 1:] + yh[:, :-1]), axis=1)], axis=1)
    for rnd in range(2):
        for p, ctx in zip(libs, ctxs):
            ms = []
            for _ in range(6):
                out = B.cumtrapz(y, 1e-3, leading_zero=True, ctx=ctx)
                ms.append(ctx.last_kernel_ms()[0])
            err = float(np.max(np.abs(out - ref)) / np.max(np.abs(ref)))
            print("%-22s %-14s min %.4f ms  median %.4f ms  max dev %.1e of max|I|" % (
                os.path.basename(p), "x".join(map(str, shape)), min(ms[2:]), float(np.median(ms[2:])), err), flush=True)
