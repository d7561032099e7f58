#!/usr/bin/env python
"""tools/ab_libs_xcorr.py LIB.so [LIB.so ...] [n] — direct ACF kernel time through several BUILDS of libmdhip.so in one
process (one 1e6-sample series by default), results compared."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib  # noqa: E402
from mdproptools_amd import backend as B  # noqa: E402

libs = [a.split(":")[0] for a in sys.argv[1:] if a.split(":")[0].endswith(".so")]
opts = [dict(kv.split("=") for kv in a.split(":")[1].split(",")) if ":" in a else {} for a in sys.argv[1:] if a.split(":")[0].endswith(".so")]  # LIB.so:key=value,key=value
n = next((int(a) for a in sys.argv[1:] if a.isdigit()), 1_000_000)
METHOD = B.XCORR_FFT if "fft" in sys.argv[1:] else B.XCORR_DIRECT  # `fft`: the FFT estimator (kernel time of the whole pipeline)


def ctx_of(path):
    _lib._lib = None
    _lib.STRICT = False
    _lib.LIB_PATH = os.path.abspath(path)
    return _lib.Context(0)


ctxs = [ctx_of(p) for p in libs]
for c_, o_ in zip(ctxs, opts):
    for k_, v_ in o_.items():
        c_.set_option(k_, int(v_))
x = torch.from_numpy(np.random.default_rng(0).standard_normal((3, n))).cuda()
ref = None
for rnd in range(2):
    for k_lib, (p, ctx) in enumerate(zip(libs, ctxs)):
        best = 1e9
        for rep in range(3 if METHOD == B.XCORR_DIRECT else 10):
            out = B.xcorr(x, method=METHOD, ctx=ctx)
            best = min(best, ctx.last_kernel_ms()[0])
        if ref is None:
            ref = out
        err = float(np.max(np.abs(out[:, : n // 2] - ref[:, : n // 2])) / ref[0, 0])
        print("%-28s %-24s %.4f ms  %.1f TFLOP/s  (diff to first %.1e acf0)" % (os.path.basename(p), ",".join("%s=%s" % kv for kv in opts[k_lib].items()), best, 3 * n * (n + 1.0) / best * 1e-9, err), flush=True)
