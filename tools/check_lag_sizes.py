#!/usr/bin/env python
"""tools/check_lag_sizes.py LIB.so [key=value ...] — the fused full-lag MSD kernel of one BUILD of libmdhip.so against
the difference kernel (lag_variant 1, exact) over series lengths that cover every transform size 2^9 .. 2^13, full and
truncated lag ranges, ragged groups. For variant builds (tools/build_variant.sh ... -DF3_MIN_M=9 puts the wave-private
kernel on every size). Exit code 1 when a result is outside the bound the library reports."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdproptools_amd import _lib  # noqa: E402
from mdproptools_amd import backend as B  # noqa: E402

lib = [a for a in sys.argv[1:] if a.endswith(".so")][0]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
_lib._lib = None
_lib.LIB_PATH = os.path.abspath(lib)
ctx = _lib.Context(0)
for k, v in opts:
    ctx.set_option(k, int(v))
rng = np.random.default_rng(5)
bad = 0
cases = [(F, None) for F in (257, 300, 511, 512, 513, 700, 1000, 1024, 1025, 1500, 2047, 2048, 2049, 3000, 4095, 4096,
                              4097, 5000, 6000, 8191, 8192)]
cases += [(9000, 7000), (12000, 4000), (16000, 300), (3000, 1000), (6000, 2100), (5000, 3100), (1000, 20)]
for F, max_lag in cases:
    max_lag = F - 1 if max_lag is None else max_lag
    E = int(rng.integers(3, 40))
    cut = int(rng.integers(0, E + 1))
    goff = [0, cut, cut, E]
    r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-500, 500, (1, 3, E))
    ctx.set_option("lag_variant", 1)
    exact = B.lag_msd(r, max_lag, goff, scale=0.5, ctx=ctx)
    ctx.set_option("lag_variant", 2)
    fft = B.lag_msd(r, max_lag, goff, scale=0.5, ctx=ctx)
    bound = ctx.last_rel_bound()
    nz = exact > 0
    rel = float((np.abs(fft[nz] - exact[nz]) / exact[nz]).max()) if nz.any() else 0.0
    ok = rel <= max(bound, 1e-13) and (fft[0] == 0.0).all()
    bad += not ok
    print("F %6d max_lag %6d E %3d  %-22s rel %.2e bound %.2e %s" % (F, max_lag, E, ctx.last_kernel_name(), rel, bound,
                                                                    "ok" if ok else "BAD"), flush=True)
print("check_lag_sizes: %d cases, %d bad" % (len(cases), bad))
sys.exit(1 if bad else 0)
