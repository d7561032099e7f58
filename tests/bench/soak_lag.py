#!/usr/bin/env python
"""tests/bench/soak_lag.py [trials] [seed] — random shapes through the three sources of the fused full-lag MSD kernel
(`lag_direct` 0: transposed copy, 1: read in place, 2: clusters transposing their tiles inside the kernel): frames
2049 .. 8192 (the range the in-kernel form takes: five staging units per lane up to 5120 frames, eight beyond), entities 1 .. 1500 (odd and even column counts, fewer columns than
clusters), one to six groups with empty and one-entity ones, random scale. Every result must agree with the transposed
path within the sum of the two reported bounds, and a second call must reproduce the first bit for bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdproptools_amd import backend as B  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = B.default_context()
ctx.set_option("lag_variant", 2)
took = {0: 0, 1: 0, 2: 0}
try:
    for t in range(trials):
        F = int(rng.integers(2049, 8193))
        E = int(rng.choice([1, 2, 5, 16, 17, 85, 86, 300, 333, 1024, 1500, int(rng.integers(1, 1500))]))
        G = int(rng.integers(1, 7))
        cuts = np.sort(rng.integers(0, E + 1, G - 1)) if G > 1 else np.array([], dtype=np.int64)
        goff = [0] + [int(c) for c in cuts] + [E]
        scale = float(rng.choice([1.0, 0.7, 1e-10]))
        max_lag = F - 1 if rng.random() < 0.7 else int(rng.integers(1, F))
        r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-50, 50, (1, 3, E))
        ctx.set_option("lag_direct", 0)
        ref = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
        b0 = ctx.last_rel_bound()
        for mode in (2, 1):
            ctx.set_option("lag_direct", mode)
            got = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
            b1 = ctx.last_rel_bound()
            again = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
            assert np.array_equal(got, again), ("not reproducible", mode, F, E, goff)
            nz = ref > 0
            err = float((np.abs(got[nz] - ref[nz]) / ref[nz]).max()) if nz.any() else 0.0
            assert err <= b0 + b1 + 1e-15, (mode, F, E, goff, max_lag, err, b0, b1)
            assert (got[0] == 0.0).all()
        if (t + 1) % 10 == 0:
            print("trial %d ok (F %d E %d groups %s max_lag %d)" % (t + 1, F, E, goff, max_lag), flush=True)
finally:
    ctx.set_option("lag_variant", -1)
    ctx.set_option("lag_direct", -1)
print("soak_lag: %d shapes, the three sources agree within their bounds, every call reproducible" % trials)
