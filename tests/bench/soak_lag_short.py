#!/usr/bin/env python
"""tests/bench/soak_lag_short.py [trials] [seed] — random shapes through the one-wave-per-series full-lag MSD kernel
(msd_power_w1_kernel: F <= 1536 frames, F + max_lag <= 3072; padded length 1024 / 2048 / 3072) against the block-wide
kernels of rounds 2-5 (`lag_w1` 0): frames 2 .. 1536 (weight on the limits 512 / 1024 / 1536 of the three padded lengths),
entities 1 .. 3000 (fewer series than one block's twelve waves, more than one block per CU), one to six groups with empty
and one-entity ones, random scale and max_lag, batches of 1 MB now and then. Agreement within the sum of the two reported
bounds, every fifth shape against the exact-difference kernel within the bound, bit-for-bit reproducibility."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdproptools_amd import backend as B  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = B.default_context()
took = 0
try:
    for t in range(trials):
        F = int(rng.choice([2, 3, 511, 512, 513, 1023, 1024, 1025, 1535, 1536, int(rng.integers(2, 1537)), int(rng.integers(2, 1537))]))
        E = int(rng.choice([1, 2, 5, 11, 12, 13, 64, 300, 1024, 3000, int(rng.integers(1, 3000))]))
        G = int(rng.integers(1, 7))
        cuts = np.sort(rng.integers(0, E + 1, G - 1)) if G > 1 else np.array([], dtype=np.int64)
        goff = [0] + [int(c) for c in cuts] + [E]
        scale = float(rng.choice([1.0, 0.7, 1e-10]))
        max_lag = F - 1 if rng.random() < 0.6 else int(rng.integers(0, F))
        r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-50, 50, (1, 3, E))
        ctx.set_option("lag_variant", 2)
        ctx.set_option("lag_batch_mb", int(rng.choice([-1, -1, 1])))
        ctx.set_option("lag_w1", 0)
        ref = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
        b0 = ctx.last_rel_bound()
        ctx.set_option("lag_w1", 1)
        got = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
        b1 = ctx.last_rel_bound()
        took += ctx.last_kernel_name() == "msd_power_w1_kernel"
        again = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
        assert np.array_equal(got, again), ("not reproducible", F, E, goff)
        nz = ref > 0
        err = float((np.abs(got[nz] - ref[nz]) / ref[nz]).max()) if nz.any() else 0.0
        assert err <= b0 + b1 + 1e-15, (F, E, goff, max_lag, err, b0, b1)
        assert (got[0] == 0.0).all() and (got[~nz] == 0.0).all()
        if t % 5 == 0 and E <= 1100:
            ctx.set_option("lag_variant", 1)
            exact = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
            nze = exact > 0
            err = float((np.abs(got[nze] - exact[nze]) / exact[nze]).max()) if nze.any() else 0.0
            assert err <= b1 + 1e-15, ("vs exact", F, E, goff, max_lag, err, b1)
        if (t + 1) % 100 == 0:
            print("trial %d ok (F %d E %d groups %s max_lag %d)" % (t + 1, F, E, goff, max_lag), flush=True)
finally:
    for k in ("lag_variant", "lag_batch_mb", "lag_w1"):
        ctx.set_option(k, -1)
print("soak_lag_short: %d shapes (%d through msd_power_w1_kernel) agree with the block-wide kernels within their bounds, every call reproducible" % (trials, took))
