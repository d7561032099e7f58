#!/usr/bin/env python
"""tests/bench/soak_lag_long.py [trials] [seed] — random shapes through the full-lag MSD paths for trajectories beyond the
fused kernels (F + max_lag > 16 384, frames 8193 .. 26 000): the residue-class kernels (`lag_residue` 1: padded length
24 576 or 49 152; 2: the three-class first form up to 12 288 frames) against the batched power-of-two transforms
(`lag_residue` 0), entities 1 .. 400, one to five groups with empty and one-entity ones, random scale and max_lag, batches of
1-8 MB so that batches, blocks and segments straddle. Every result must agree with the batched path within the sum of the
two reported bounds, a sample of lags with the exact-difference kernel within the bound, and a second call must reproduce
the first bit for bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdproptools_amd import backend as B  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = B.default_context()
names = {}
try:
    for t in range(trials):
        F = int(rng.choice([8193, 12288, 12289, 24576, int(rng.integers(8193, 26001)), int(rng.integers(8193, 12289))]))
        E = int(rng.choice([1, 2, 5, 16, 17, 85, 300, int(rng.integers(1, 400))]))
        if F > 16000:
            E = min(E, 90)
        G = int(rng.integers(1, 6))
        cuts = np.sort(rng.integers(0, E + 1, G - 1)) if G > 1 else np.array([], dtype=np.int64)
        goff = [0] + [int(c) for c in cuts] + [E]
        scale = float(rng.choice([1.0, 0.7, 1e-10]))
        max_lag = F - 1 if rng.random() < 0.6 else int(rng.integers(max(1, 16385 - F), F))
        r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-50, 50, (1, 3, E))
        ctx.set_option("lag_variant", 2)
        ctx.set_option("lag_batch_mb", int(rng.choice([-1, 1, 2, 8])))
        ctx.set_option("lag_residue", 0)
        ref = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
        b0 = ctx.last_rel_bound()
        nz = ref > 0
        for mode in (1, 2):
            ctx.set_option("lag_residue", mode)
            got = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
            b1 = ctx.last_rel_bound()
            names[ctx.last_kernel_name()] = names.get(ctx.last_kernel_name(), 0) + 1
            again = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
            assert np.array_equal(got, again), ("not reproducible", mode, F, E, goff)
            err = float((np.abs(got[nz] - ref[nz]) / ref[nz]).max()) if nz.any() else 0.0
            assert err <= b0 + b1 + 1e-15, (mode, F, E, goff, max_lag, err, b0, b1)
            assert (got[~nz] == 0.0).all()
        if t % 5 == 0 and E <= 100:  # (the exact kernel is O(F^2) per entity)
            ctx.set_option("lag_variant", 1)
            exact = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
            nze = exact > 0
            err = float((np.abs(got[nze] - exact[nze]) / exact[nze]).max()) if nze.any() else 0.0
            assert err <= b1 + 1e-15, ("vs exact", F, E, goff, max_lag, err, b1)
        if (t + 1) % 10 == 0:
            print("trial %d ok (F %d E %d groups %s max_lag %d)" % (t + 1, F, E, goff, max_lag), flush=True)
finally:
    for k in ("lag_variant", "lag_batch_mb", "lag_residue"):
        ctx.set_option(k, -1)
print("kernels taken:", names)
print("soak_lag_long: %d shapes, the residue-class paths agree with the batched transforms within their bounds, every call reproducible" % trials)
