#!/usr/bin/env python
"""tests/bench/soak_lag_ends.py [trials] [seed] — the default full-lag MSD path (lag_variant 3) on data that sits around its
1e-10 bound: random walks of 300 .. 26 000 frames riding on a slow oscillation of random amplitude, 1 .. 60 entities, one to
four groups (empty and one-entity ones), random max_lag and scale. Whatever the library decides — the spectral result stands,
a few lags at the ends of the range are recomputed from the difference form, or the whole call goes to the difference kernel —
the result must lie within the bound it reports of the exact-difference kernel (within 1e-12 where it reports 0), the rows it
left alone must be the spectral path's bit for bit, and a second call must reproduce the first."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdproptools_amd import backend as B  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = B.default_context()
took = {"spectral": 0, "ends": 0, "whole": 0}
try:
    for t in range(trials):
        F = int(rng.choice([300, 1500, 3000, 5000, 9000, 13000, 26000, int(rng.integers(300, 20000))]))
        E = int(rng.integers(1, 61))
        G = int(rng.integers(1, 5))
        cuts = np.sort(rng.integers(0, E + 1, G - 1)) if G > 1 else np.array([], dtype=np.int64)
        goff = [0] + [int(c) for c in cuts] + [E]
        max_lag = F - 1 if rng.random() < 0.7 else int(rng.integers(1, F))
        scale = float(rng.choice([1.0, 0.5]))
        amp = float(rng.choice([0.0, 3.0, 9.0, 17.0, 40.0]))
        tt = np.arange(F)[:, None, None]
        r = (np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + amp * np.sin(2 * np.pi * tt / F + rng.uniform(0, 6.28, (1, 3, E)))
             + rng.uniform(-50, 50, (1, 3, E)))
        ctx.set_option("lag_variant", 1)
        exact = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
        ctx.set_option("lag_variant", 2)
        spec = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
        ctx.set_option("lag_variant", 3)
        got = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
        bound, name = ctx.last_rel_bound(), ctx.last_kernel_name()
        again = B.lag_msd(r, max_lag, goff, scale=scale, ctx=ctx)
        assert np.array_equal(got, again), ("not reproducible", F, E, goff)
        nz = exact > 0
        err = float((np.abs(got[nz] - exact[nz]) / exact[nz]).max()) if nz.any() else 0.0
        assert bound <= 1e-10 and err <= max(bound, 1e-12), (F, E, goff, max_lag, amp, name, err, bound)
        if "lag_low_lags" in name:
            took["ends"] += 1
            changed = np.where(np.any(got != spec, axis=(1, 2)))[0]
            assert all(k <= 24 or k >= max_lag + 1 - 24 for k in changed), (F, max_lag, changed)
        elif name.startswith("lag_msd_"):
            took["whole"] += 1
        else:
            took["spectral"] += 1
            assert np.array_equal(got, spec)
        if (t + 1) % 20 == 0:
            print("trial %d ok (F %d E %d groups %s max_lag %d amp %g: %s)" % (t + 1, F, E, goff, max_lag, amp, name[:40]), flush=True)
finally:
    ctx.set_option("lag_variant", -1)
print("decisions:", took)
print("soak_lag_ends: %d shapes within the reported bounds of the difference kernel, every call reproducible" % trials)
