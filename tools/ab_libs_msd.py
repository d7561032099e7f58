#!/usr/bin/env python
"""tools/ab_libs_msd.py LIB.so [LIB.so ...] — single-origin MSD, fixed-lag windows and molecule COM at C4 shape through
several BUILDS of libmdhip.so in one process: kernel time, GB/s of algorithmic bytes, results compared."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib  # noqa: E402
from mdproptools_amd import backend as B  # noqa: E402

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
E, F = 50_000, 2000


def ctx_of(path):
    _lib._lib = None
    _lib.STRICT = False
    _lib.LIB_PATH = os.path.abspath(path)
    return _lib.Context(0)


ctxs = [ctx_of(p) for p in libs]
r = torch.from_numpy(np.cumsum(np.random.default_rng(0).normal(0, 0.1, (F, 3, E)), axis=0)).cuda()
pairs = np.column_stack([np.zeros(F), np.arange(F)]).astype(np.int32)
mass = np.random.default_rng(1).uniform(1, 20, E)
seg = np.arange(0, E + 1, 10).astype(np.int64)
out = torch.empty((F, 3, len(seg) - 1), dtype=torch.float64, device="cuda")
ref = {}
for rnd in range(2):
    for p, ctx in zip(libs, ctxs):
        res = {}
        for name, fn, nbytes in (("msd_pairs", lambda: B.msd_pairs(r, pairs, [0, E], ctx=ctx), 24.0 * E * F),
                                 ("msd_windows", lambda: B.msd_windows(r, 4, ctx=ctx), 24.0 * E * (F // 4)),
                                 ("segment_com", lambda: B.segment_com(r, mass, seg, out=out, ctx=ctx)[0], 24.0 * E * F * 1.1)):
            best = 1e9
            for rep in range(5):
                o = fn()
                best = min(best, ctx.last_kernel_ms()[0])
            o = o.cpu().numpy() if hasattr(o, "cpu") else o
            same = np.array_equal(o, ref.setdefault(name, o))
            res[name] = "%s %.3f ms %.0f GB/s%s" % (name, best, nbytes / best * 1e-6, "" if same else " DIFFERENT")
        print("%-26s %s" % (os.path.basename(p), "   ".join(res.values())), flush=True)
