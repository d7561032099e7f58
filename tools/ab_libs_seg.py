#!/usr/bin/env python
"""tools/ab_libs_seg.py LIB.so [LIB.so ...] [key=value ...] — mdhip_segment_com and mdhip_charge_flux at C4 shape (50k
atoms, 2500 x 16 + 2500 x 4 molecules, 5000 frames resident) through several BUILDS of libmdhip.so in one process
(boxes differ by ~10 %): kernel time min / median, fraction of the 8 TB/s spec, results compared with the first's."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib  # noqa: E402
from mdproptools_amd import backend as B  # noqa: E402

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
E, F = 50_000, 5000


def ctx_of(path):
    _lib._lib = None
    _lib.STRICT = False
    _lib.LIB_PATH = os.path.abspath(path)
    c = _lib.Context(0)
    for k, v in opts:
        c.set_option(k, int(v))
    return c


ctxs = [ctx_of(p) for p in libs]
r = torch.randn((F, 3, E), dtype=torch.float64, device="cuda")
off = np.concatenate([np.arange(0, 40_000, 16), np.arange(40_000, 50_001, 4)]).astype(np.int64)
mass = np.where(np.arange(E) < 40_000, 2.0, 3.0)
q = np.where(np.arange(E) % 2 == 0, 0.25, -0.25)
M = len(off) - 1
mol_type = (np.arange(M) >= 2500).astype(np.int32)
out = torch.empty((F, 3, M), dtype=torch.float64, device="cuda")
byt = (24.0 * E + 24.0 * M) * F
ref = [None, None]
for rnd in range(2):
    for p, ctx in zip(libs, ctxs):
        for which in (0, 1):
            ms = []
            for _ in range(6):
                if which == 0:
                    B.segment_com(r, mass, off, out=out, ctx=ctx)
                    res = out
                else:
                    res = B.charge_flux(r, mass, q, off, mol_type, 2, 1e5, 1.602e-19, ctx=ctx)
                    res = torch.as_tensor(res)
                ms.append(ctx.last_kernel_ms()[0])
            if ref[which] is None:
                ref[which] = res.clone()
            same = torch.equal(res, ref[which])
            ms = np.array(ms[2:])
            print("%-26s %-40s min %.4f ms  median %.4f ms  %.3f of 8 TB/s  %s" % (
                os.path.basename(p), ctx.last_kernel_name(), ms.min(), np.median(ms), byt / np.median(ms) / 1e-3 / 8e12,
                "same" if same else "DIFFERENT"), flush=True)
