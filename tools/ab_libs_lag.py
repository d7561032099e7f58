#!/usr/bin/env python
"""tools/ab_libs_lag.py LIB.so [LIB.so ...] [key=value ...] — the full lag x origin MSD at BASELINE C4 shape (50 000
entities x 5000 frames, default path) through several BUILDS of libmdhip.so in one process (boxes differ by ~10 %):
kernel time of the call, results compared with the first library's."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib  # noqa: E402
from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd import synth  # noqa: E402

libs = [a.split(":")[0] for a in sys.argv[1:] if a.split(":")[0].endswith(".so")]
lib_opts = [dict(kv.split("=") for kv in a.split(":")[1].split(",")) if ":" in a else {} for a in sys.argv[1:]
            if a.split(":")[0].endswith(".so")]  # LIB.so:key=value,key=value
opts = [a.split("=") for a in sys.argv[1:] if "=" in a and ".so" not in a]
E, F = int(os.environ.get("LAG_E", 50_000)), int(os.environ.get("LAG_F", 5000))  # (other shapes: LAG_E=30000 LAG_F=8000)


def ctx_of(path):
    _lib._lib = None
    _lib.STRICT = False
    _lib.LIB_PATH = os.path.abspath(path)
    c = _lib.Context(0)
    for k, v in opts:
        c.set_option(k, int(v))
    return c


ctxs = [ctx_of(p) for p in libs]
for c_, o_ in zip(ctxs, lib_opts):
    for k_, v_ in o_.items():
        c_.set_option(k_, int(v_))
g = torch.Generator(device="cuda")
g.manual_seed(synth.BASE_SEED + 4)
r = torch.empty((F, 3, E), dtype=torch.float64, device="cuda")
r[0] = torch.rand((3, E), generator=g, device="cuda", dtype=torch.float64) * 82.8
for f0 in range(1, F, 250):
    f1 = min(F, f0 + 250)
    r[f0:f1] = r[f0 - 1] + torch.cumsum(torch.randn((f1 - f0, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
ref = None
for rnd in range(2):
    for k_lib, (p, ctx) in enumerate(zip(libs, ctxs)):
        ms = []
        for _ in range(4):
            out = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
            ms.append(ctx.last_kernel_ms()[0])
        if ref is None:
            ref = out
        err = float(np.max(np.abs(out[1:] - ref[1:]) / ref[1:]))
        print("%-34s %-22s min %.3f ms  median %.3f ms  max rel diff vs first %.2e  bound %.2e" % (
            os.path.basename(p) + " " + ",".join("%s=%s" % kv for kv in lib_opts[k_lib].items()), ctx.last_kernel_name(), min(ms[1:]), float(np.median(ms[1:])), err, ctx.last_rel_bound()),
            flush=True)
