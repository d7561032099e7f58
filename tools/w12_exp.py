#!/usr/bin/env python
"""tools/w12_exp.py LIB.so ... — C4-shape full-lag call time through several BUILDS (timing experiments of msd_fft_w12.h:
-DW12_EXP=bits builds give wrong results and only say where the time goes). Both sources."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib  # noqa: E402
from mdproptools_amd import backend as B  # noqa: E402

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
F, E = 5000, 50_000
g = torch.Generator(device="cuda").manual_seed(1)
r = torch.cumsum(torch.randn((F, 3, E), dtype=torch.float64, device="cuda", generator=g) * 0.1, dim=0)
for p in libs:
    _lib._lib = None
    _lib.STRICT = False
    _lib.LIB_PATH = os.path.abspath(p)
    ctx = _lib.Context(0)
    ctx.set_option("lag_variant", 2)
    row = []
    for src in (0, 2):
        ctx.set_option("lag_direct", src)
        ms = []
        for _ in range(4):
            B.lag_msd(r, F - 1, [0, E], ctx=ctx)
            ms.append(ctx.last_kernel_ms()[0])
        row.append(min(ms[1:]))
    print("%-28s copy %.3f ms   staged %.3f ms   %s" % (os.path.basename(p), row[0], row[1], ctx.last_kernel_name()), flush=True)
    ctx.close()
