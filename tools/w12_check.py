#!/usr/bin/env python
"""tools/w12_check.py — the round-5 full-lag kernel (padded length 12288, msd_fft_w12.h) against the round-3/4 kernel
(lag_fft_kernel 2, padded length 16384) and the exact-difference kernel, on shapes that take it: F + max_lag in
(8192, 12288]. Both sources (lag_direct 0: transposed copy, 2: in-kernel staging). Prints times and relative differences."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib  # noqa: E402
from mdproptools_amd import backend as B  # noqa: E402

ctx = _lib.Context(0)
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
shapes = [(5000, 4096, [0, 4096]), (4500, 6000, [0, 1000, 6000]), (6144, 2053, [0, 7, 2053]), (4097, 999, [0, 999]),
          (5001, 8192, [0, 100, 100, 8000, 8192])]
if not quick:
    shapes.append((5000, 50_000, [0, 50_000]))
for F, E, goff in shapes:
    g = torch.Generator(device="cuda")
    g.manual_seed(F + E)
    r = torch.cumsum(torch.randn((F, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
    r += torch.rand((1, 3, E), generator=g, device="cuda", dtype=torch.float64) * 80.0
    res = {}
    for name, opts in (("w12/copy", {"lag_fft_kernel": 3, "lag_direct": 0}), ("w12/staged", {"lag_fft_kernel": 3, "lag_direct": 2}),
                       ("lds3/staged", {"lag_fft_kernel": 2, "lag_direct": 2}), ("difference", {"lag_variant": 1})):
        if name == "difference" and E > 10_000:
            continue
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_option("lag_variant", opts.get("lag_variant", 2))
        try:
            out = B.lag_msd(r, F - 1, goff, scale=1.0, ctx=ctx)
            ms = []
            for _ in range(3):
                out = B.lag_msd(r, F - 1, goff, scale=1.0, ctx=ctx)
                ms.append(ctx.last_kernel_ms()[0])
            res[name] = out
            print("F %5d E %6d %-12s %-28s %8.3f ms  bound %.2e" % (F, E, name, ctx.last_kernel_name()[:28], min(ms),
                                                                  ctx.last_rel_bound()), flush=True)
        finally:
            for k in opts:
                ctx.set_option(k, -1 if k != "lag_fft_kernel" else 3)
            ctx.set_option("lag_variant", -1)
    ref = res.get("difference", res["lds3/staged"])
    for name in ("w12/copy", "w12/staged", "lds3/staged"):
        a, b = res[name][1:], ref[1:]
        m = b != 0
        print("   %-12s max rel diff vs %s: %.3e" % (name, "difference" if "difference" in res else "lds3",
                                                    float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m])))), flush=True)
print("fallbacks", ctx.fallbacks())
