#!/usr/bin/env python
"""tools/call_overhead.py — wall time of one library call against the kernel time it reports, for the three MSD calls
of the C4 workload on device-resident input (50 000 entities x 5000 frames): what the host side of a call costs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd._lib import default_context  # noqa: E402

E, F = 50_000, 5000
ctx = default_context(0)
g = torch.Generator(device="cuda")
g.manual_seed(7)
r = torch.empty((F, 3, E), dtype=torch.float64, device="cuda")
r[0] = torch.rand((3, E), generator=g, device="cuda", dtype=torch.float64) * 80
for f0 in range(1, F, 250):
    f1 = min(F, f0 + 250)
    r[f0:f1] = r[f0 - 1] + torch.cumsum(torch.randn((f1 - f0, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
goff = [0, E]
pairs = np.stack([np.zeros(F, dtype=np.int64), np.arange(F, dtype=np.int64)], axis=1)
out_dev = {"pairs": torch.empty((F, 1, 4), dtype=torch.float64, device="cuda"),
           "lag": torch.empty((F, 1, 4), dtype=torch.float64, device="cuda")}
calls = {
    "msd_pairs (host result)": lambda: B.msd_pairs(r, pairs, goff, scale=1.0, ctx=ctx),
    "msd_pairs (device result)": lambda: B.msd_pairs(r, pairs, goff, scale=1.0, ctx=ctx, out=out_dev["pairs"]),
    "msd_windows tao 4": lambda: B.msd_windows(r, 4, scale=1.0, ctx=ctx),
    "lag_msd (host result)": lambda: B.lag_msd(r, F - 1, goff, scale=1.0, ctx=ctx),
    "lag_msd (device result)": lambda: B.lag_msd(r, F - 1, goff, scale=1.0, ctx=ctx, out=out_dev["lag"]),
}
for name, fn in calls.items():
    wall, kern = [], []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        wall.append(time.perf_counter() - t0)
        kern.append(ctx.last_kernel_ms()[0] + ctx.last_aux_ms())
    w, k = np.median(wall[2:]) * 1e3, np.median(kern[2:])
    print("%-28s wall %.3f ms  kernels %.3f ms  host side %.3f ms" % (name, w, k, w - k), flush=True)
