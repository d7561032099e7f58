#!/usr/bin/env python
"""tools/lag_sizes.py [E] — full-lag MSD (max_lag = F - 1, one group) over trajectory lengths on both sides of every kernel
boundary of the spectral path: kernel time per call and per unit of transform work E * F * log2(L) (L = the padded length
the path transforms at), relative to the C4 point (F = 5000). VERDICT r05 item 4: no length more than 1.5x the C4 point per
unit. Random walks generated on the device (0.1 per frame), spectral path forced (lag_variant 2), results within the
reported bound of 1e-10 checked against the status word only (parity per size: tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdproptools_amd import backend as B  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
ctx = B.default_context()
ctx.set_option("lag_variant", 2)
if len(sys.argv) > 2:  # A/B: from how many frames on the 12288-point kernel takes series of F + max_lag <= 8192 (0: never)
    ctx.set_option("lag_w12_min_f", int(sys.argv[2]))
    print("lag_w12_min_f =", sys.argv[2])
rows = []
for F in (300, 512, 700, 1000, 1400, 1535, 1536, 1800, 2048, 2500, 3000, 4096, 5000, 6144, 8192, 8193, 10_000, 12_288, 12_289, 20_000, 24_576):
    g = torch.Generator(device="cuda").manual_seed(100 + F)
    r = torch.empty((F, 3, E), dtype=torch.float64, device="cuda")
    for f0 in range(0, F, 500):
        r[f0:f0 + 500] = torch.randn((min(500, F - f0), 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1
    torch.cumsum(r, dim=0, out=r)
    out = torch.empty((F, 1, 4), dtype=torch.float64, device="cuda")
    ms = []
    for rep in range(4):
        B.lag_msd(r, F - 1, [0, E], out=out, ctx=ctx)
        ms.append(ctx.last_kernel_ms()[0])
    kernel = ctx.last_kernel_name()
    bound = ctx.last_rel_bound()
    L = (1024 * ((2 * F - 1 + 1023) // 1024) if kernel.startswith("msd_power_w1_") else 49152 if "msd_power_w12o" in kernel else 24576 if kernel.startswith(("msd_power_w12p", "msd_power_w12r")) else 12288 if kernel.startswith("msd_power_w12")
         else 1 << int(np.ceil(np.log2(2 * F - 1))))
    t = float(np.median(ms[1:]))
    rows.append((F, L, kernel, t, t / (E * F * np.log2(L)) * 1e9, bound, float(out[1, 0, 3].item())))
    del r, out
    torch.cuda.empty_cache()
ref = [x for x in rows if x[0] == 5000][0][4]
print("full-lag MSD, E = %d entities, one group, max_lag = F - 1 (kernel ms: the library's own events, median of 3)" % E)
print("%7s %7s %-24s %10s %16s %8s %10s" % ("F", "L", "kernel", "ms", "ps/(E F log2 L)", "vs C4", "bound"))
for F, L, kernel, t, unit, bound, _ in rows:
    print("%7d %7d %-24s %10.3f %16.4f %8.2f %10.1e" % (F, L, kernel.replace("msd_power_", "").replace("_kernel", "")[:24], t, unit * 1e3, unit / ref, bound))
worst = max(x[4] for x in rows) / ref
print("worst per-unit cost: %.2f x the C4 point (bar 1.5)" % worst)
