import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
ctx = _lib.Context(0)
F, E = 5000, 8192
g = torch.Generator(device="cuda"); g.manual_seed(F + E)
r = torch.cumsum(torch.randn((F, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
poison = torch.full((F, 3, E), 1e100, dtype=torch.float64, device="cuda")
ctx.set_option("lag_variant", 2)
for kern in (3, 2):
    ctx.set_option("lag_fft_kernel", kern)
    ctx.set_option("lag_direct", 0)
    ref = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
    for trial in range(4):
        ctx.set_option("lag_direct", 0)
        B.lag_msd(poison, F - 1, [0, E], scale=1.0, ctx=ctx)   # WS_AUX1 (ring and transposed copy share it) = poison
        ctx.set_option("lag_direct", 2)
        out = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
        bad = ~np.isfinite(out) | (np.abs(out) > 1e50)
        rel = np.abs(out[1:] - ref[1:]) / ref[1:]
        print("kern", kern, "trial", trial, ctx.last_kernel_name(), "non-finite/huge entries:", int(bad.sum()), "max rel", float(np.nanmax(rel)), flush=True)
