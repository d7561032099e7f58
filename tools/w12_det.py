import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
F, E = 5000, 8192
g = torch.Generator(device="cuda"); g.manual_seed(F + E)
r = torch.cumsum(torch.randn((F, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
for p in libs:
    _lib._lib = None; _lib.STRICT = False; _lib.LIB_PATH = os.path.abspath(p)
    ctx = _lib.Context(0)
    ctx.set_option("lag_variant", 2)
    for kern in (3, 2):
        ctx.set_option("lag_fft_kernel", kern)
        ctx.set_option("lag_direct", 0)
        ref = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
        ctx.set_option("lag_direct", 2)
        outs = [B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx) for _ in range(6)]
        same = sum(np.array_equal(o, outs[0]) for o in outs)
        rel = max(float(np.max(np.abs(o[1:] - ref[1:]) / ref[1:])) for o in outs)
        print("%-24s kern %d %-22s identical runs %d/6  max rel vs copy %.2e" % (os.path.basename(p), kern, ctx.last_kernel_name(), same, rel), flush=True)
    ctx.close()
