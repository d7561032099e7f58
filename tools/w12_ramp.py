import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _lib.STRICT = False; _lib.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(0)
for F, E in ((6144, 1024), (6144, 2048), (6144, 4096), (6000, 1024), (4097, 2048), (5000, 1024)):
    t = torch.arange(F, dtype=torch.float64, device="cuda")[:, None, None]
    c = torch.arange(3 * E, dtype=torch.float64, device="cuda").reshape(1, 3, E)
    r = (16384.0 * c + t).contiguous()
    ctx.set_option("lag_variant", 2)
    k2 = np.arange(F, dtype=np.float64) ** 2
    for src in (0, 2):
        ctx.set_option("lag_direct", src)
        o = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
        d = np.abs(o[1:, 0, :3] - k2[1:, None])
        print("F %d E %d src %d %s max |msd - k^2| %.3g (axis maxima %s) fallbacks %d" % (F, E, src, ctx.last_kernel_name(), d.max(), d.max(axis=0), ctx.fallbacks()), flush=True)
