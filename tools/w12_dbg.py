import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
ctx = _lib.Context(0)
for F, E in ((5000, 4096), (6144, 4096), (5000, 8192)):
    g = torch.Generator(device="cuda"); g.manual_seed(F + E)
    r = torch.cumsum(torch.randn((F, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
    ctx.set_option("lag_variant", 2)
    outs = {}
    for name, src in (("copy", 0), ("st1", 2), ("st2", 2)):
        ctx.set_option("lag_direct", src)
        outs[name] = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
    a, b, c = outs["copy"], outs["st1"], outs["st2"]
    print(F, E, "staged deterministic:", np.array_equal(b, c))
    rel = np.abs(b[1:] - a[1:]) / a[1:]
    print("  max rel", rel.max(), "at lag/axis", np.unravel_index(np.argmax(rel), rel.shape))
    for ax in range(3):
        rr = rel[:, 0, ax]
        print("  axis", ax, "max", rr.max(), "lags with rel>1e-11:", int((rr > 1e-11).sum()), "first", np.nonzero(rr > 1e-11)[0][:8] + 1)
    # sign of difference
    d = (b[1:, 0, :3] - a[1:, 0, :3])
    print("  diff sign: mean", d.mean(axis=0), " sample lags 1,2,10,100,1000:", d[[0, 1, 9, 99, 999], 0])
