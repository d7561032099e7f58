import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
ctx = _lib.Context(0)
F, E = 5000, 8192
t = torch.arange(F, dtype=torch.float64, device="cuda")[:, None, None]
c = torch.arange(3 * E, dtype=torch.float64, device="cuda").reshape(1, 3, E)
for name, r in (("x = 1000 c + t", (1000.0 * c + t).contiguous()), ("x = t", (0.0 * c + t).contiguous()),
                ("x = c", (c + 0.0 * t).contiguous())):
    ctx.set_option("lag_variant", 2)
    ctx.set_option("lag_direct", 0)
    ref = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
    ctx.set_option("lag_direct", 2)
    for trial in range(3):
        out = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
        d = out[:, 0, :3] - ref[:, 0, :3]
        k = np.arange(F)
        print(name, "trial", trial, "max abs diff", float(np.abs(d).max()), "at lag", int(np.argmax(np.abs(d).max(axis=1))),
              " ref[10]", ref[10, 0, :3], " exact k^2 dev (copy)", float(np.abs(ref[1:, 0, 0] - k[1:] ** 2.0).max()), flush=True)
