"""Probe of the staged full-lag kernel at F = 6144 (one segment per axis): inputs const-per-column / ramp in t / both."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _lib.STRICT = False; _lib.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = _lib.Context(0)
F, E = 6144, 1024
t = torch.arange(F, dtype=torch.float64, device="cuda")[:, None, None]
c = torch.arange(3 * E, dtype=torch.float64, device="cuda").reshape(1, 3, E)
ctx.set_option("lag_variant", 2)
np.set_printoptions(linewidth=200)
one = torch.zeros((F, 3, E), dtype=torch.float64, device="cuda")
def impulse(t0, col):
    r = torch.zeros((F, 3 * E), dtype=torch.float64, device="cuda")
    r[t0, col] = 1.0
    return r.reshape(F, 3, E).contiguous()
cases = [("const", (16384.0 * c + 0.0 * t).contiguous()), ("const1", (1.0 + 0.0 * c + 0.0 * t).contiguous()),
         ("t", (0.0 * c + t).contiguous()), ("both", (16384.0 * c + t).contiguous()),
         ("both1024", (1024.0 * c + t).contiguous())]
for name, r in cases:
    outs = {}
    for src in (0, 2):
        ctx.set_option("lag_direct", src)
        outs[src] = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)   # [F][1][4]
    d = outs[2][:, 0, :3] - outs[0][:, 0, :3]
    print(name, ctx.last_kernel_name(), "fallbacks", ctx.fallbacks(), "max|d|", np.abs(d).max(axis=0), flush=True)
    for k in (1, 2, 3, 4, 8, 16, 100, 1000, 3072, 6000, 6143):
        print("   lag", k, "staged", outs[2][k, 0, :3], "copy", outs[0][k, 0, :3])
