import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
_lib.STRICT = False
_lib.LIB_PATH = os.path.abspath("tools/_bin/libmdhip_w12verify.so")
ctx = _lib.Context(0)
F, E = 5000, 8192
t = torch.arange(F, dtype=torch.float64, device="cuda")[:, None, None]
c = torch.arange(3 * E, dtype=torch.float64, device="cuda").reshape(1, 3, E)
r = (16384.0 * c + t).contiguous()
ctx.set_option("lag_variant", 2)
ctx.set_option("lag_direct", 2)
for trial in range(2):
    B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
    torch.cuda.synchronize()
    print("trial", trial, "done", ctx.last_kernel_name(), flush=True)
import numpy as np
ctx.set_option("lag_direct", 0)
ref = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
ctx.set_option("lag_direct", 2)
for trial in range(3):
    out = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
    d = np.abs(out[:, 0, :3] - ref[:, 0, :3])
    print("verify build: staged vs copy max abs diff", float(d.max()), "at lag", int(np.argmax(d.max(axis=1))), flush=True)
