"""Host time of ISSUING one asynchronous full-lag MSD call at C4 shape (the GPU idle, nothing to wait for)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
ctx = B.default_context()
F, E = 5000, 50000
g = torch.Generator(device="cuda").manual_seed(1)
r = torch.cumsum(torch.randn((F, 3, E), dtype=torch.float64, device="cuda", generator=g) * 0.1, dim=0)
out = torch.empty((F, 1, 4), dtype=torch.float64, device="cuda")
for kind in ("host result", "device result"):
    ts = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h = B.lag_msd(r, F - 1, [0, E], ctx=ctx, async_=True, out=out if kind == "device result" else None)
        t1 = time.perf_counter()
        h.wait()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    print(kind, "issue ms", ["%.3f" % a for a, _ in ts[2:]], "wait ms", ["%.2f" % b for _, b in ts[2:6]])
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(5):
    h = B.lag_msd(r, F - 1, [0, E], ctx=ctx, async_=True, out=out)
    h.wait()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
