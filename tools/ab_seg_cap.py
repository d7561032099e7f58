"""segment_com at C4 size (50k atoms x 2000 frames) with seg_cap 1024 / 512 / 0 (= chosen by pick_seg_cap) in one process, for several molecule sizes
(uniform 3, 4, 10, 16, 40 atoms; the bench's mix of 16- and 4-atom molecules)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
ctx = _lib.Context(0)
E, F = 50_000, 2000
r = torch.from_numpy(np.cumsum(np.random.default_rng(0).normal(0, 0.1, (F, 3, E)), axis=0)).cuda()
mass = np.random.default_rng(1).uniform(1, 20, E)
cases = {"mix 16/4": np.concatenate([np.arange(0, 40_000, 16), np.arange(40_000, 50_001, 4)]).astype(np.int64)}
for m in (3, 4, 10, 16, 40):
    cases["uniform %d" % m] = np.arange(0, E - E % m + 1, m).astype(np.int64)
for name, seg in cases.items():
    out = torch.empty((F, 3, len(seg) - 1), dtype=torch.float64, device="cuda")
    row = []
    ref = None
    for cap in (1024, 512, 0, 1024, 512, 0):
        ctx.set_option("seg_cap", cap)
        best = 1e9
        for rep in range(5):
            B.segment_com(r, mass, seg, out=out, ctx=ctx)
            best = min(best, ctx.last_kernel_ms()[0])
        o = out.cpu().numpy()
        ref = o if ref is None else ref
        row.append("%d: %.3f ms%s" % (cap, best, "" if np.array_equal(o, ref) else " DIFF"))
    print("%-12s %s" % (name, "   ".join(row)), flush=True)
