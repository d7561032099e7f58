#!/usr/bin/env python
"""tools/ab_segment.py — kernel time of mdhip_segment_com at C4 shape (50k atoms, 2500x16 + 2500x4 molecules,
5000 frames, 6 GB resident), 10 repetitions: min / median device time and GB/s (24 B per atom per frame read,
24 B per molecule written)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd._lib import default_context  # noqa: E402

E, F = 50_000, 5000
ctx = default_context(0)
r = torch.randn((F, 3, E), dtype=torch.float64, device="cuda")
off = np.concatenate([np.arange(0, 40_000, 16), np.arange(40_000, 50_001, 4)]).astype(np.int64)
mass = np.where(np.arange(E) < 40_000, 2.0, 3.0)
M = len(off) - 1
out = torch.empty((F, 3, M), dtype=torch.float64, device="cuda")
byt = (24.0 * E + 24.0 * M) * F
ref = None
for frame, cap, vec, gy in ((1, 0, 1, 0), (0, 0, 1, 0), (1, 0, 1, 0), (0, 256, 1, 0), (0, 512, 1, 0), (0, 0, 0, 0), (0, 0, 1, 128),
                            (0, 0, 1, 1250)):
    ctx.set_option("seg_frame", frame)
    ctx.set_option("seg_cap", cap)
    ctx.set_option("seg_vec", vec)
    ctx.set_option("seg_gy", gy)
    ms = []
    for _ in range(8):
        B.segment_com(r, mass, off, out=out, ctx=ctx)
        ms.append(ctx.last_kernel_ms()[0])
    ms = np.array(ms[2:])
    if ref is None:
        ref = out.clone()
    assert torch.equal(out, ref), "results differ"
    print("%-40s frame %d cap %4d vec %d gy %3d  min %.4f ms  median %.4f ms  -> %.0f GB/s (%.3f of 8 TB/s)" % (
        ctx.last_kernel_name(), frame, cap, vec, gy, ms.min(), np.median(ms), byt / np.median(ms) / 1e6, byt / np.median(ms) / 1e-3 / 8e12))
