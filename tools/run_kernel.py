#!/usr/bin/env python
"""
tools/run_kernel.py <what> — runs one library call a few times (for rocprofv3 passes over a single kernel).
  lag [F] [E]   full-lag MSD           xcorr [n]   direct ACF           com   segment COM (C4 shape, 500 frames)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from mdproptools_amd import backend as B
    from mdproptools_amd import synth
    from mdproptools_amd._lib import default_context

    ctx = default_context(0)
    what = sys.argv[1]
    if what == "lag":
        F = int(sys.argv[2]) if len(sys.argv) > 2 else 2100
        E = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
        r = torch.from_numpy(synth.random_walk(E, F)).cuda()
        for _ in range(3):
            B.lag_msd(r, F - 1, [0, E])
            print(ctx.last_kernel_name(), ctx.last_kernel_ms()[0])
    elif what == "c2":
        cfg = synth.rdf_config("C2")
        n, L, F = cfg["n_atoms"], cfg["box_len"], cfg["n_frames"]
        xyz = torch.from_numpy(synth.rdf_frames(n, range(F), L, cfg["seed_offset"])).cuda()
        ty = synth.rdf_types(n)
        rel = np.array(synth.ALL_PAIRS_4)
        box = np.full((F, 3), L)
        for _ in range(4):
            B.rdf_loop(xyz, ty, box, rel, cfg["r_cut"], cfg["bin_size"], int(cfg["r_cut"] / cfg["bin_size"]), per_frame=False, ctx=ctx)
            print(ctx.last_kernel_name(), ctx.last_kernel_ms()[0], ctx.last_aux_ms())
    elif what == "xcorr":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
        p = torch.from_numpy(synth.ar1_series(n)).cuda()
        for _ in range(3):
            B.xcorr(p, method=B.XCORR_DIRECT)
            print(ctx.last_kernel_name(), ctx.last_kernel_ms()[0])
    elif what == "com":
        E, F = 50_000, 500
        r = torch.from_numpy(synth.random_walk(E, F)).cuda()
        off = np.concatenate([np.arange(0, 40_000, 16), np.arange(40_000, 50_001, 4)]).astype(np.int64)
        mass = np.where(np.arange(E) < 40_000, 2.0, 3.0)
        out = torch.empty((F, 3, len(off) - 1), dtype=torch.float64, device="cuda")
        for _ in range(3):
            B.segment_com(r, mass, off, out=out)
            print(ctx.last_kernel_name(), ctx.last_kernel_ms()[0])


if __name__ == "__main__":
    main()
