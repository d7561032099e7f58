#!/usr/bin/env python
"""tools/per_frame_cost.py — what the drop-in functions call: `_rdf_loop` with PER-FRAME output (every frame normalised with
its own volume, rdf_cn.py:502-521) against the frame-summed call the bench's headline times, at C2's and C1's shapes:
kernel + pre-pass ms per 200 frames, wall per call, results identical (sum of the per-frame rows == the summed rows)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd import synth  # noqa: E402

ctx = B.default_context()
for name in ("C2", "C1"):
    cfg = synth.rdf_config(name)
    n, L, F = cfg["n_atoms"], cfg["box_len"], cfg["n_frames"]
    nb = int(cfg["r_cut"] / cfg["bin_size"])
    xyz = torch.from_numpy(synth.rdf_frames(n, range(F), L, cfg["seed_offset"])).cuda()
    ty = synth.rdf_types(n) if name == "C2" else synth.c1_types()
    rel = np.array(synth.ALL_PAIRS_4 if name == "C2" else synth.C1_RELATIONS, dtype=np.int32)
    box = np.full((F, 3), L)
    res = {}
    for per_frame in (False, True):
        ms, wall = [], []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = B.rdf_loop(xyz, ty, box, rel, cfg["r_cut"], cfg["bin_size"], nb, per_frame=per_frame, ctx=ctx)
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(ctx.last_kernel_ms()[0] + ctx.last_aux_ms())
        res[per_frame] = out
        print("%s per_frame=%-5s %-40s kernel + pre-pass %.3f ms, wall %.3f ms per call" % (
            name, per_frame, ctx.last_kernel_name(), float(np.median(ms[1:])), float(np.median(wall[1:]))), flush=True)
    same = np.array_equal(res[True][0].sum(axis=0), res[False][0]) and np.array_equal(res[True][1].sum(axis=0), res[False][1])
    print("   sum of per-frame rows == frame-summed rows:", same)
