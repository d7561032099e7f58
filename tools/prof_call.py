#!/usr/bin/env python
"""tools/prof_call.py — cProfile of 200 C2 rdf_loop calls (device-resident frames): where the host side of a step goes."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd import synth  # noqa: E402
from mdproptools_amd._lib import default_context  # noqa: E402

ctx = default_context(0)
cfg = synth.rdf_config("C2")
n, L, F = cfg["n_atoms"], cfg["box_len"], cfg["n_frames"]
xd = torch.from_numpy(synth.rdf_frames(n, range(F), L, cfg["seed_offset"])).cuda()
ty = synth.rdf_types(n)
rel = np.array(synth.ALL_PAIRS_4, dtype=np.int32)
box = np.full((F, 3), L)


def step():
    return B.rdf_loop(xd, ty, box, rel, 20.0, 0.05, 400, per_frame=False, ctx=ctx)


for _ in range(5):
    step()
t0 = time.perf_counter()
for _ in range(50):
    step()
wall = (time.perf_counter() - t0) / 50
print("step %.3f ms, kernels %.3f + pre-pass %.3f ms" % (wall * 1e3, ctx.last_kernel_ms()[0], ctx.last_aux_ms()))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(12)
