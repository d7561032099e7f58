#!/usr/bin/env python
"""tools/h2d_pipe.py [n_calls] — C2 histogram calls on page-locked HOST frames, issued asynchronously (each before the one
before it is waited for). For a kernel + memory-copy trace (rocprofv3 --kernel-trace --memory-copy-trace): does the copy of
call k + 1 run under the sweep of call k?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd import synth  # noqa: E402
from mdproptools_amd._lib import default_context  # noqa: E402

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = synth.rdf_config("C2")
n, L, F = cfg["n_atoms"], cfg["box_len"], cfg["n_frames"]
nb = int(cfg["r_cut"] / cfg["bin_size"])
ctx = default_context(0)
x = synth.rdf_frames(n, range(F), L, cfg["seed_offset"])
pin = torch.empty(x.shape, dtype=torch.float64, pin_memory=True)
pin.numpy()[...] = x
xp = pin.numpy()
ty = synth.rdf_types(n)
rel = np.array(synth.ALL_PAIRS_4, dtype=np.int32)
box = np.full((F, 3), L)


def issue():
    return B.rdf_loop(xp, ty, box, rel, cfg["r_cut"], cfg["bin_size"], nb, per_frame=False, ctx=ctx, async_=True)


issue().wait()
issue().wait()
torch.cuda.synchronize()
t0 = time.perf_counter()
prev = None
stamps = []
for _ in range(n_calls):
    ta = time.perf_counter()
    h = issue()
    tb = time.perf_counter()
    if prev is not None:
        prev.wait()
    stamps.append((tb - ta, time.perf_counter() - tb))
    prev = h
prev.wait()
dt = (time.perf_counter() - t0) / n_calls
print("pipelined host-resident calls: %.3f ms per call" % (dt * 1e3))
print("issue ms:", " ".join("%.2f" % (a * 1e3) for a, _ in stamps))
print("wait  ms:", " ".join("%.2f" % (b * 1e3) for _, b in stamps))
