"""Per-frame vs frame-summed RDF call on a C2/C3-shaped resident trajectory: wall, kernel, pre-pass.
python tools/pf_diag.py <n_atoms> <n_frames>"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdproptools_amd import backend as B
from mdproptools_amd._lib import default_context

n, F = int(sys.argv[1]), int(sys.argv[2])
ctx = default_context(0)
rng = np.random.default_rng(5)
L = 50.0 * (n / 1e4) ** (1 / 3)
xyz = torch.rand((F, 3, n), dtype=torch.float64, device="cuda") * L
ty = (1 + np.arange(n) % 4).astype(np.int32)
rel = np.array([[a, b] for a in range(1, 5) for b in range(a, 5)])
box = np.full((F, 3), L)
for per_frame, js, infl in ((False, 0, 1), (True, 0, 1), (True, 4, 1), (True, 8, 1), (True, 4, 2), (True, 2, 1)):
    ctx.set_option("rdf_jsplit", js)
    ctx.set_option("rdf_inflight", infl)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=per_frame, ctx=ctx)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("per_frame", per_frame, "jsplit", js, "inflight", infl, "wall ms %.3f" % (dt * 1e3), "kernel", ctx.last_kernel_name(), ctx.last_kernel_ms(),
          "prepass ms %.3f" % ctx.last_aux_ms(), flush=True)
