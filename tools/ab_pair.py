#!/usr/bin/env python
"""tools/ab_pair.py KEY=V1,V2,... [C2|C3] [rdf|cn] — A/B of one library OPTION on the pair kernel inside ONE process (boxes differ
by ~10 %, runs inside a process by ~1 %): kernel time (min / median of 8 launches) per value, results asserted equal.
For two code variants use tools/ab_libs.py with two BUILDS: a kernel that carries both variants behind a run-time flag
is not either of them (the two-slot pair block looked 5 % faster that way and was 2 % slower build against build)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd import synth  # noqa: E402
from mdproptools_amd._lib import default_context  # noqa: E402

key, vals = sys.argv[1].split("=")
vals = [int(v) for v in vals.split(",")]
which = sys.argv[2] if len(sys.argv) > 2 else "C2"
op = sys.argv[3] if len(sys.argv) > 3 else "rdf"  # rdf | cn
cfg = synth.rdf_config(which)
n, L = cfg["n_atoms"], cfg["box_len"]
F = cfg["n_frames"] if which == "C2" else 64
xyz = torch.from_numpy(synth.rdf_frames(n, range(F), L, cfg["seed_offset"])).cuda()
ty = synth.rdf_types(n)
rel = np.array(synth.ALL_PAIRS_4)
box = np.full((F, 3), L)
ctx = default_context(0)
ref = None
for rnd in range(2):
    for v in vals:
        ctx.set_option(key, v)
        ms = []
        for _ in range(8):
            if op == "cn":
                out = (B.cn_loop(xyz, ty, box, rel, synth.cn_cutoffs(len(rel)), per_frame=False, ctx=ctx),) * 2
            else:
                out = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=False, ctx=ctx)
            ms.append(ctx.last_kernel_ms()[0])
        if ref is None:
            ref = out
        assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
        ms = np.array(ms[2:])
        print("%s %s=%d  %s  min %.4f ms  median %.4f ms" % (which, key, v, ctx.last_kernel_name(), ms.min(), np.median(ms)))
