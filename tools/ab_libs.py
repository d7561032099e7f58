#!/usr/bin/env python
"""tools/ab_libs.py LIB_A.so LIB_B.so [C2|C3] [rdf|cn|rdf_cn] — A/B of two BUILDS of libmdhip.so inside one process (boxes
differ by ~10 %): the C2 (or 64 frames of C3) call alternately through each library, kernel time min / median, results
equal."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib  # noqa: E402
from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd import synth  # noqa: E402


def ctx_of(path):
    _lib._lib = None
    _lib.STRICT = False
    _lib.LIB_PATH = os.path.abspath(path)
    return _lib.Context(0)


libs = sys.argv[1:3]
which = sys.argv[3] if len(sys.argv) > 3 else "C2"
op = sys.argv[4] if len(sys.argv) > 4 else "rdf"
ctxs = [ctx_of(p) for p in libs]
cfg = synth.rdf_config(which)
n, L = cfg["n_atoms"], cfg["box_len"]
F = cfg["n_frames"] if which == "C2" else 64
xyz = torch.from_numpy(synth.rdf_frames(n, range(F), L, cfg["seed_offset"])).cuda()
ty = synth.rdf_types(n)
rel = np.array(synth.ALL_PAIRS_4)
box = np.full((F, 3), L)
ref = None
for rnd in range(3):
    for p, ctx in zip(libs, ctxs):
        ms, aux = [], []
        for _ in range(8):
            if op == "rdf":
                out = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=False, ctx=ctx)
            elif op == "cn":
                out = (B.cn_loop(xyz, ty, box, rel, synth.cn_cutoffs(len(rel)), per_frame=False, ctx=ctx),) * 2
            else:
                o = B.rdf_cn_loop(xyz, ty, box, rel, 20.0, 0.05, 400, synth.cn_cutoffs(len(rel)), per_frame=False, ctx=ctx)
                out = (o[0], o[3])
            ms.append(ctx.last_kernel_ms()[0])
            aux.append(ctx.last_aux_ms())
        if ref is None:
            ref = out
        same = np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
        assert same or os.environ.get("AB_ALLOW_DIFFERENT"), "results differ"
        ms = np.array(ms[2:])
        print("%s %s %-28s min %.4f ms  median %.4f ms  pre-pass %.4f ms%s" % (which, op, os.path.basename(p), ms.min(), np.median(ms), float(np.median(aux[2:])), "" if same else "  DIFFERENT RESULTS (timing experiment)"))
