"""Diagnostic for the packed-f32 sweep: python tools/pk_diag.py <rdf_pk value> <n_atoms> <n_frames> [per_frame] [n_types]
(more than 5 types: every 4th type pair as a relation, all types named -> class rows)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdproptools_amd import backend as B
from mdproptools_amd._lib import Context

pk, n, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
per_frame = len(sys.argv) < 5 or sys.argv[4] != "0"
rng = np.random.default_rng(5)
L = 50.0 * (n / 1e4) ** (1 / 3)
xyz = rng.uniform(0, L, (F, 3, n))
nt = int(sys.argv[5]) if len(sys.argv) > 5 else 4
ty = (1 + np.arange(n) % nt).astype(np.int32)
rel = np.array([[a, b] for a in range(1, nt + 1) for b in range(a, nt + 1)])
if nt > 5:
    rel = rel[::4]
box = np.full((F, 3), L)
res = {}
for v in (0, pk):
    ctx = Context(0)
    ctx.set_option("rdf_pk", v)
    for rep in range(3):
        res[v] = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=per_frame, ctx=ctx)
        print("rdf_pk", v, "rep", rep, "kernel", ctx.last_kernel_name(), "ms", ctx.last_kernel_ms(), "sum", int(res[v][0].sum()), flush=True)
same = all(np.array_equal(a, b) for a, b in zip(res[0][:2], res[pk][:2])) and res[0][2] == res[pk][2]
print("identical:", same)
if not same:
    d = res[pk][0].astype(np.int64) - res[0][0].astype(np.int64)
    print("full diff nonzero:", np.count_nonzero(d), "sum", d.sum(), "abs", np.abs(d).sum(), "ov", res[0][2], res[pk][2])
    sys.exit(1)
