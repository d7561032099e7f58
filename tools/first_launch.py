"""Where the first call's wall time goes: imports, context, first launch of each kernel family."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
t0 = time.time()
import numpy as np
import torch
t1 = time.time(); print("import torch        %.2f s" % (t1 - t0))
from mdproptools_amd import _lib, backend
t2 = time.time(); print("import package      %.2f s" % (t2 - t1))
ctx = _lib.Context()
t3 = time.time(); print("context create      %.2f s" % (t3 - t2))
rng = np.random.default_rng(0)
n = 2000
xyz = np.ascontiguousarray((rng.random((1, n, 3)) * 20).transpose(0, 2, 1))
types = rng.integers(1, 3, n)
box = np.array([[20.0, 20.0, 20.0]])
rel = np.array([[1, 1], [1, 2]])
for rep in range(3):
    t = time.time()
    backend.rdf_loop(xyz, types, box, rel, 10.0, 0.1, 100, ctx=ctx)
    print("rdf call %d          %.3f s  kernel ms %s" % (rep, time.time() - t, ctx.last_kernel_ms()))
r = rng.random((64, 3, 500))
for rep in range(2):
    t = time.time()
    backend.msd_windows(r, 8, ctx=ctx)
    print("msd_windows call %d  %.3f s" % (rep, time.time() - t))
a = rng.random((3, 4096))
for rep in range(2):
    t = time.time()
    backend.xcorr(a, ctx=ctx)
    print("xcorr(fft) call %d   %.3f s" % (rep, time.time() - t))
t = time.time(); torch.zeros(4, device="cuda").sum().item()
print("torch first kernel  %.3f s" % (time.time() - t))
