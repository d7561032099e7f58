#!/usr/bin/env python
"""tools/gk_diag.py — where the wall time of backend.green_kubo / cumtrapz goes (C5 size, resident series)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib, backend as B, synth  # noqa: E402

ctx = _lib.default_context(0)
n = 1_000_000
p = torch.from_numpy(synth.ar1_series(n)).cuda()
torch.cuda.synchronize()


def t(fn, reps=6):
    out = []
    keep = None
    for _ in range(reps):
        t0 = time.perf_counter()
        keep = fn()
        out.append((time.perf_counter() - t0) * 1e3)
    return " ".join("%.2f" % v for v in out), keep


print("result_array 24MB      ", t(lambda: _lib.result_array((3, n), device=0))[0])
print("xcorr fft              ", t(lambda: B.xcorr(p, ctx=ctx))[0], "kernel", ctx.last_kernel_ms())
a = B.xcorr(p, ctx=ctx)
ad = torch.from_numpy(np.ascontiguousarray(a)).cuda()
print("cumtrapz host in       ", t(lambda: B.cumtrapz(a, 1e-15, ctx=ctx))[0], "kernel", ctx.last_kernel_ms())
print("cumtrapz dev in        ", t(lambda: B.cumtrapz(ad, 1e-15, ctx=ctx))[0], "kernel", ctx.last_kernel_ms())
od = torch.empty((3, n - 1), dtype=torch.float64, device="cuda")
print("cumtrapz dev in/out    ", t(lambda: B.cumtrapz(ad, 1e-15, ctx=ctx, out=od))[0], "kernel", ctx.last_kernel_ms())
print("green_kubo all         ", t(lambda: B.green_kubo(p, acf_scale=2.0, dx=1e-15, integral_scale=3.0, want_mean=True, ctx=ctx), 8)[0],
      "kernel", ctx.last_kernel_ms())
print("green_kubo integral    ", t(lambda: B.green_kubo(p, acf_scale=2.0, dx=1e-15, integral_scale=3.0, want_acf=False, ctx=ctx), 8)[0],
      "kernel", ctx.last_kernel_ms())
print("pool", {k: len(v) for k, v in _lib.PINNED.free.items()}, _lib.PINNED.live)
