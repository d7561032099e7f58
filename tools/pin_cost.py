#!/usr/bin/env python
"""tools/pin_cost.py — what a page-locked result array costs: hipHostMalloc + first touch, against the copy it speeds up."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib  # noqa: E402

lib = _lib.load()
ctx = _lib.default_context(0)
for mb in (8, 24, 64):
    n = mb << 20
    ts = []
    ptrs = []
    for _ in range(4):
        out = C.c_void_p()
        t0 = time.perf_counter()
        lib.mdhip_host_alloc_on(0, n, C.byref(out))
        ts.append(time.perf_counter() - t0)
        ptrs.append(out.value)
    d = torch.empty(n // 8, dtype=torch.float64, device="cuda").normal_()
    torch.cuda.synchronize()
    # D2H into pageable numpy / into the pinned block
    host = np.empty(n // 8)
    t0 = time.perf_counter()
    for _ in range(5):
        torch.from_numpy(host).copy_(d)
    t_page = (time.perf_counter() - t0) / 5
    pinned = torch.empty(n // 8, dtype=torch.float64, pin_memory=True)
    t0 = time.perf_counter()
    for _ in range(5):
        pinned.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
    t_pin = (time.perf_counter() - t0) / 5
    fresh = np.empty(n // 8)
    t0 = time.perf_counter()
    torch.from_numpy(fresh).copy_(d)
    t_fresh = time.perf_counter() - t0
    t0 = time.perf_counter()
    a = np.empty(n // 8)
    a[...] = 0
    t_touch = time.perf_counter() - t0
    tf = []
    for p in ptrs:
        t0 = time.perf_counter()
        lib.mdhip_host_free(C.c_void_p(p))
        tf.append(time.perf_counter() - t0)
    print("%3d MB: hipHostMalloc %s ms, free %.2f ms | D2H pageable (warm pages) %.2f ms, (fresh pages) %.2f ms, pinned %.2f ms | "
          "first touch of a fresh numpy array %.2f ms" % (mb, " ".join("%.2f" % (t * 1e3) for t in ts), np.mean(tf) * 1e3,
                                                          t_page * 1e3, t_fresh * 1e3, t_pin * 1e3, t_touch * 1e3), flush=True)
