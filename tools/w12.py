#!/usr/bin/env python
"""tools/w12.py <mode> [LIB.so ...] — the fused full-lag MSD kernel for padded length 12288 (csrc/msd_fft_w12.h) on the GPU box.

  check [quick]   against the round-3/4 kernel (lag_fft_kernel 2, padded length 16384) and the exact-difference kernel on
                  shapes that take it (F + max_lag in (8192, 12288]), both sources (lag_direct 0: transposed copy, 2: in-kernel
                  staging): times and relative differences                                  -> profiles/rNN_w12_check.txt
  exp LIB.so ...  C4-shape call time through several BUILDS (-DW12_EXP=bits timing builds give wrong results and only say
                  where the time goes; tools/build_variant.sh makes them), both sources      -> profiles/rNN_w12_exp.txt
  det LIB.so ...  six staged runs per build: bit-identical? relative difference to the copy source
  ramp [LIB.so]   integer ramp x[t][c] = 16384 c + t: MSD(k) = k^2 exactly (the input that exposed the unguarded store
                  hazard of round 5; the regression test is tests/test_gpu_hardening.py)

(One script since round 6; rounds 4-5 kept eight one-off files from the kernel's bring-up.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mdproptools_amd import _lib  # noqa: E402
from mdproptools_amd import backend as B  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "check"
libs = [a for a in sys.argv[2:] if a.endswith(".so")]


def context_of(path):
    if path:
        _lib._lib = None
        _lib.STRICT = False
        _lib.LIB_PATH = os.path.abspath(path)
    return _lib.Context(0)


def walk(F, E, seed, offset=0.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = torch.cumsum(torch.randn((F, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
    if offset:
        r += torch.rand((1, 3, E), generator=g, device="cuda", dtype=torch.float64) * offset
    return r


def mode_check():
    ctx = context_of(libs[0] if libs else None)
    quick = "quick" in sys.argv[2:]
    shapes = [(5000, 4096, [0, 4096]), (4500, 6000, [0, 1000, 6000]), (6144, 2053, [0, 7, 2053]), (4097, 999, [0, 999]),
              (5001, 8192, [0, 100, 100, 8000, 8192])]
    if not quick:
        shapes.append((5000, 50_000, [0, 50_000]))
    for F, E, goff in shapes:
        r = walk(F, E, F + E, 80.0)
        res = {}
        for name, opts in (("w12/copy", {"lag_fft_kernel": 3, "lag_direct": 0}), ("w12/staged", {"lag_fft_kernel": 3, "lag_direct": 2}),
                           ("lds3/staged", {"lag_fft_kernel": 2, "lag_direct": 2}), ("difference", {"lag_variant": 1})):
            if name == "difference" and E > 10_000:
                continue
            for k, v in opts.items():
                ctx.set_option(k, v)
            ctx.set_option("lag_variant", opts.get("lag_variant", 2))
            try:
                out = B.lag_msd(r, F - 1, goff, scale=1.0, ctx=ctx)
                ms = []
                for _ in range(3):
                    out = B.lag_msd(r, F - 1, goff, scale=1.0, ctx=ctx)
                    ms.append(ctx.last_kernel_ms()[0])
                res[name] = out
                print("F %5d E %6d %-12s %-28s %8.3f ms  bound %.2e" % (F, E, name, ctx.last_kernel_name()[:28], min(ms),
                                                                      ctx.last_rel_bound()), flush=True)
            finally:
                for k in opts:
                    ctx.set_option(k, -1 if k != "lag_fft_kernel" else 3)
                ctx.set_option("lag_variant", -1)
        ref = res.get("difference", res["lds3/staged"])
        for name in ("w12/copy", "w12/staged", "lds3/staged"):
            a, b = res[name][1:], ref[1:]
            m = b != 0
            print("   %-12s max rel diff vs %s: %.3e" % (name, "difference" if "difference" in res else "lds3",
                                                        float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m])))), flush=True)
    print("fallbacks", ctx.fallbacks())


def mode_exp():
    F, E = 5000, 50_000
    r = walk(F, E, 1)
    for p in libs:
        ctx = context_of(p)
        ctx.set_option("lag_variant", 2)
        row = []
        for src in (0, 2):
            ctx.set_option("lag_direct", src)
            ms = []
            for _ in range(4):
                B.lag_msd(r, F - 1, [0, E], ctx=ctx)
                ms.append(ctx.last_kernel_ms()[0])
            row.append(min(ms[1:]))
        print("%-28s copy %.3f ms   staged %.3f ms   %s" % (os.path.basename(p), row[0], row[1], ctx.last_kernel_name()), flush=True)
        ctx.close()


def mode_det():
    F, E = 5000, 8192
    r = walk(F, E, F + E)
    for p in libs or [None]:
        ctx = context_of(p)
        ctx.set_option("lag_variant", 2)
        for kern in (3, 2):
            ctx.set_option("lag_fft_kernel", kern)
            ctx.set_option("lag_direct", 0)
            ref = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
            ctx.set_option("lag_direct", 2)
            outs = [B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx) for _ in range(6)]
            same = sum(np.array_equal(o, outs[0]) for o in outs)
            rel = max(float(np.max(np.abs(o[1:] - ref[1:]) / ref[1:])) for o in outs)
            print("%-24s kern %d %-22s identical runs %d/6  max rel vs copy %.2e" % (os.path.basename(p or "libmdhip.so"), kern,
                                                                                    ctx.last_kernel_name(), same, rel), flush=True)
        ctx.close()


def mode_ramp():
    ctx = context_of(libs[0] if libs else None)
    for F, E in ((6144, 1024), (6144, 2048), (6144, 4096), (6000, 1024), (4097, 2048), (5000, 1024)):
        t = torch.arange(F, dtype=torch.float64, device="cuda")[:, None, None]
        c = torch.arange(3 * E, dtype=torch.float64, device="cuda").reshape(1, 3, E)
        r = (16384.0 * c + t).contiguous()
        ctx.set_option("lag_variant", 2)
        k2 = np.arange(F, dtype=np.float64) ** 2
        for src in (0, 2):
            ctx.set_option("lag_direct", src)
            o = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
            d = np.abs(o[1:, 0, :3] - k2[1:, None])
            print("F %d E %d src %d %s max |msd - k^2| %.3g (axis maxima %s) fallbacks %d" % (
                F, E, src, ctx.last_kernel_name(), d.max(), d.max(axis=0), ctx.fallbacks()), flush=True)


{"check": mode_check, "exp": mode_exp, "det": mode_det, "ramp": mode_ramp}[mode]()
