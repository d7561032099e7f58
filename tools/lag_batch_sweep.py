#!/usr/bin/env python3
"""tools/lag_batch_sweep.py — the batched full-lag path (10 000 frames x 50k entities) against the size of a batch of series
(option lag_batch_mb): does the first pass's output stay in the memory-side cache when a batch is small?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import torch

    from mdproptools_amd import backend as B
    from mdproptools_amd import synth
    from mdproptools_amd._lib import default_context
    from tools.run_secondary import c4_walk

    ctx = default_context(0)
    dev = torch.device("cuda", 0)
    E, F = 50_000, int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    r = c4_walk(torch, dev, synth, E, F)
    ctx.set_option("lag_variant", 2)
    ref = None
    for mb in (4096, 2048, 1024, 512, 256, 128, 64, 32, 8192, 16384):
        ctx.set_option("lag_batch_mb", mb)
        ms = []
        for _ in range(3):
            out = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
            ms.append(ctx.last_kernel_ms()[0])
        if ref is None:
            ref = out
        d = float(np.max(np.abs(out[1:] - ref[1:]) / ref[1:]))
        print("lag_batch_mb %6d  kernel ms %8.3f   max rel diff vs 4096: %.2e  bound %.1e" % (mb, np.median(ms), d, ctx.last_rel_bound()), flush=True)


if __name__ == "__main__":
    main()
