#!/usr/bin/env python
"""
tools/sweep_rdf.py — A/B timings of the pair kernel's organisation knobs (results never depend on them):
BASELINE C2 (10k atoms x 200 frames, L = 50) and C3 geometry (100k atoms, 16 frames, L = 104), frame-summed
and per-frame output. Prints one line per setting: kernel ms, pre-pass ms, kernel name.

    python tools/sweep_rdf.py [c2] [c3] [c3cn]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(B, ctx, fn, reps=4):
    fn()
    best, aux = 1e30, 0.0
    for _ in range(reps):
        fn()
        ms = ctx.last_kernel_ms()[0]
        if ms < best:
            best, aux = ms, ctx.last_aux_ms()
    return best, aux, ctx.last_kernel_name()


def main():
    import torch

    from mdproptools_amd import backend as B
    from mdproptools_amd import synth
    from mdproptools_amd._lib import Context

    which = sys.argv[1:] or ["c2", "c3", "c3cn"]
    dev = torch.device("cuda", 0)
    settings = [
        dict(),
        dict(rdf_sj=2),
        dict(rdf_sj=0),
        dict(rdf_jsplit=1),
        dict(rdf_jsplit=2),
        dict(rdf_jsplit=4),
        dict(rdf_slots=4),
        dict(rdf_slots=64),
        dict(rdf_rows=0),
        dict(rdf_inflight=4),
        dict(rdf_inflight=16),
        dict(rdf_sort=0),
        dict(rdf_sort=1),
        dict(rdf_cull=0),
    ]
    rel = np.array(synth.ALL_PAIRS_4)
    for name in which:
        cfg = synth.rdf_config("C2" if name == "c2" else "C3")
        n, L = cfg["n_atoms"], cfg["box_len"]
        F = cfg["n_frames"] if name == "c2" else 16
        xyz = torch.from_numpy(synth.rdf_frames(n, range(F), L, cfg["seed_offset"])).to(dev)
        ty = synth.rdf_types(n)
        box = np.full((F, 3), L)
        for st in settings:
            if name != "c2" and st.get("rdf_cull") == 0:
                continue
            for per_frame in (False, True):
                ctx = Context(0)
                for k, v in st.items():
                    ctx.set_option(k, v)
                if name == "c3cn":
                    cuts = synth.cn_cutoffs(len(rel))
                    fn = lambda: B.cn_loop(xyz, ty, box, rel, cuts, per_frame=per_frame, ctx=ctx)
                else:
                    fn = lambda: B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=per_frame, ctx=ctx)
                ms, aux, kn = run(B, ctx, fn)
                print("%-5s %-22s per_frame=%d  kernel %8.3f ms  prepass %6.3f ms  %s"
                      % (name, st or "default", per_frame, ms, aux, kn), flush=True)
                ctx.close()
        del xyz


if __name__ == "__main__":
    main()
