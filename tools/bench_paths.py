#!/usr/bin/env python
"""
tools/bench_paths.py — timings of every kernel on the hot path at BASELINE.json-like sizes (one GPU),
inputs resident in HBM. Prints one JSON object per line; `kernel_ms` is the HIP-event time of the
dominant kernel inside the library call (mdhip_last_kernel_ms), `call_ms` the wall time of the call.
Roofline figures use SURVEY.md §8d's algorithmic work per unit.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, ctx, reps=3):
    fn()
    best_call, best_k = 1e30, 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best_call = min(best_call, (time.perf_counter() - t0) * 1e3)
        best_k = min(best_k, ctx.last_kernel_ms()[0])
    return best_call, best_k


def main():
    import torch

    from mdproptools_amd import backend as B
    from mdproptools_amd import synth
    from mdproptools_amd._lib import default_context

    ctx = default_context(0)
    dev = torch.device("cuda", 0)
    which = sys.argv[1:] or ["rdf_c3", "cn_c3", "rect", "residence", "msd", "lag", "xcorr", "scan", "com"]
    out = []

    if "rdf_c3" in which or "cn_c3" in which:
        cfg = synth.rdf_config("C3")
        n, L, F = cfg["n_atoms"], cfg["box_len"], 16  # 16 of the 1000 frames: cost is linear in frames
        xyz = torch.from_numpy(synth.rdf_frames(n, range(F), L, cfg["seed_offset"])).to(dev)
        ty = synth.rdf_types(n)
        rel = np.array(synth.ALL_PAIRS_4)
        box = np.full((F, 3), L)
        pairs = F * n * (n - 1) // 2
        if "rdf_c3" in which:
            call, k = timed(lambda: B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=False), ctx)
            out.append(dict(path="rdf C3 (100k atoms, 16 frames, 3% in cutoff)", call_ms=call, kernel_ms=k,
                            pairs_per_s=pairs / (k * 1e-3), fp64_frac=pairs * 18 / (k * 1e-3) / 39.3e12))
        if "cn_c3" in which:
            cuts = synth.cn_cutoffs(len(rel))
            call, k = timed(lambda: B.cn_loop(xyz, ty, box, rel, cuts, per_frame=False), ctx)
            out.append(dict(path="cn C3 (per-relation cutoffs 2.3-6.8 A)", call_ms=call, kernel_ms=k,
                            pairs_per_s=pairs / (k * 1e-3), fp64_frac=pairs * 18 / (k * 1e-3) / 39.3e12))
        del xyz

    if "rect" in which:
        from mdproptools_amd._lib import Context

        n, m, L, F = 100_000, 10_000, 104.0, 16  # C3 geometry with 10-atom molecules
        rng = np.random.default_rng(7)
        xyz = torch.from_numpy(rng.random((F, 3, n)) * L).to(dev)
        sites = torch.from_numpy(rng.random((F, 3, m)) * L).to(dev)
        ty = synth.rdf_types(n)
        st = (1 + np.arange(m) % 3).astype(np.int32)
        rel = np.array([[a, b] for a in range(1, 5) for b in range(1, 4)])
        box = np.full((F, 3), L)
        for cull, tag in ((0, "dense"), (-1, "culled")):
            c2 = Context(0)
            c2.set_option("rdf_cull", cull)
            call, k = timed(lambda: B.rdf_mol_loop(xyz, ty, sites, st, box, rel, 20.0, 0.05, 400, per_frame=False,
                                                   ctx=c2), c2)
            out.append(dict(path="rdf atoms x sites, %s (100k atoms x 10k sites, 16 frames, 12 relations)" % tag,
                            call_ms=call, kernel_ms=k, pairs_per_s=F * n * m / (k * 1e-3), kernel=c2.last_kernel_name()))
            c2.close()
        del xyz, sites

    if "residence" in which:
        ni, nj, F, L = 2000, 20_000, 500, 60.0  # e.g. 2000 cations against 20 000 solvent oxygens
        rng = np.random.default_rng(9)
        xi = rng.random((1, 3, ni)) * L + np.cumsum(rng.normal(0, 0.05, (F, 3, ni)), axis=0)
        xj = rng.random((1, 3, nj)) * L + np.cumsum(rng.normal(0, 0.05, (F, 3, nj)), axis=0)
        di, dj = torch.from_numpy(xi).to(dev), torch.from_numpy(xj).to(dev)
        box = np.full((F, 3), L)
        call, k = timed(lambda: B.shell_residence(di, dj, box, 0.0, 3.0 ** 2), ctx)
        counts, nrec = B.shell_residence(di, dj, box, 0.0, 9.0)
        out.append(dict(path="residence autocorrelation (2000 x 20000 atoms, 500 frames, shell 0-3 A)", call_ms=call,
                        kernel_ms=k, pair_tests_per_s=F * ni * nj / (k * 1e-3), records=nrec))

    if "msd" in which or "lag" in which:
        E, F = 50_000, 1000  # C4 entities, 1000 of its 5000 frames
        r = torch.from_numpy(synth.random_walk(E, F)).to(dev)
        if "msd" in which:
            pairs = [(0, t) for t in range(F)]
            call, k = timed(lambda: B.msd_pairs(r, pairs, [0, E], scale=1e-10), ctx)
            out.append(dict(path="msd single origin (50k entities, 1000 frame pairs)", call_ms=call, kernel_ms=k,
                            frame_pairs_per_s=F / (k * 1e-3), hbm_GBps=24.0 * E * F / (k * 1e-3) / 1e9,
                            hbm_frac=24.0 * E * F / (k * 1e-3) / 8e12))
            call, k = timed(lambda: B.msd_windows(r, 4, scale=1e-10), ctx)
            out.append(dict(path="msd fixed lag tao=4 (250 kept frames)", call_ms=call, kernel_ms=k,
                            hbm_GBps=24.0 * E * (F // 4) / (k * 1e-3) / 1e9))
        if "lag" in which:
            El = 4096
            rl = r[:, :, :El].contiguous()
            call, k = timed(lambda: B.lag_msd(rl, F - 1, [0, El], scale=1.0), ctx, reps=2)
            fp = F * (F - 1) / 2
            out.append(dict(path="lag msd full (4096 entities x 1000 frames, all lags)", call_ms=call, kernel_ms=k,
                            frame_pairs_per_s=fp / (k * 1e-3), fp64_TFLOPs=12.0 * El * fp / (k * 1e-3) / 1e12,
                            fp64_frac=12.0 * El * fp / (k * 1e-3) / 78.6e12))
        del r

    if "xcorr" in which or "scan" in which:
        for n in (100_000, 1_000_000):
            p = torch.from_numpy(synth.ar1_series(n)).to(dev)
            if "xcorr" in which:
                call, k = timed(lambda: B.xcorr(p, method=B.XCORR_FFT), ctx)
                byts = 3 * 2 * 16 * 2 * n * 3  # 2 x 16 B x 2n x 3 transforms per series
                out.append(dict(path="acf fft n=%d x3 series" % n, call_ms=call, kernel_ms=k,
                                hbm_GBps=byts / (k * 1e-3) / 1e9))
                call, k = timed(lambda: B.xcorr(p, method=B.XCORR_DIRECT), ctx, reps=1 if n > 200_000 else 3)
                sp = 3 * n * (n + 1) / 2
                out.append(dict(path="acf direct n=%d x3 series" % n, call_ms=call, kernel_ms=k,
                                sample_pairs_per_s=sp / (k * 1e-3), fp64_TFLOPs=2 * sp / (k * 1e-3) / 1e12,
                                fp64_frac=2 * sp / (k * 1e-3) / 78.6e12))
            if "scan" in which:
                call, k = timed(lambda: B.cumtrapz(p, 1e-15), ctx)
                out.append(dict(path="cumtrapz n=%d x3" % n, call_ms=call, kernel_ms=k,
                                hbm_GBps=16.0 * 3 * n / (k * 1e-3) / 1e9))

    if "com" in which:
        n, F = 100_000, 64
        attr = torch.from_numpy(np.random.default_rng(1).random((F, 3, n))).to(dev)
        mass = 1.0 + (np.arange(n) % 4)
        off = np.arange(0, n + 1, 10, dtype=np.int64)
        call, k = timed(lambda: B.segment_com(attr, mass, off), ctx)
        out.append(dict(path="segment_com 100k atoms x 64 frames (10-atom molecules)", call_ms=call, kernel_ms=k,
                        hbm_GBps=(24.0 * n + 2.4 * n) * F / (k * 1e-3) / 1e9))
        q = np.where(np.arange(n) % 10 == 0, 1.0, -1.0 / 9)
        st = (np.arange(len(off) - 1) * 3 // (len(off) - 1)).astype(np.int32)
        call, k = timed(lambda: B.charge_flux(attr, mass, q, off, st, 3, 1e5, 1.6e-19), ctx)
        out.append(dict(path="charge_flux 100k atoms x 64 frames", call_ms=call, kernel_ms=k,
                        hbm_GBps=24.0 * n * F / (k * 1e-3) / 1e9))

    for o in out:
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in o.items()}))


if __name__ == "__main__":
    main()
