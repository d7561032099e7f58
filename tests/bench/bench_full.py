#!/usr/bin/env python
"""
tests/bench/bench_full.py — the BASELINE.json configurations at FULL size on one MI355X, wall time of every
library call with the inputs resident in HBM, next to the CPU oracle (oracle/cpu_ref.c, one core) timed
on a bounded sample of the same workload and extrapolated linearly (every one of these costs is linear in
the number of frames / frame pairs).

    python tests/bench/bench_full.py [c3] [c4] [c5]

C3: 100 000 atoms x 1000 frames, L = 104 A: RDF (10 relations, r_cut 20 A, 400 bins) + CN.
C4: 50 000 atoms x 5000 frames random walk: single-origin MSD (allatom), fixed-lag MSD (tao = 4), molecule
    COM (2500 x 16 + 2500 x 4 atoms) + per-type MSD, and the full lag x origin average (max_lag = F - 1).
C5: n = 1e6 x 3 series: FFT and direct ACF, cumulative trapezoid.
Prints one JSON object per line.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def wall(fn, sync, reps=2):
    """Wall time of the last of `reps` calls (the first one grows the library's workspaces)."""
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        out = fn()
        sync()
        dt = time.perf_counter() - t0
    return dt, out


def emit(**kw):
    print(json.dumps({k: (float("%.6g" % v) if isinstance(v, float) else v) for k, v in kw.items()}), flush=True)


def main():
    import torch

    from mdproptools_amd import backend as B
    from mdproptools_amd import synth
    from mdproptools_amd._lib import default_context
    from oracle import cref

    cref.build()
    ctx = default_context(0)
    dev = torch.device("cuda", 0)
    sync = torch.cuda.synchronize
    which = sys.argv[1:] or ["c3", "c4", "c5"]
    cores = os.cpu_count() or 0

    if "c3" in which:
        cfg = synth.rdf_config("C3")
        n, L, F = cfg["n_atoms"], cfg["box_len"], cfg["n_frames"]
        t0 = time.perf_counter()
        xyz = torch.empty((F, 3, n), dtype=torch.float64, device=dev)
        for f0 in range(0, F, 50):
            xyz[f0:f0 + 50] = torch.from_numpy(synth.rdf_frames(n, range(f0, min(F, f0 + 50)), L, cfg["seed_offset"])).to(dev)
        gen_s = time.perf_counter() - t0
        ty = synth.rdf_types(n)
        rel = np.array(synth.ALL_PAIRS_4)
        box = np.full((F, 3), L)
        cuts = synth.cn_cutoffs(len(rel))
        pairs = F * n * (n - 1) // 2
        B.rdf_loop(xyz[:8], ty, box[:8], rel, 20.0, 0.05, 400, per_frame=False)  # warm-up (module load, workspaces)
        t_rdf, (full, part, ov) = wall(lambda: B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=False), sync)
        k_rdf, aux_rdf = ctx.last_kernel_ms()[0], ctx.last_aux_ms()
        t_rdf_pf, _ = wall(lambda: B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=True), sync)
        t_cn, cn = wall(lambda: B.cn_loop(xyz, ty, box, rel, cuts, per_frame=False), sync)
        k_cn = ctx.last_kernel_ms()[0]
        frac_in = float(full.sum()) / 2.0 / pairs
        assert abs(frac_in - 4.0 / 3.0 * np.pi * 20.0 ** 3 / L ** 3) < 1e-4
        # CPU: 1 full-size frame of RDF and of CN (5e9 pairs each, ~1 min each on one core)
        x0 = xyz[0].cpu().numpy()
        tc0 = time.perf_counter()
        cf, cp, _ = cref.rdf_pairs(x0, ty, rel, [L] * 3, 400.0, 0.05, 400)
        cpu_rdf = time.perf_counter() - tc0
        tc0 = time.perf_counter()
        ccn = cref.cn_pairs(x0, ty, rel, [L] * 3, [c * c for c in cuts])
        cpu_cn = time.perf_counter() - tc0
        f0, p0, _ = B.rdf_loop(xyz[:1], ty, box[:1], rel, 20.0, 0.05, 400)
        assert np.array_equal(f0[0], cf) and np.array_equal(p0[0], cp)  # frame 0 against the oracle, bit-exact
        assert np.array_equal(B.cn_loop(xyz[:1], ty, box[:1], rel, cuts)[0], ccn)
        emit(config="C3 rdf", atoms=n, frames=F, gpu_wall_s=t_rdf, gpu_wall_per_frame_output_s=t_rdf_pf,
             pair_kernel_s=k_rdf * 1e-3, prepass_s=aux_rdf * 1e-3, pairs_per_s=pairs / t_rdf,
             fp64_frac_kernel=pairs * 18 / (k_rdf * 1e-3) / 39.3e12,
             cpu_one_frame_s=cpu_rdf, cpu_extrapolated_s=cpu_rdf * F, speedup=cpu_rdf * F / t_rdf, cpu_cores=1,
             host_cores=cores, synth_gen_s=gen_s)
        emit(config="C3 cn", atoms=n, frames=F, gpu_wall_s=t_cn, pair_kernel_s=k_cn * 1e-3,
             pairs_per_s=pairs / t_cn, cpu_one_frame_s=cpu_cn, cpu_extrapolated_s=cpu_cn * F,
             speedup=cpu_cn * F / t_cn)
        del xyz
        torch.cuda.empty_cache()

    if "c4" in which:
        E, F = 50_000, 5000
        g = torch.Generator(device=dev)
        g.manual_seed(synth.BASE_SEED + 4)
        r = torch.empty((F, 3, E), dtype=torch.float64, device=dev)
        r[0] = torch.rand((3, E), generator=g, device=dev, dtype=torch.float64) * 82.8
        for f0 in range(1, F, 250):
            f1 = min(F, f0 + 250)
            st = torch.randn((f1 - f0, 3, E), generator=g, device=dev, dtype=torch.float64) * 0.1
            r[f0:f1] = r[f0 - 1] + torch.cumsum(st, dim=0)
            del st
        pairs = [(0, t) for t in range(F)]
        B.msd_pairs(r[:4], [(0, 1)], [0, E], scale=1e-10)
        t_msd, s1 = wall(lambda: B.msd_pairs(r, pairs, [0, E], scale=1e-10), sync)
        k_msd = ctx.last_kernel_ms()[0]
        t_win, win = wall(lambda: B.msd_windows(r, 4, scale=1e-10), sync)
        k_win = ctx.last_kernel_ms()[0]
        # molecules: 2500 x 16 atoms + 2500 x 4 atoms, masses 1 + type
        off = np.concatenate([np.arange(0, 40_000, 16), np.arange(40_000, 50_001, 4)]).astype(np.int64)
        mass = np.where(np.arange(E) < 40_000, 2.0, 3.0)
        M = len(off) - 1
        com_d = torch.empty((F, 3, M), dtype=torch.float64, device=dev)
        t_com, _ = wall(lambda: B.segment_com(r, mass, off, out=com_d), sync)
        k_com = ctx.last_kernel_ms()[0]
        t_msdc, s2 = wall(lambda: B.msd_pairs(com_d, pairs, [0, 2500, M], scale=1e-10), sync)
        msd_last = s1[-1, 0, 3] / E / 1e-20
        assert abs(msd_last / (3 * 0.01 * (F - 1)) - 1.0) < 0.02, msd_last
        t_lag, lag = wall(lambda: B.lag_msd(r, F - 1, [0, E], scale=1.0), sync)
        np.testing.assert_allclose(lag[1:, 0, 3] / (3 * 0.01 * np.arange(1, F)), 1.0, atol=0.2)
        k_lag = ctx.last_kernel_ms()[0]
        fp_all = F * (F - 1) / 2
        ctx.set_option("lag_variant", 2)
        t_lagf, lagf = wall(lambda: B.lag_msd(r, F - 1, [0, E], scale=1.0), sync)
        k_lagf, bound = ctx.last_kernel_ms()[0], ctx.last_rel_bound()
        ctx.set_option("lag_variant", 1)
        fft_err = float(np.max(np.abs(lagf[1:] - lag[1:]) / lag[1:]))
        # CPU: the oracle's single-origin loop on 40 frame pairs of the full 50k entities
        rs = r[:41].cpu().numpy()
        tc0 = time.perf_counter()
        cs = cref.msd_pairs(rs, [(0, t) for t in range(41)], [0, E])
        cpu_msd = (time.perf_counter() - tc0) / 41
        np.testing.assert_allclose(s1[:41, 0, :] / 1e-20, np.asarray(cs).reshape(41, -1, 4)[:, 0, :], rtol=1e-10)
        emit(config="C4 msd single origin", entities=E, frames=F, gpu_wall_s=t_msd, kernel_s=k_msd * 1e-3,
             frame_pairs_per_s=F / t_msd, hbm_GBps_kernel=24.0 * E * F / (k_msd * 1e-3) / 1e9,
             hbm_frac_kernel=24.0 * E * F / (k_msd * 1e-3) / 8e12,
             cpu_per_frame_pair_s=cpu_msd, cpu_extrapolated_s=cpu_msd * F, speedup=cpu_msd * F / t_msd)
        emit(config="C4 msd fixed lag tao=4", gpu_wall_s=t_win, kernel_s=k_win * 1e-3,
             hbm_GBps_kernel=24.0 * E * (F // 4) / (k_win * 1e-3) / 1e9)
        emit(config="C4 com (5000 molecules) + per-type msd", com_wall_s=t_com, com_kernel_s=k_com * 1e-3,
             com_hbm_GBps_kernel=(32.0 * E + 24.0 * M) * F / (k_com * 1e-3) / 1e9, msd_wall_s=t_msdc)
        emit(config="C4 full lag x origin msd (max_lag = F-1)", gpu_wall_s=t_lag, kernel_s=k_lag * 1e-3,
             frame_pairs=fp_all, frame_pairs_per_s=fp_all / t_lag,
             fp64_TFLOPs_kernel=12.0 * E * fp_all / (k_lag * 1e-3) / 1e12,
             fp64_frac_kernel=12.0 * E * fp_all / (k_lag * 1e-3) / 78.6e12,
             cpu_extrapolated_s=cpu_msd * fp_all)
        emit(config="C4 full lag x origin msd, FFT variant", gpu_wall_s=t_lagf, device_s=k_lagf * 1e-3,
             frame_pairs_per_s=fp_all / t_lagf, max_rel_diff_vs_difference_kernel=fft_err, reported_bound=bound)
        del r, com_d
        torch.cuda.empty_cache()

    if "c5" in which:
        n = 1_000_000
        p = torch.from_numpy(synth.ar1_series(n)).to(dev)
        B.xcorr(p[:, :4096].contiguous(), method=B.XCORR_FFT)
        t_fft, a_fft = wall(lambda: B.xcorr(p, method=B.XCORR_FFT), sync)
        t_dir, a_dir = wall(lambda: B.xcorr(p, method=B.XCORR_DIRECT), sync, reps=1)
        k_dir = ctx.last_kernel_ms()[0]
        t_int, _ = wall(lambda: B.cumtrapz(a_fft, 1e-15), sync)
        # the FFT estimator's rounding error (~1e-16 n acf[0]) is divided by n - k: it reaches ~1e-10 acf[0] in the
        # last lags at n = 1e6, in numpy's FFT as well; the first half of the lags is the meaningful comparison
        err = max(float(np.max(np.abs(a_fft[k] - a_dir[k]))) / float(a_dir[k][0]) for k in range(3))
        err_half = max(float(np.max(np.abs(a_fft[k][:n // 2] - a_dir[k][:n // 2]))) / float(a_dir[k][0]) for k in range(3))
        assert err < 1e-8 and err_half < 1e-10, (err, err_half)
        # CPU: numpy FFT estimator (what the reference calls) on the full series; the direct estimator on 20 000
        # lags of one series, extrapolated by the pair count
        ph = p.cpu().numpy()
        tc0 = time.perf_counter()
        for k in range(3):
            fa = np.fft.fft(ph[k], 2 * n)
            np.fft.ifft(fa * np.conj(fa))[:n].real / (n - np.arange(n))
        cpu_fft = time.perf_counter() - tc0
        tc0 = time.perf_counter()
        cref.xcorr_direct(ph[0], ph[0], n_lags=200)
        cpu_dir = (time.perf_counter() - tc0) / (200 * n - 200 * 199 / 2)  # seconds per sample pair
        sp = 3 * n * (n + 1) / 2
        emit(config="C5 acf fft", n=n, series=3, gpu_wall_s=t_fft, cpu_numpy_s=cpu_fft, speedup=cpu_fft / t_fft)
        emit(config="C5 acf direct", gpu_wall_s=t_dir, kernel_s=k_dir * 1e-3, sample_pairs_per_s=sp / t_dir,
             fp64_frac_kernel=2 * sp / (k_dir * 1e-3) / 78.6e12, fft_vs_direct_max_err_over_acf0=err, fft_vs_direct_max_err_first_half=err_half,
             cpu_extrapolated_s=cpu_dir * sp, speedup=cpu_dir * sp / t_dir)
        emit(config="C5 cumtrapz", gpu_wall_s=t_int)


if __name__ == "__main__":
    main()
