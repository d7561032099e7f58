#!/usr/bin/env python
"""
tools/run_secondary.py <workload> [reps] — ONE library call of the non-pair kernels at the size bench.py's c4 / c5
legs use, repeated `reps` times (default 3), for `rocprofv3 --kernel-trace --stats` and `--pmc` passes over a single
call (tools/pmc_secondary.sh). Prints one line per repetition: dominant kernel, its HIP-event time.

  msd_pairs    single-origin MSD, 50k entities x 5000 frame pairs            (msd_pairs_kernel)
  msd_windows  fixed-lag windows, tao 4                                      (msd_windows_kernel)
  com          per-molecule centres, 2500 x 16 + 2500 x 4 atoms, 5000 frames (segment_staged_kernel<false,...>)
  flux         charge flux of the same molecules, 5000 frames                (segment_staged_kernel<true,...>, type_sum)
  lag_fft      full lag x origin MSD, default path                           (msd_power_lds_kernel + inverse)
  lag_diff     the same through the exact-difference kernel                  (lag_msd_lds_kernel)
  acf_fft      3 x 1e6-sample autocorrelation by FFT                         (fft_pass_kernel<...>, xcorr_spectrum)
  acf_direct   the same by direct lag sums                                   (xcorr_direct_kernel)
  cumtrapz     running integral of 3 x 1e6 samples                           (scan kernels)
  lag_long     full lag x origin MSD of 10 000 frames x 50k entities         (the batched path: transforms in HBM, fft_pow2.hip)
  residence    shell residence counts, 315 x 11 280 atoms x 1000 frames      (shell_pairs_kernel, sort, residence_lag_kernel)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def c4_walk(torch, device, synth, E=50_000, F=5000, block=250):
    g = torch.Generator(device=device)
    g.manual_seed(synth.BASE_SEED + 4)
    r = torch.empty((F, 3, E), dtype=torch.float64, device=device)
    r[0] = torch.rand((3, E), generator=g, device=device, dtype=torch.float64) * 82.8
    for f0 in range(1, F, block):
        f1 = min(F, f0 + block)
        st = torch.randn((f1 - f0, 3, E), generator=g, device=device, dtype=torch.float64) * 0.1
        r[f0:f1] = r[f0 - 1] + torch.cumsum(st, dim=0)
        del st
    return r


def main():
    import torch

    from mdproptools_amd import backend as B
    from mdproptools_amd import synth
    from mdproptools_amd._lib import default_context

    what = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    ctx = default_context(0)
    for kv in os.environ.get("MDHIP_OPTS", "").split():  # (A/B runs: "key=value ..." context options)
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    dev = torch.device("cuda", 0)
    E, F = 50_000, 5000
    off = np.concatenate([np.arange(0, 40_000, 16), np.arange(40_000, 50_001, 4)]).astype(np.int64)
    mass = np.where(np.arange(E) < 40_000, 2.0, 3.0)
    M = len(off) - 1
    if what == "lag_long":
        F = int(os.environ.get("LAG_F", 10_000))
    if what in ("msd_pairs", "msd_windows", "com", "flux", "lag_fft", "lag_diff", "lag_long"):
        r = c4_walk(torch, dev, synth, E, F)
    if what == "msd_pairs":
        pairs = [(0, t) for t in range(F)]
        call = lambda: B.msd_pairs(r, pairs, [0, E], scale=1e-10, ctx=ctx)
    elif what == "msd_windows":
        call = lambda: B.msd_windows(r, 4, scale=1e-10, ctx=ctx)
    elif what == "com":
        out = torch.empty((F, 3, M), dtype=torch.float64, device=dev)
        call = lambda: B.segment_com(r, mass, off, out=out, ctx=ctx)
    elif what == "flux":
        q = np.where(np.arange(E) < 40_000, 0.125, -0.25)
        st = np.concatenate([np.zeros(2500, np.int32), np.ones(2500, np.int32)])
        call = lambda: B.charge_flux(r, mass, q, off, st, 2, 1e5, 1.602e-19, ctx=ctx)
    elif what in ("lag_fft", "lag_long"):
        ctx.set_option("lag_variant", 2)
        call = lambda: B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
    elif what == "lag_diff":
        ctx.set_option("lag_variant", 1)
        call = lambda: B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
    elif what == "residence":
        ri, rj = synth.residence_walk()
        xi, xj = torch.from_numpy(ri).to(dev), torch.from_numpy(rj).to(dev)
        bx = np.full((len(ri), 3), 104.0)
        call = lambda: B.shell_residence(xi, xj, bx, 0.0, 2.325 ** 2, ctx=ctx)
    elif what in ("acf_fft", "acf_direct", "cumtrapz"):
        p = torch.from_numpy(synth.ar1_series(1_000_000)).to(dev)
        if what == "acf_fft":
            call = lambda: B.xcorr(p, method=B.XCORR_FFT, ctx=ctx)
        elif what == "acf_direct":
            call = lambda: B.xcorr(p, method=B.XCORR_DIRECT, ctx=ctx)
        else:
            call = lambda: B.cumtrapz(p, 1e-15, ctx=ctx)
    else:
        raise SystemExit("unknown workload %r" % what)
    for _ in range(reps):
        call()
        print(what, ctx.last_kernel_name(), "%.4f ms" % ctx.last_kernel_ms()[0], flush=True)


if __name__ == "__main__":
    main()
