"""Three-launch FFT autocorrelation (option fft_mid, fft_mid_acf_kernel) against the four-launch path and the direct lag
sums: python tools/fft_mid_check.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
ctx = _lib.Context(0)
g = torch.Generator(device="cuda").manual_seed(3)
for n, P in ((131073, 1), (200000, 2), (262144, 3), (300001, 1), (524288, 2), (1000000, 3), (1048576, 1), (70000, 2)):
    a = torch.randn((P, n), dtype=torch.float64, device="cuda", generator=g)
    a = torch.cumsum(a, dim=1) * 0.01 + a
    res = {}
    for mid in (0, 1, 2):
        ctx.set_option("fft_mid", mid)
        out = torch.empty((P, n), dtype=torch.float64, device="cuda")
        ts = []
        for _ in range(4):
            B.xcorr(a, None, method=B.XCORR_FFT, ctx=ctx, out=out)
            ts.append(ctx.last_kernel_ms()[0])
        res[mid] = (out.cpu().numpy(), min(ts[1:]), ctx.last_kernel_ms()[1])
    d = np.abs(res[1][0] - res[0][0]).max() / np.abs(res[0][0][:, 0]).max()
    # a few lags by direct sums in numpy
    ah = a.cpu().numpy()
    chk = 0.0
    for k in (0, 1, 7, n // 3, n - 2):
        ref = np.array([np.dot(ah[p, k:], ah[p, :n - k]) / (n - k) for p in range(P)])
        chk = max(chk, np.abs(res[1][0][:, k] - ref).max() / np.abs(res[1][0][:, 0]).max())
    print("n %8d P %d  fft_mid=2 %.1f us " % (n, P, res[2][1] * 1e3), np.abs(res[2][0] - res[1][0]).max())
    print("n %8d P %d  four launches %.1f us (%d)  three %.1f us (%d)   max |diff| / acf[0] %.2e   vs direct sums %.2e"
          % (n, P, res[0][1] * 1e3, res[0][2], res[1][1] * 1e3, res[1][2], d, chk), flush=True)
