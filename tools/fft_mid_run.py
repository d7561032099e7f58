import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
ctx = _lib.Context(0)
g = torch.Generator(device="cuda").manual_seed(3)
n, P = 1000000, 3
a = torch.randn((P, n), dtype=torch.float64, device="cuda", generator=g)
out = torch.empty((P, n), dtype=torch.float64, device="cuda")
for mid in (0, 1):
    ctx.set_option("fft_mid", mid)
    for _ in range(5):
        B.xcorr(a, None, method=B.XCORR_FFT, ctx=ctx, out=out)
torch.cuda.synchronize()
