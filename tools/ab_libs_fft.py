"""tools/ab_libs_fft.py LIB.so ... — FFT autocorrelation at C5 shape (3 x 1e6) and 1 x 2^20 through several builds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdproptools_amd import _lib, backend as B
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
g = torch.Generator(device="cuda").manual_seed(3)
data = {(n, P): torch.randn((P, n), dtype=torch.float64, device="cuda", generator=g) for n, P in ((1000000, 3), (1048576, 1), (200000, 2))}
ref = {}
for rnd in range(2):
    for p in libs:
        _lib._lib = None; _lib.STRICT = False; _lib.LIB_PATH = os.path.abspath(p)
        ctx = _lib.Context(0)
        row = []
        for (n, P), a in data.items():
            out = torch.empty((P, n), dtype=torch.float64, device="cuda")
            ts = []
            for _ in range(5):
                B.xcorr(a, None, method=B.XCORR_FFT, ctx=ctx, out=out)
                ts.append(ctx.last_kernel_ms()[0])
            o = out.cpu().numpy()
            same = np.array_equal(o, ref.setdefault((n, P), o))
            row.append("n %d P %d %.1f us%s" % (n, P, min(ts[1:]) * 1e3, "" if same else " DIFFERENT"))
        print("%-24s %s" % (os.path.basename(p), "   ".join(row)), flush=True)
        ctx.close()
