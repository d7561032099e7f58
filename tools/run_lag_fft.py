"""One C4-size full-lag MSD through the FFT variant (for rocprofv3): python3 tools/run_lag_fft.py [F E reps] [key=value ...]
(context options, e.g. lag_fft_kernel=2 lag_direct=0)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from mdproptools_amd import backend as B

nums = [a for a in sys.argv[1:] if "=" not in a]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
F, E, reps = (int(a) for a in (nums + ["5000", "50000", "3"][len(nums):]))
ctx = B.default_context()
g = torch.Generator(device="cuda").manual_seed(1)
r = torch.cumsum(torch.randn((F, 3, E), dtype=torch.float64, device="cuda", generator=g) * 0.1, dim=0)
ctx.set_option("lag_variant", 2)
for k, v in opts:
    ctx.set_option(k, int(v))
for _ in range(reps):
    out = B.lag_msd(r, F - 1, [0, E])
    print("device ms", ctx.last_kernel_ms()[0], ctx.last_kernel_name(), "bound", ctx.last_rel_bound(), "msd[1]", out[1, 0, 3])
