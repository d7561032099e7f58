import torch, numpy as np
y = torch.randn(3, 1_000_000, dtype=torch.float64, device="cuda")
o = torch.empty_like(y)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5): o.copy_(y)
    ts = []
    for _ in range(20):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(s); o.copy_(y); b.record(s); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    print("device copy 24 MB -> 24 MB: min %.1f us median %.1f us" % (min(ts), float(np.median(ts))))
    ts = []
    for _ in range(20):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(s); torch.add(y, 1.0, out=o); b.record(s); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    print("elementwise add 24 MB -> 24 MB: min %.1f us median %.1f us" % (min(ts), float(np.median(ts))))
    z = torch.empty(1, device="cuda")
    ts = []
    for _ in range(20):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(s); z.add_(1.0); b.record(s); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    print("one-element kernel: min %.1f us median %.1f us" % (min(ts), float(np.median(ts))))
