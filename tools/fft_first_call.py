import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from mdproptools_amd import backend as B
from mdproptools_amd._lib import default_context
ctx = default_context(0)
n = int(sys.argv[1])
s = np.random.default_rng(0).standard_normal(n)
B.cumtrapz(s, 1.0)  # context + module load
t0 = time.perf_counter(); B.xcorr(s, method=B.XCORR_FFT); t1 = time.perf_counter(); B.xcorr(s, method=B.XCORR_FFT); t2 = time.perf_counter()
s2 = np.random.default_rng(0).standard_normal(n + 1000)
B.xcorr(s2, method=B.XCORR_FFT); t3 = time.perf_counter()
print("n=%d first %.3f s second %.4f s other length %.3f s" % (n, t1 - t0, t2 - t1, t3 - t2))
