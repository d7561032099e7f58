#!/usr/bin/env python
"""
tests/bench/soak_pk.py [trials] [seed] [oracle] — long randomised differential run of the packed-f32 classification sweep
(rdf_pk = 1) against the all-f64 sweep (rdf_pk = 0): the generator of
tests/test_gpu_parity.py::test_packed_f32_sweep_equals_f64_sweep with more trials and larger frames. Prints the first
mismatch and exits 1, or a summary (cases, how often the packed kernel engaged, atom pairs compared). With `oracle`
frame 0 of every case is also compared with oracle/cpu_ref.c (slow: use a few hundred trials).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    from mdproptools_amd import backend as B
    from mdproptools_amd._lib import Context
    from test_gpu_parity import _pk_case

    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle = len(sys.argv) > 3 and sys.argv[3] == "oracle"
    if oracle:
        from oracle import cref

        cref.build()
    f64, pk = Context(0), Context(0)
    for ctx, v in ((f64, 0), (pk, 1)):
        ctx.set_option("rdf_cull", 1)
        ctx.set_option("rdf_pk", v)
    engaged, pairs = 0, 0
    layouts = {}
    for trial in range(trials):
        xyz, ty, box, rel, r_cut, bin_size, nbins = _pk_case(rng, trial)
        per_frame = bool(trial % 2)
        a = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=f64)
        b = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=pk)
        engaged += any(t in pk.last_kernel_name() for t in ("<3,", "<4,", "<5,", "<6,"))
        key = (pk.last_kernel_name(), pk.last_kernel_ms()[1])  # (kernel instance, launches: > 1 = class rows in passes)
        layouts[key] = layouts.get(key, 0) + 1
        n = xyz.shape[2]
        pairs += xyz.shape[0] * n * (n - 1) // 2
        if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]):
            d = b[0].astype(np.int64) - a[0].astype(np.int64)
            print("MISMATCH trial %d n=%d box=%s r_cut=%.4f bin=%.3f kernel=%s: %d words differ, sum %d, overflow %d vs %d"
                  % (trial, n, box[0], r_cut, bin_size, pk.last_kernel_name(), np.count_nonzero(d), d.sum(), a[2], b[2]))
            sys.exit(1)
        if oracle:
            f0 = 0
            cf, cp, _ = cref.rdf_pairs(xyz[f0], ty, rel, box[f0], r_cut * r_cut, bin_size, nbins)
            bf = b[0][f0] if per_frame else None
            if per_frame and not (np.array_equal(bf, cf) and np.array_equal(b[1][f0], cp)):
                print("ORACLE MISMATCH trial %d n=%d box=%s r_cut=%.4f bin=%.3f" % (trial, n, box[0], r_cut, bin_size))
                sys.exit(1)
        if trial % 50 == 49:
            print("trial %d ok (%d with the packed kernel, %.3g atom pairs so far)" % (trial + 1, engaged, pairs), flush=True)
    for (name, launches), cnt in sorted(layouts.items()):
        print("   %5d x %s, %d launch(es)" % (cnt, name, launches))
    print("soak_pk: %d cases identical, packed kernel in %d, %.4g atom pairs" % (trials, engaged, pairs))


if __name__ == "__main__":
    main()
