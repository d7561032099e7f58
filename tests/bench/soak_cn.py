#!/usr/bin/env python
"""
tests/bench/soak_cn.py [trials] [seed] [oracle] — long randomised differential run of the one-sweep RDF + CN call
(mdhip_rdf_cn_atomic) against the two separate calls, on the generator of the packed-sweep stress set
(tests/test_gpu_parity.py::_pk_case: lattices on bin edges, blobs, NPT boxes, strays box lengths outside the cell,
ordered rows and class rows, cutoff on and inside a bin) with coordination cutoffs on bin edges, between them, shared,
zero and beyond r_cut. With `oracle`, the CN counts of frame 0 of every 5th case are also compared with
oracle/cpu_ref.c. Prints the first mismatch and exits 1, or a summary.
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
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    oracle = len(sys.argv) > 3 and sys.argv[3] == "oracle"
    if oracle:
        from oracle import cref

        cref.build()
    ctx = Context(0)
    ctx.set_option("rdf_cull", 1)
    ctx.set_option("cn_pk", 1)
    edge_table = Context(0)  # mdhip_cn_atomic on its f64 edge-table kernel: the independent side of the comparison
    edge_table.set_option("rdf_cull", 1)
    edge_table.set_option("cn_pk", 0)
    fused, pairs = 0, 0
    for trial in range(trials):
        xyz, ty, box, rel, r_cut, bin_size, nbins = _pk_case(rng, trial)
        R = len(rel)
        cuts = list(rng.uniform(0.05, 0.999, R) * r_cut)
        if trial % 3 == 0:
            cuts[0] = bin_size * int(rng.integers(1, nbins))     # exactly a bin edge
        if trial % 4 == 1 and R > 1:
            cuts[1] = cuts[0]
        if trial % 5 == 2:
            cuts[-1] = 0.0
        if trial % 9 == 7:
            cuts[0] = r_cut                                        # the RDF cutoff itself
        if trial % 11 == 10:
            cuts[0] = 1.2 * r_cut                                  # beyond it: two sweeps inside the call
        per_frame = bool(trial % 2)
        a = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=ctx)
        cn = B.cn_loop(xyz, ty, box, rel, cuts, per_frame=per_frame, ctx=edge_table)
        cn_pk = B.cn_loop(xyz, ty, box, rel, cuts, per_frame=per_frame, ctx=ctx)  # coarse histogram + split bins
        if not np.array_equal(cn_pk, cn):
            print("MISMATCH (cn through the packed sweep) trial %d cuts=%s\n edge table %s\n packed %s" % (trial, cuts, cn, cn_pk))
            sys.exit(1)
        f, p_, ov, cn2 = B.rdf_cn_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, cuts, per_frame=per_frame, ctx=ctx)
        name = ctx.last_kernel_name()
        fused += name.endswith(", true>")
        n = xyz.shape[2]
        pairs += xyz.shape[0] * n * (n - 1) // 2
        ok = np.array_equal(f, a[0]) and np.array_equal(p_, a[1]) and ov == a[2] and np.array_equal(cn2, cn)
        if ok and oracle and trial % 5 == 0:
            ref = cref.cn_pairs(xyz[0], ty, rel, box[0], [c * c for c in cuts])
            got = cn2[0] if per_frame else None
            ok = (not per_frame) or np.array_equal(got, ref)
        if not ok:
            print("MISMATCH trial %d n=%d box=%s r_cut=%.4f bin=%.3f cuts=%s kernel=%s\n cn two calls %s\n cn one sweep %s"
                  % (trial, n, box[0], r_cut, bin_size, cuts, name, cn, cn2))
            sys.exit(1)
        if trial % 50 == 49:
            print("trial %d ok (%d in one sweep, %.3g atom pairs so far)" % (trial + 1, fused, pairs), flush=True)
    print("soak_cn: %d cases identical, one-sweep kernel in %d, %.4g atom pairs" % (trials, fused, pairs))


if __name__ == "__main__":
    main()
