#!/usr/bin/env python
"""
tests/bench/soak_cull.py [trials] [seed] [oracle] — long randomised differential run of the culled scalar-j sweep against the
dense sweep (same generator as tests/test_gpu_parity.py::test_culled_path_randomised_against_dense, more trials,
also atoms x sites). Prints the first mismatch and exits 1, or a summary.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    from mdproptools_amd import backend as B
    from mdproptools_amd._lib import Context

    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle = len(sys.argv) > 3 and sys.argv[3] == "oracle"  # also compare with oracle/cpu_ref.c (slow: fewer atoms)
    if oracle:
        from oracle import cref

        cref.build()
    dense, culled = Context(0), Context(0)
    dense.set_option("rdf_cull", 0)
    culled.set_option("rdf_cull", 1)
    pairs = 0
    for trial in range(trials):
        n = int(rng.integers(2100, 3000 if oracle else 9000))
        L = rng.uniform(15.0, 70.0, 3)
        lo = rng.uniform(-1.5, 1.5, 3) * L
        F = int(rng.integers(1, 4))
        if trial % 3 == 0:
            centres = rng.uniform(0, 1, (8, 3))
            frac = (centres[rng.integers(0, 8, n)] + rng.normal(0, rng.uniform(0.02, 0.2), (n, 3))) % 1.0
            xyz = np.stack([(frac.T * L[:, None] + lo[:, None])] * F) + rng.normal(0, 0.05, (F, 3, n))
        elif trial % 3 == 1:
            g = int(round(n ** (1 / 3))) + 1  # a lattice: many exactly equal distances, d == L/2 hits
            idx = rng.choice(g ** 3, n, replace=False)
            frac = np.stack([idx % g, (idx // g) % g, idx // (g * g)]).astype(np.float64) / g
            xyz = np.stack([frac * L[:, None] + lo[:, None]] * F)
        else:
            xyz = rng.uniform(0, 1, (F, 3, n)) * L[None, :, None] + lo[None, :, None]
        if trial % 4 == 1:
            k = rng.choice(n, 40, replace=False)
            xyz[:, :, k] += rng.integers(-2, 3, (F, 3, 40)) * L[None, :, None]
        r_cut = float(rng.uniform(0.04, 0.499) * L.min())
        bin_size = float(rng.choice([0.05, 0.1, 0.02]))
        nbins = max(1, int(r_cut / bin_size))
        n_types = int(rng.integers(1, 8))
        ty = rng.integers(1, n_types + 1, n).astype(np.int32)
        rel = np.array([[1, 1], [1, n_types], [n_types, n_types]])
        box = np.tile(L, (F, 1))
        per_frame = bool(trial % 2)
        a = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=dense)
        b = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=culled)
        ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
        m = max(8, n // 7)
        sites = np.ascontiguousarray(xyz[:, :, :m] + rng.normal(0, 0.3, (F, 3, m)))
        st = rng.integers(1, 4, m).astype(np.int32)
        rel2 = np.array([[1, 1], [n_types, 3], [1, 2]])
        c = B.rdf_mol_loop(xyz, ty, sites, st, box, rel2, r_cut, bin_size, nbins, per_frame=per_frame, ctx=dense)
        d = B.rdf_mol_loop(xyz, ty, sites, st, box, rel2, r_cut, bin_size, nbins, per_frame=per_frame, ctx=culled)
        ok = ok and np.array_equal(c[0], d[0]) and c[1] == d[1]
        pairs += F * n * (n - 1) // 2
        if oracle:
            full = b[0] if per_frame else None
            for f in range(F if per_frame else 0):
                cf, cp, _ = cref.rdf_pairs(xyz[f], ty, rel, L, r_cut * r_cut, bin_size, nbins)
                ok = ok and np.array_equal(full[f], cf) and np.array_equal(b[1][f], cp)
                rp, _ = cref.rdf_rect(xyz[f], ty, sites[f], st, rel2, L, r_cut * r_cut, bin_size, nbins)
                ok = ok and np.array_equal(d[0][f], rp)
        if not ok:
            print("MISMATCH trial %d n=%d L=%s lo=%s r_cut=%r bin=%r types=%d per_frame=%d" %
                  (trial, n, L, lo, r_cut, bin_size, n_types, per_frame))
            sys.exit(1)
    print("ok: %d trials, %.3g atom pairs, culled == dense everywhere" % (trials, pairs))


if __name__ == "__main__":
    main()
