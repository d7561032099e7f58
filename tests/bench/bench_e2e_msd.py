#!/usr/bin/env python
"""
tests/bench/bench_e2e_msd.py — end-to-end wall time of `Diffusion.get_msd_from_dump` on TEXT dumps, streamed
(text -> page-locked batches -> device-resident trajectory, mdproptools_amd/stream.py) against the
load-everything-first route, on frames of BASELINE C4's atom count.

    python tests/bench/bench_e2e_msd.py [n_atoms] [n_frames] [allatom|com]

The dump files hold `id type xu yu zu` with 16 distinct coordinate bodies cycled over the frames (rendering text is
the slow part of making a synthetic trajectory; parsing cost does not depend on the values). One JSON line.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    kind = sys.argv[3] if len(sys.argv) > 3 else "allatom"
    from mdproptools_amd import io as mio
    from mdproptools_amd.dynamical import diffusion as dm

    per_mol = 10
    kw = dict(msd_type=kind, avg_interval=True, tao_coeff=4)
    if kind == "com":
        kw.update(num_mols=[n // per_mol], num_atoms_per_mol=[per_mol], mass=[12.0, 1.0])
    rng = np.random.default_rng(5)
    L = 100.0
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        base = rng.random((n, 3)) * L
        ids = np.arange(1, n + 1)
        ty = 1 + (ids % 2)
        bodies = []
        for k in range(16):
            xyz = base + 0.05 * k + 0.01 * rng.standard_normal((n, 3))
            rows = np.column_stack([ids, ty, xyz])
            import io as pyio

            buf = pyio.StringIO()
            np.savetxt(buf, rows, fmt="%d %d %.6f %.6f %.6f")
            bodies.append(buf.getvalue())
        for f in range(F):
            with open(os.path.join(tmp, "dump.nvt.%d.dump" % (f * 1000)), "wt") as fh:
                fh.write("ITEM: TIMESTEP\n%d\nITEM: NUMBER OF ATOMS\n%d\nITEM: BOX BOUNDS pp pp pp\n" % (f * 1000, n))
                fh.write(("0.0 %r\n" % L) * 3)
                fh.write("ITEM: ATOMS id type xu yu zu\n")
                fh.write(bodies[f % 16])
        t_write = time.perf_counter() - t0
        text_bytes = sum(os.path.getsize(os.path.join(tmp, x)) for x in os.listdir(tmp))
        d = dm.Diffusion(timestep=1, units="real", outputs_dir=tmp, diff_dir=tmp)
        # the first read of freshly written files pays the page-table population of their mappings: one untimed pass
        list(mio.iter_native_frames(os.path.join(tmp, "dump.nvt.*.dump"), ["id", "type", "xu", "yu", "zu"]))
        d.get_msd_from_dump("dump.nvt.0.dump", **kw)  # library + FFT-free kernels warm
        best = {}
        outs = {}
        for rep in range(2):
            for on in (True, False):
                dm.STREAM = on
                t0 = time.perf_counter()
                out = d.get_msd_from_dump("dump.nvt.*.dump", **kw)
                dt = time.perf_counter() - t0
                if on not in best or dt < best[on]:
                    best[on] = dt
                outs[on] = out
        dm.STREAM = True
        for a, b in zip(outs[True], outs[False]):
            assert np.array_equal(a.to_numpy(), b.to_numpy())
        t0 = time.perf_counter()
        list(mio.iter_native_frames(os.path.join(tmp, "dump.nvt.*.dump"), ["id", "type", "xu", "yu", "zu"]))
        parse_only = time.perf_counter() - t0
    print(json.dumps(dict(
        workload="%d dump files x %d atoms (id type xu yu zu), get_msd_from_dump(msd_type=%r, avg_interval=True)"
                 % (F, n, kind),
        text_MB=text_bytes / 1e6, coordinates_MB=F * n * 24 / 1e6, streamed_s=best[True], load_all_first_s=best[False],
        parse_all_frames_only_s=parse_only, identical_dataframes=True, host_cores=os.cpu_count(),
        synth_write_s=t_write)))


if __name__ == "__main__":
    main()
