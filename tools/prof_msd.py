#!/usr/bin/env python
"""tools/prof_msd.py [n_atoms] [n_frames] [allatom|com] — cProfile of one streamed Diffusion.get_msd_from_dump call on
synthetic text dumps (where the host time of the drop-in goes)."""
import cProfile
import io as pyio
import os
import pstats
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    kind = sys.argv[3] if len(sys.argv) > 3 else "allatom"
    from mdproptools_amd import io as mio
    from mdproptools_amd.dynamical import diffusion as dm

    kw = dict(msd_type=kind, avg_interval=True, tao_coeff=4)
    if kind == "com":
        kw.update(num_mols=[n // 10], num_atoms_per_mol=[10], mass=[12.0, 1.0])
    rng = np.random.default_rng(5)
    with tempfile.TemporaryDirectory() as tmp:
        ids = np.arange(1, n + 1)
        buf = pyio.StringIO()
        np.savetxt(buf, np.column_stack([ids, 1 + ids % 2, rng.random((n, 3)) * 100]), fmt="%d %d %.6f %.6f %.6f")
        body = buf.getvalue()
        for f in range(F):
            with open(os.path.join(tmp, "dump.nvt.%d.dump" % (f * 1000)), "wt") as fh:
                fh.write("ITEM: TIMESTEP\n%d\nITEM: NUMBER OF ATOMS\n%d\nITEM: BOX BOUNDS pp pp pp\n" % (f * 1000, n))
                fh.write("0.0 100.0\n" * 3 + "ITEM: ATOMS id type xu yu zu\n" + body)
        d = dm.Diffusion(timestep=1, units="real", outputs_dir=tmp, diff_dir=tmp)
        list(mio.iter_native_frames(os.path.join(tmp, "dump.nvt.*.dump"), ["id", "type", "xu", "yu", "zu"]))
        d.get_msd_from_dump("dump.nvt.*.dump", **kw)
        pr = cProfile.Profile()
        pr.enable()
        d.get_msd_from_dump("dump.nvt.*.dump", **kw)
        pr.disable()
        st = pstats.Stats(pr, stream=sys.stdout)
        st.sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
