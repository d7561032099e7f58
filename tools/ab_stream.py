#!/usr/bin/env python
"""tools/ab_stream.py — host-side rate of the streaming reader alone (no GPU work): FrameStream drained as fast as it
delivers, against the one-shot parallel reader, on synthetic dump files of C2 and C3 frame size."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdproptools_amd import io as mio  # noqa: E402
from mdproptools_amd import synth  # noqa: E402
from mdproptools_amd.stream import FrameStream  # noqa: E402


def write(tmp, n, F):
    L = 50.0 * (n / 10_000) ** (1 / 3)
    ty = synth.rdf_types(n)
    xyz = synth.rdf_frames(n, range(F), L, 2)
    for f in range(F):
        with open(os.path.join(tmp, "dump.nvt.%d.dump" % (f * 1000)), "wt") as fh:
            fh.write("ITEM: TIMESTEP\n%d\nITEM: NUMBER OF ATOMS\n%d\nITEM: BOX BOUNDS pp pp pp\n" % (f * 1000, n))
            fh.write(("0.0 %r\n" % L) * 3)
            fh.write("ITEM: ATOMS id type x y z\n")
            np.savetxt(fh, np.column_stack([np.arange(1, n + 1), ty, xyz[f].T]), fmt="%d %d %.6f %.6f %.6f")
    return os.path.join(tmp, "dump.nvt.*.dump")


for n, F in ((100_000, 60),):
    with tempfile.TemporaryDirectory() as tmp:
        pat = write(tmp, n, F)
        for rep, w in enumerate((32, 32, 16, 8, 4, 64)):
            os.environ["MDHIP_STREAM_WORKERS"] = str(w)
            t0 = time.perf_counter()
            fr = list(mio.iter_native_frames(pat, ["id", "type", "x", "y", "z"]))
            t_list = time.perf_counter() - t0
            del fr
            t0 = time.perf_counter()
            st = FrameStream(pat)
            nb = 0
            for b in st:
                nb += len(b)
            t_stream = time.perf_counter() - t0
            print("n=%d F=%d  one-shot %.4f s   stream %.4f s (%d frames, pinned=%s, parse_s(sum)=%.3f, buffer waits %.3f)"
                  % (n, F, t_list, t_stream, nb, st.stats["pinned"], st.stats["parse_s"], st.stats["wait_for_buffer_s"]))
