"""
The streaming ingest layer (mdproptools_amd/stream.py) on CPU: batches come in file order with the same doubles as
the one-shot reader, split where the atom count changes or the byte budget is reached, with a bounded ring of staging
buffers that are handed back after use; a reader error reaches the consumer. (Without a HIP runtime the staging
buffers are plain numpy memory; on the GPU box they are page-locked: tests/test_gpu_dropin.py.)
"""
import numpy as np
import pytest

from mdproptools_amd import io as mio
from mdproptools_amd.stream import FrameStream


def _write_traj(tmp_path, sizes):
    rng = np.random.default_rng(4)
    tables = []
    for k, n in enumerate(sizes):
        tbl = np.column_stack([rng.permutation(n) + 1, 1 + np.arange(n) % 3, np.round(rng.uniform(0, 9, (n, 3)), 5)])
        mio.write_dump(str(tmp_path / ("dump.nvt.%d.dump" % (k * 50))), k * 50, [[0, 9 + 0.1 * k]] * 3,
                       ["id", "type", "x", "y", "z"], tbl)
        tables.append(tbl[np.argsort(tbl[:, 0])])
    return str(tmp_path / "dump.nvt.*.dump"), tables


def test_stream_batches_equal_one_shot_reader(tmp_path):
    sizes = [40, 40, 40, 40, 40, 25, 25, 40, 40, 40, 40]
    pattern, tables = _write_traj(tmp_path, sizes)
    st = FrameStream(pattern, batch_bytes=3 * 24 * 40, depth=2)  # 3 frames of 40 atoms per batch
    seen, shapes = [], []
    for batch in st:
        shapes.append(batch.xyz.shape)
        for fr in batch:
            seen.append((fr.timestep, fr.ids.copy(), fr.types.copy(), fr.xyz.copy(), fr.lengths))
    assert shapes == [(3, 3, 40), (2, 3, 40), (2, 3, 25), (3, 3, 40), (1, 3, 40)]
    assert [s[0] for s in seen] == [50 * k for k in range(len(sizes))]
    for (ts, ids, types, xyz, lengths), tbl, k in zip(seen, tables, range(len(sizes))):
        np.testing.assert_array_equal(ids, tbl[:, 0])
        np.testing.assert_array_equal(types, tbl[:, 1])
        np.testing.assert_array_equal(xyz, tbl[:, 2:5].T)
        assert lengths == pytest.approx((9 + 0.1 * k,) * 3, rel=1e-15)
    assert st.stats["frames"] == len(sizes) and st.stats["batches"] == 5
    assert len(st._bufs) == 0  # the ring was freed at the end


def test_stream_ring_is_bounded_and_reused(tmp_path):
    pattern, _ = _write_traj(tmp_path, [30] * 12)
    st = FrameStream(pattern, batch_bytes=24 * 30, depth=2)  # one frame per batch, two buffers
    addrs = set()
    for batch in st:
        addrs.add(batch.xyz.__array_interface__["data"][0])
        assert len(st._bufs) <= 2
    assert len(addrs) <= 2


def test_stream_explicit_files_and_errors(tmp_path):
    pattern, tables = _write_traj(tmp_path, [20, 20, 20, 20])
    files = mio._sorted_matches(pattern)[1:3]  # a rank's share of the files
    got = [fr.timestep for b in FrameStream(pattern, files=files) for fr in b]
    assert got == [50, 100]
    bad = tmp_path / "dump.nvt.75.dump"
    bad.write_text("ITEM: TIMESTEP\n75\nITEM: NUMBER OF ATOMS\n2\nITEM: BOX BOUNDS pp pp pp\n0 1\n0 1\n0 1\n"
                   "ITEM: ATOMS id type x y z\n1 1 0.5 0.5\n2 1 0.1 0.2 0.3\n")
    with pytest.raises(ValueError, match="fewer values"):
        for _b in FrameStream(pattern):
            pass
    # a consumer that stops early does not leave the producer thread behind
    st = FrameStream(str(tmp_path / "dump.nvt.[0-9]0.dump"), batch_bytes=24 * 20, depth=2)
    for _b in st:
        break
    st.close()
    assert not st._thread.is_alive()


def test_stream_multi_frame_files(tmp_path):
    """Files that hold several frames: their frames are parsed concurrently (the file's mapping is closed by whichever
    parse finishes last) and still come out in order, across batch boundaries."""
    rng = np.random.default_rng(6)
    tables, k = [], 0
    for fidx in range(3):
        text = []
        for _ in range(4):  # four frames per file
            n = 35
            tbl = np.column_stack([rng.permutation(n) + 1, 1 + np.arange(n) % 2, np.round(rng.uniform(0, 8, (n, 3)), 5)])
            p = tmp_path / "one.dump"
            mio.write_dump(str(p), k * 10, [[0, 8]] * 3, ["id", "type", "x", "y", "z"], tbl)
            text.append(p.read_text())
            tables.append(tbl[np.argsort(tbl[:, 0])])
            k += 1
        (tmp_path / ("traj.%d.dump" % fidx)).write_text("".join(text))
    (tmp_path / "one.dump").unlink()
    got = []
    for batch in FrameStream(str(tmp_path / "traj.*.dump"), batch_bytes=5 * 24 * 35, depth=2):
        for fr in batch:
            got.append((fr.timestep, fr.xyz.copy()))
    assert [g[0] for g in got] == [10 * i for i in range(12)]
    for (ts, xyz), tbl in zip(got, tables):
        np.testing.assert_array_equal(xyz, tbl[:, 2:5].T)
