"""
CPU, world_size 2, gloo: the frame-sharding and collective logic of mdproptools_amd.dist with the
oracle standing in for the GPU kernels. Results must equal the single-process oracle exactly for
integer work, and to rounding for the gathered floating-point rows.
"""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_case():
    rng = np.random.default_rng(99)
    F, n = 5, 300  # odd frame count: shards of 3 and 2
    L = np.array([14.0, 15.0, 16.0])
    xyz = rng.uniform(0, 1, (F, 3, n)) * L[None, :, None]
    ty = rng.integers(1, 4, n).astype(np.int32)
    rel = np.array([[1, 1], [1, 2], [2, 3], [3, 3]])
    box = np.tile(L, (F, 1)) * (1 + 0.01 * np.arange(F))[:, None]
    r = np.cumsum(rng.normal(0, 0.1, (F, 3, n)), axis=0)
    return F, n, L, xyz, ty, rel, box, r


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from mdproptools_amd import dist as D
    from oracle import cref

    dist.init_process_group("gloo", rank=rank, world_size=world)
    F, n, L, xyz, ty, rel, box, r = _make_case()
    lo, hi = D.frame_shard(F)

    def rdf_sum(x, t, b, rl, rc, dd, nb):
        full = np.zeros(nb, np.uint64)
        part = np.zeros((len(rl), nb), np.uint64)
        ov = 0
        for f in range(len(x)):
            a, p, o = cref.rdf_pairs(x[f], t, rl, b[f], rc * rc, dd, nb)
            full, part, ov = full + a, part + p, ov + o
        return full, part, ov

    def rdf_frames(x, t, b, rl, rc, dd, nb):
        res = [cref.rdf_pairs(x[f], t, rl, b[f], rc * rc, dd, nb) for f in range(len(x))]
        return (np.stack([q[0] for q in res]).reshape(len(x), nb),
                np.stack([q[1] for q in res]).reshape(len(x), len(rl), nb), sum(q[2] for q in res))

    def cn_sum(x, t, b, rl, cuts):
        return sum(cref.cn_pairs(x[f], t, rl, b[f], [c * c for c in cuts]) for f in range(len(x)))

    def msd(rr, pairs, goff, sc):
        return cref.msd_pairs(np.asarray(rr) * sc, pairs, goff)

    full, part, ov = D.rdf_sharded(xyz[lo:hi], ty, box[lo:hi], rel, 6.0, 0.05, 120, compute=rdf_sum)
    pf, pp, _ = D.rdf_sharded_per_frame(xyz[lo:hi], ty, box[lo:hi], rel, 6.0, 0.05, 120, F, compute=rdf_frames)
    cn = D.cn_sharded(xyz[lo:hi], ty, box[lo:hi], rel, [2.0, 3.0, 4.0, 5.5], compute=cn_sum)
    sums = D.msd_single_origin_sharded(r[lo:hi], F, [0, 100, n], scale=1e-10, origin_frame=0, compute=msd)
    sums4 = D.msd_single_origin_sharded(r[lo:hi], F, [0, 100, n], scale=1e-10, origin_frame=4, compute=msd)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), full=full, part=part, ov=ov, pf=pf, pp=pp, cn=cn,
             sums=sums, sums4=sums4, shard=np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_shard_partition():
    from mdproptools_amd.dist import frame_shard

    for F in (0, 1, 2, 7, 8, 200, 1001):
        for world in (1, 2, 3, 8):
            blocks = [frame_shard(F, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == F
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_paths_world2_gloo(tmp_path):
    import torch.multiprocessing as mp

    from oracle import cref

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    F, n, L, xyz, ty, rel, box, r = _make_case()
    full = np.zeros(120, np.uint64)
    part = np.zeros((4, 120), np.uint64)
    per = []
    for f in range(F):
        a, p, o = cref.rdf_pairs(xyz[f], ty, rel, box[f], 36.0, 0.05, 120)
        full, part = full + a, part + p
        per.append((a, p))
    cn = sum(cref.cn_pairs(xyz[f], ty, rel, box[f], [4.0, 9.0, 16.0, 30.25]) for f in range(F))
    sums = cref.msd_pairs(r * 1e-10, [(0, t) for t in range(F)], [0, 100, n])
    sums4 = cref.msd_pairs(r * 1e-10, [(4, t) for t in range(F)], [0, 100, n])
    shards = []
    for rank in range(2):
        g = np.load(tmp_path / ("rank%d.npz" % rank))
        shards.append(tuple(g["shard"]))
        np.testing.assert_array_equal(g["full"], full)
        np.testing.assert_array_equal(g["part"], part)
        np.testing.assert_array_equal(g["pf"], np.stack([q[0] for q in per]))
        np.testing.assert_array_equal(g["pp"], np.stack([q[1] for q in per]))
        np.testing.assert_array_equal(g["cn"], cn)
        np.testing.assert_allclose(g["sums"], sums, rtol=1e-14)
        np.testing.assert_allclose(g["sums4"], sums4, rtol=1e-14)
    assert shards == [(0, 3), (3, 5)]
