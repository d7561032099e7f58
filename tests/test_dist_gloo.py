"""
CPU, world_size 2 / 3 / 8, gloo: the frame-sharding and collective logic of mdproptools_amd.dist with the
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


def _walk(F, n):
    return np.cumsum(np.random.default_rng(1000 + F).normal(0, 0.1, (F, 3, n)), axis=0)


def _np_windows(r, scale):
    """mdhip_msd_windows(r, tao=1) in numpy: per entity, the sums over consecutive frames of the squared steps
    (diffusion.py:225-237)."""
    d = np.diff(np.asarray(r) * scale, axis=0) ** 2  # [F-1,3,E]
    s = d.sum(axis=0)
    return np.column_stack([s[0], s[1], s[2], ((d[:, 0] + d[:, 1]) + d[:, 2]).sum(axis=0)])


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
        if len(x) == 0:  # (a rank without frames, F < world: what backend.rdf_loop returns for F = 0)
            return np.zeros((0, nb), np.uint64), np.zeros((0, len(rl), nb), np.uint64), 0
        res = [cref.rdf_pairs(x[f], t, rl, b[f], rc * rc, dd, nb) for f in range(len(x))]
        return (np.stack([q[0] for q in res]).reshape(len(x), nb),
                np.stack([q[1] for q in res]).reshape(len(x), len(rl), nb), sum(q[2] for q in res))

    def cn_sum(x, t, b, rl, cuts):
        return sum((cref.cn_pairs(x[f], t, rl, b[f], [c * c for c in cuts]) for f in range(len(x))),
                   np.zeros(len(rl), np.uint64))

    def msd(rr, pairs, goff, sc):
        return cref.msd_pairs(np.asarray(rr) * sc, pairs, goff)

    full, part, ov = D.rdf_sharded(xyz[lo:hi], ty, box[lo:hi], rel, 6.0, 0.05, 120, compute=rdf_sum)
    pending = D.rdf_sharded_async(xyz[lo:hi], ty, box[lo:hi], rel, 6.0, 0.05, 120, compute=rdf_sum)
    afull, apart, aov = pending.wait()  # the in-flight variant gives the same sums
    assert np.array_equal(afull, full) and np.array_equal(apart, part) and int(aov[0]) == ov
    pf, pp, pov = D.rdf_sharded_per_frame(xyz[lo:hi], ty, box[lo:hi], rel, 6.0, 0.05, 120, F, compute=rdf_frames)
    assert pov == ov  # (one collective: the overflow count rides in the packed rows' last column)
    cn = D.cn_sharded(xyz[lo:hi], ty, box[lo:hi], rel, [2.0, 3.0, 4.0, 5.5], compute=cn_sum)
    sums = D.msd_single_origin_sharded(r[lo:hi], F, [0, 100, n], scale=1e-10, origin_frame=0, compute=msd)
    sums4 = D.msd_single_origin_sharded(r[lo:hi], F, [0, 100, n], scale=1e-10, origin_frame=4, compute=msd)
    # compute-bound paths: full-lag MSD sharded by entities (groups straddle the rank boundary: 100 | 200 entities
    # against slices of 150 + 150), direct ACF sharded by lag ranges of equal work
    e_lo, e_hi = D.entity_shard(n)

    def lag(rr, ml, goff, sc):
        return cref.lag_msd(np.asarray(rr) * sc, np.arange(ml + 1), goff)

    lagm = D.lag_msd_sharded(r[:, :, e_lo:e_hi], (e_lo, e_hi), F - 1, [0, 100, n], scale=2.0, compute=lag)
    series = np.cumsum(np.random.default_rng(5).normal(size=(2, 400)), axis=1)

    def xc(aa, bb, k0, nl):
        bb = aa if bb is None else bb
        return np.stack([cref.xcorr_direct(aa[p], bb[p], n_lags=k0 + nl)[k0:] for p in range(len(aa))])

    acf = D.xcorr_direct_sharded(series, None, compute=xc)
    ccf = D.xcorr_direct_sharded(series[0], series[1], n_lags=300,
                                 compute=lambda aa, bb, k0, nl: cref.xcorr_direct(aa, bb, n_lags=k0 + nl)[k0:])
    # fixed-lag windows by frames with a one-frame halo: tao 2 over shards (3, 2) and tao 3 / 5 over 11 frames (6, 5)
    wins = {}
    for F2, tao in ((F, 2), (11, 3), (11, 5), (11, 7), (2, 1)):
        rw = _walk(F2, n)
        l2, h2 = D.frame_shard(F2)
        wins["win_%d_%d" % (F2, tao)] = D.msd_windows_sharded(rw[l2:h2], F2, tao, scale=3.0, compute=_np_windows)
    # the fused MSD step (one all-gather before the kernels, ONE all-reduce after them): 11 frames in shards of 6 + 5,
    # the origin in either shard, tao 1 / 3 / 7 (7: the upper shard's only kept frame reaches back across the boundary)
    steps = {}
    F3 = 11
    rs = _walk(F3, n)
    l3, h3 = D.frame_shard(F3)
    stand_in = {
        "origin": lambda rr, r0, goff, sc: cref.msd_pairs(np.concatenate([r0[None], rr]) * sc,
                                                          [(0, 1 + t) for t in range(len(rr))], goff),
        "windows": lambda rr, tao, sc: _np_windows(rr[::tao], sc),
        "lag": lambda x, ml, goff, sc: cref.lag_msd(np.asarray(x) * sc, np.arange(ml + 1), goff),
    }
    for origin, tao in ((0, 1), (7, 3), (2, 7)):
        a, b, c, _st = D.msd_step_sharded(rs[l3:h3], rs[:, :, e_lo:e_hi], F3, (e_lo, e_hi), [0, 100, n], tao, scale=1e-10,
                                          lag_scale=2.0, origin_frame=origin, compute=stand_in)
        steps["step_%d_%d_single" % (origin, tao)] = a
        steps["step_%d_%d_win" % (origin, tao)] = b
        steps["step_%d_%d_lag" % (origin, tao)] = c
    # fewer frames than ranks at world 8 (ranks without a frame contribute zeros): 5 frames, origin in the last
    # non-empty rank, tao 2 (kept frames 0, 2, 4: every window crosses a rank boundary at world >= 3)
    F4 = 5
    r4 = _walk(F4, n)
    l4, h4 = D.frame_shard(F4)
    a, b, c, _st = D.msd_step_sharded(r4[l4:h4], r4[:, :, e_lo:e_hi], F4, (e_lo, e_hi), [0, 100, n], 2, scale=1e-10,
                                      lag_scale=2.0, origin_frame=F4 - 1, compute=stand_in)
    steps["step5_single"], steps["step5_win"], steps["step5_lag"] = a, b, c
    sums_last = D.msd_single_origin_sharded(r[lo:hi], F, [0, 100, n], scale=1e-10, origin_frame=F - 1, compute=msd)
    # per-frame charge flux gathered in frame order (the reference's only frame-parallel gather, conductivity.py:190-194)
    vel = np.random.default_rng(77).normal(0, 1e-3, (F, 3, n))
    seg_off = np.arange(0, n + 1, 3)
    seg_type = np.arange(len(seg_off) - 1) % 2
    q_atom = np.tile([0.5, -1.0, 0.25], n // 3)
    flux_fn = _numpy_dynamical_backend()[4]
    flux = D.charge_flux_sharded(vel[lo:hi], F, np.ones(n), q_atom, seg_off, seg_type, 2, 1e5, 1.6e-19, compute=flux_fn)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), full=full, part=part, ov=ov, pf=pf, pp=pp, cn=cn,
             sums=sums, sums4=sums4, sums_last=sums_last, flux=flux, shard=np.array([lo, hi]), lagm=lagm, acf=acf, ccf=ccf,
             eshard=np.array([e_lo, e_hi]), **wins, **steps)
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


def test_lag_ranges_balance_the_work():
    from mdproptools_amd.dist import lag_ranges

    for n, n_lags, world in ((1000, 1000, 2), (1000, 1000, 8), (10 ** 6, 10 ** 6, 8), (500, 37, 3), (10, 10, 16)):
        b = lag_ranges(n, n_lags, world)
        assert b[0] == 0 and b[-1] == n_lags and len(b) == world + 1 and all(x <= y for x, y in zip(b, b[1:]))
        work = [sum(n - k for k in range(b[r], b[r + 1])) if n <= 1000 else
                (b[r + 1] - b[r]) * (2 * n - b[r] - b[r + 1] + 1) / 2 for r in range(world)]
        if n_lags >= 100 * world:
            assert max(work) <= 1.02 * sum(work) / world


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_paths_gloo(tmp_path, world):
    """Every sharded path at 2, 3 and 8 ranks (the target machine has 8): frame counts that do not divide
    (5 and 11 frames), FEWER frames than ranks at 8 (ranks without a frame contribute zeros), the origin in the last
    non-empty rank, tao values that do not divide the shards, groups that straddle several entity-shard boundaries."""
    import torch.multiprocessing as mp

    from mdproptools_amd.dist import entity_shard, frame_shard
    from oracle import cref

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
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
    lagm = cref.lag_msd(r * 2.0, np.arange(F), [0, 100, n])
    series = np.cumsum(np.random.default_rng(5).normal(size=(2, 400)), axis=1)
    acf = np.stack([cref.xcorr_direct(series[p], series[p]) for p in range(2)])
    ccf = cref.xcorr_direct(series[0], series[1], n_lags=300)
    shards = []
    sums_last = cref.msd_pairs(r * 1e-10, [(F - 1, t) for t in range(F)], [0, 100, n])
    vel = np.random.default_rng(77).normal(0, 1e-3, (F, 3, n))
    seg_off = np.arange(0, n + 1, 3)
    flux = _numpy_dynamical_backend()[4](vel, np.ones(n), np.tile([0.5, -1.0, 0.25], n // 3), seg_off,
                                          np.arange(len(seg_off) - 1) % 2, 2, 1e5, 1.6e-19)
    r4 = _walk(5, n)
    for rank in range(world):
        g = np.load(tmp_path / ("rank%d.npz" % rank))
        shards.append(tuple(g["shard"]))
        assert tuple(g["eshard"]) == entity_shard(n, rank, world)
        np.testing.assert_allclose(g["sums_last"], sums_last, rtol=1e-14)
        # (every frame's vector comes from exactly one rank; numpy's stand-in sums a shard's block in an order that depends
        # on the block's shape, hence to rounding — the device kernel's per-frame order does not: tests/test_gpu_fullsize.py)
        np.testing.assert_allclose(g["flux"], flux, rtol=1e-12, atol=1e-40)
        np.testing.assert_allclose(g["step5_single"], cref.msd_pairs(r4 * 1e-10, [(4, t) for t in range(5)], [0, 100, n]),
                                   rtol=1e-14)
        np.testing.assert_allclose(g["step5_win"], _np_windows(r4[::2], 1e-10), rtol=1e-13)
        np.testing.assert_allclose(g["step5_lag"], cref.lag_msd(r4 * 2.0, np.arange(5), [0, 100, n]), rtol=1e-12)
        np.testing.assert_allclose(g["lagm"], lagm, rtol=1e-12)
        np.testing.assert_array_equal(g["acf"], acf)  # every lag comes from exactly one rank: identical
        np.testing.assert_array_equal(g["ccf"], ccf)
        np.testing.assert_array_equal(g["full"], full)
        np.testing.assert_array_equal(g["part"], part)
        np.testing.assert_array_equal(g["pf"], np.stack([q[0] for q in per]))
        np.testing.assert_array_equal(g["pp"], np.stack([q[1] for q in per]))
        np.testing.assert_array_equal(g["cn"], cn)
        np.testing.assert_allclose(g["sums"], sums, rtol=1e-14)
        np.testing.assert_allclose(g["sums4"], sums4, rtol=1e-14)
        for F2, tao in ((F, 2), (11, 3), (11, 5), (11, 7), (2, 1)):
            np.testing.assert_allclose(g["win_%d_%d" % (F2, tao)], _np_windows(_walk(F2, n)[::tao], 3.0), rtol=1e-13)
        rs = _walk(11, n)
        for origin, tao in ((0, 1), (7, 3), (2, 7)):
            key = "step_%d_%d_" % (origin, tao)
            np.testing.assert_allclose(g[key + "single"], cref.msd_pairs(rs * 1e-10, [(origin, t) for t in range(11)],
                                                                         [0, 100, n]), rtol=1e-14)
            np.testing.assert_allclose(g[key + "win"], _np_windows(rs[::tao], 1e-10), rtol=1e-13)
            np.testing.assert_allclose(g[key + "lag"], cref.lag_msd(rs * 2.0, np.arange(11), [0, 100, n]), rtol=1e-12)
    assert shards == [frame_shard(F, rank, world) for rank in range(world)]
    if world == 2:
        assert shards == [(0, 3), (3, 5)]
    if world == 8:
        assert shards[5:] == [(5, 5)] * 3  # three ranks hold no frame of the 5


def _post_group_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MDHIP_STEP_POST_GROUP="1")
    import torch.distributed as dist

    from mdproptools_amd import dist as D
    from oracle import cref

    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, F = 120, 9
    stand_in = {
        "origin": lambda rr, r0, goff, sc: cref.msd_pairs(np.concatenate([r0[None], rr]) * sc,
                                                          [(0, 1 + t) for t in range(len(rr))], goff),
        "windows": lambda rr, tao, sc: _np_windows(rr[::tao], sc),
        "lag": lambda x, ml, goff, sc: cref.lag_msd(np.asarray(x) * sc, np.arange(ml + 1), goff),
    }
    lo, hi = D.frame_shard(F)
    e_lo, e_hi = D.entity_shard(n)
    walks = [np.cumsum(np.random.default_rng(300 + k).normal(0, 0.1, (F, 3, n)), axis=0) for k in range(4)]
    # a pipeline as bench.py --workload c4 runs it: step k + 1 is ISSUED (its all-gather on the default communicator)
    # before step k is waited for (its all-reduce on the second communicator)
    inflight, out = [], []
    for k, rw in enumerate(walks):
        inflight.append(D.msd_step_sharded_async(rw[lo:hi], rw[:, :, e_lo:e_hi], F, (e_lo, e_hi), [0, 50, n], 2,
                                                 scale=1e-10, lag_scale=2.0, origin_frame=k, compute=stand_in))
        if len(inflight) > 1:
            out.append(inflight.pop(0).wait())
    out.append(inflight.pop(0).wait())
    assert len(D._POST_GROUP) == 1  # the second communicator exists and was used
    np.savez(os.path.join(out_dir, "post%d.npz" % rank), **{"s%d_%d" % (k, j): o[j] for k, o in enumerate(out) for j in range(3)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_fused_step_with_the_second_communicator_gloo(tmp_path, world):
    """MDHIP_STEP_POST_GROUP=1 (the opt-in that lets step k + 1's all-gather overtake step k's all-reduce): the closing
    all-reduce of every step runs on a second communicator while the next step's pre-exchange is already issued on the
    default one — four pipelined steps, results equal to the oracle's on every rank."""
    import torch.multiprocessing as mp

    from oracle import cref

    mp.spawn(_post_group_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    n, F = 120, 9
    for k in range(4):
        rw = np.cumsum(np.random.default_rng(300 + k).normal(0, 0.1, (F, 3, n)), axis=0)
        for rank in range(world):
            g = np.load(tmp_path / ("post%d.npz" % rank))
            np.testing.assert_allclose(g["s%d_0" % k], cref.msd_pairs(rw * 1e-10, [(k, t) for t in range(F)], [0, 50, n]),
                                       rtol=1e-14)
            np.testing.assert_allclose(g["s%d_1" % k], _np_windows(rw[::2], 1e-10), rtol=1e-13)
            np.testing.assert_allclose(g["s%d_2" % k], cref.lag_msd(rw * 2.0, np.arange(F), [0, 50, n]), rtol=1e-12)


# ------------------------------------------------------------------ drop-in functions under torch.distributed
def _oracle_backend():
    """The four pair loops of mdproptools_amd.backend with the C oracle behind them (CPU stand-in)."""
    from oracle import cref

    def rdf_loop(xyz, types, box, rel, r_cut, ddr, nbins, per_frame=True, ctx=None, edges=None):
        res = [cref.rdf_pairs(xyz[f], types if np.ndim(types) == 1 else types[f], rel, box[f], float(r_cut) ** 2,
                              ddr, nbins) for f in range(len(xyz))]
        return (np.stack([q[0] for q in res]), np.stack([q[1] for q in res]), sum(q[2] for q in res))

    def cn_loop(xyz, types, box, rel, cuts, per_frame=True, ctx=None):
        return np.stack([cref.cn_pairs(xyz[f], types if np.ndim(types) == 1 else types[f], rel, box[f],
                                       [float(c) ** 2 for c in cuts]) for f in range(len(xyz))])

    return rdf_loop, cn_loop


def _dropin_case(tmp_dir, n_files):
    from mdproptools_amd import io as mio

    rng = np.random.default_rng(5)
    n = 240
    paths = []
    for k in range(n_files):
        L = 13.0 + 0.1 * k
        tbl = np.column_stack([rng.permutation(n) + 1, 1 + (np.arange(n) % 3), rng.uniform(0, L, (n, 3))])
        p = os.path.join(tmp_dir, "dump.nvt.%d.dump" % (k * 100))
        mio.write_dump(p, k * 100, [[0, L]] * 3, ["id", "type", "x", "y", "z"], tbl)
        paths.append(p)
    return os.path.join(tmp_dir, "dump.nvt.*.dump")


def _dropin_worker(rank, world, port, tmp_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from mdproptools_amd import backend
    from mdproptools_amd.structural import rdf_cn

    backend.rdf_loop, backend.cn_loop = _oracle_backend()
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pattern = os.path.join(tmp_dir, "dump.nvt.*.dump")
    out = os.path.join(tmp_dir, "w%d" % world)
    os.makedirs(out, exist_ok=True)
    g = rdf_cn.calc_atomic_rdf(5.0, 0.1, 3, [1.0, 2.0, 3.0], [[1, 1, 2], [1, 2, 3]], pattern,
                               path_or_buff=os.path.join(out, "rdf.csv"))
    c = rdf_cn.calc_atomic_cn([2.0, 3.0, 4.5], 0.1, 3, [1.0, 2.0, 3.0], [[1, 1, 2], [1, 2, 3]], pattern,
                              path_or_buff=os.path.join(out, "cn.csv"))
    np.savez(os.path.join(out, "rank%d.npz" % rank), g=g.to_numpy(), c=c.to_numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_dropin_rdf_cn_sharded_over_files_world2_gloo(tmp_path):
    """calc_atomic_rdf / calc_atomic_cn under torch.distributed: ranks parse their own share of the dump files,
    per-frame g(r) rows are all-gathered and summed in frame order -> bit for bit the single-process result on
    every rank; only rank 0 writes the CSV."""
    import torch.multiprocessing as mp

    _dropin_case(str(tmp_path), 5)
    mp.spawn(_dropin_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_dropin_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    one = np.load(tmp_path / "w1" / "rank0.npz")
    assert np.isfinite(one["g"]).all() and one["g"][:, 1].sum() > 0
    for rank in range(2):
        two = np.load(tmp_path / "w2" / ("rank%d.npz" % rank))
        np.testing.assert_array_equal(two["g"], one["g"])
        np.testing.assert_array_equal(two["c"], one["c"])
    assert (tmp_path / "w2" / "rdf.csv").exists() and (tmp_path / "w2" / "cn.csv").exists()
    assert open(tmp_path / "w2" / "rdf.csv").read() == open(tmp_path / "w1" / "rdf.csv").read()


def test_dropin_rdf_cn_fewer_frames_than_ranks_gloo(tmp_path):
    """Two dump files, three ranks: the trajectory is not split by files (every rank parses both and takes its share of
    the frames), rank 2 holds no frame and contributes no rows — the result is the single-process one on every rank."""
    import torch.multiprocessing as mp

    _dropin_case(str(tmp_path), 2)
    mp.spawn(_dropin_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_dropin_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    one = np.load(tmp_path / "w1" / "rank0.npz")
    for rank in range(3):
        three = np.load(tmp_path / "w3" / ("rank%d.npz" % rank))
        np.testing.assert_array_equal(three["g"], one["g"])
        np.testing.assert_array_equal(three["c"], one["c"])


def _numpy_dynamical_backend():
    """segment_com / msd_pairs / msd_windows / charge_flux of mdproptools_amd.backend as plain numpy (CPU
    stand-ins with the same signatures; only their sharding around them is under test here)."""

    def segment_com(attr, atom_mass, seg_off, atom_q=None, out=None, ctx=None):
        off = np.asarray(seg_off[:-1], dtype=np.int64)
        mass = np.asarray(atom_mass, dtype=np.float64)
        seg_mass = np.add.reduceat(mass, off)
        com = np.add.reduceat(np.asarray(attr) * mass, off, axis=2) / seg_mass
        return com, seg_mass, None if atom_q is None else np.add.reduceat(np.asarray(atom_q), off)

    def msd_pairs(r, pairs, group_off, scale=1.0, per_entity=False, ctx=None):
        r = np.asarray(r) * scale
        pe = np.stack([np.concatenate([(r[b] - r[a]) ** 2, ((r[b] - r[a]) ** 2).sum(axis=0)[None]]).T
                       for a, b in np.asarray(pairs)])  # [P,E,4]
        go = np.asarray(group_off)
        sums = np.stack([pe[:, go[g]:go[g + 1]].sum(axis=1) for g in range(len(go) - 1)], axis=1)
        return (sums, pe) if per_entity else sums

    def msd_pairs_cols(r, pairs, group_off, cols, scale=1.0, ctx=None):
        sums, pe = msd_pairs(r, pairs, group_off, scale=scale, per_entity=True)
        cols[:] = pe.reshape(-1, 4).T
        return sums

    def msd_origin(r, origin, group_off, scale=1.0, cols=None, out=None, ctx=None):
        stacked = np.concatenate([np.asarray(origin)[None], np.asarray(r)])
        pairs = [(0, 1 + t) for t in range(len(r))]
        sums, pe = msd_pairs(stacked, pairs, group_off, scale=scale, per_entity=True)
        if cols is not None:
            cols[:] = pe.reshape(-1, 4).T
        return sums

    def msd_windows(r, tao, scale=1.0, ctx=None, out=None):
        kept = (np.asarray(r) * scale)[::tao]
        d2 = (kept[1:] - kept[:-1]) ** 2  # [W,3,E]
        return np.concatenate([d2.sum(axis=0), d2.sum(axis=(0, 1))[None]]).T

    def charge_flux(vel, atom_mass, atom_q, seg_off, seg_type, n_types, vel_conv, charge_conv, ctx=None):
        com, _, q = segment_com(vel, atom_mass, seg_off, atom_q=atom_q)
        jm = (com * vel_conv) * (q * charge_conv)  # [F,3,M]
        return np.stack([np.stack([jm[:, k, np.asarray(seg_type) == t].sum(axis=1) for t in range(n_types)])
                         for k in range(3)])

    return segment_com, msd_pairs, msd_pairs_cols, msd_windows, charge_flux, msd_origin


def _dynamical_worker(rank, world, port, tmp_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from mdproptools_amd import backend
    from mdproptools_amd.dynamical.conductivity import Conductivity
    from mdproptools_amd.dynamical.diffusion import Diffusion

    (backend.segment_com, backend.msd_pairs, backend.msd_pairs_cols, backend.msd_windows,
     backend.charge_flux, backend.msd_origin) = _numpy_dynamical_backend()
    from mdproptools_amd.dynamical import diffusion as dm

    dm.STREAM = False  # the streamed route keeps the trajectory on the GPU; its two-rank run is tests/test_gpu_dropin.py
    from mdproptools_amd.dynamical import conductivity as cm

    cm.STREAM = True  # (host-side batches through the stand-in flux function: runs on the CPU too)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    out = os.path.join(tmp_dir, "w%d" % world)
    os.makedirs(out, exist_ok=True)
    d = Diffusion(timestep=1, units="real", outputs_dir=tmp_dir, diff_dir=out)
    msd, msd_all, msd_int = d.get_msd_from_dump("dyn.*.dump", msd_type="com", num_mols=[20, 10],
                                                num_atoms_per_mol=[3, 2], mass=[1.0, 12.0], com_drift=True,
                                                avg_interval=True, tao_coeff=2)
    aa, _ = d.get_msd_from_dump("dyn.*.dump", msd_type="allatom")
    c = Conductivity("dyn.*.dump", [20, 10], [3, 2], 1000.0, mass=[1.0, 12.0], temp=300.0, timestep=1,
                     units="real", working_dir=tmp_dir)
    j = c.get_charge_flux()
    np.savez(os.path.join(out, "rank%d.npz" % rank), msd=msd.to_numpy(), msd_all=msd_all.to_numpy(),
             msd_int=msd_int.to_numpy(), aa=aa.to_numpy(), j=j, time=np.asarray(c.time))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_dropin_diffusion_conductivity_sharded_parse_world2_gloo(tmp_path):
    """Diffusion.get_msd_from_dump and Conductivity.get_charge_flux under torch.distributed: ranks parse their own
    share of the files and REDUCE their own frames (MSD: origin frame broadcast, per-frame sums and per-entity columns
    gathered, fixed-lag windows through a one-frame halo and an all-reduce; flux: per-frame vectors gathered) -> the
    single-process result on every rank (msd_int to rounding: its windows are summed rank by rank)."""
    import torch.multiprocessing as mp

    from mdproptools_amd import io as mio

    rng = np.random.default_rng(8)
    n = 20 * 3 + 10 * 2
    x0 = rng.uniform(0, 20, (n, 3))
    cols = ["id", "type", "q", "xu", "yu", "zu", "vx", "vy", "vz"]
    ty = np.concatenate([np.tile([1, 2, 1], 20), np.tile([2, 2], 10)])
    q = np.concatenate([np.tile([0.5, -1.0, 0.5], 20), np.tile([1.0, 0.0], 10)])
    for k in range(7):
        x0 = x0 + rng.normal(0, 0.2, (n, 3))
        perm = rng.permutation(n)
        tbl = np.column_stack([np.arange(1, n + 1), ty, q, x0, rng.normal(0, 1e-3, (n, 3))])[perm]
        mio.write_dump(str(tmp_path / ("dyn.%d.dump" % (k * 50))), k * 50, [[0, 20.0]] * 3, cols, tbl)
    mp.spawn(_dynamical_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_dynamical_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    one = np.load(tmp_path / "w1" / "rank0.npz")
    assert one["msd"].shape[0] == 7 and one["j"].shape == (3, 2, 7) and np.abs(one["j"]).max() > 0
    for rank in range(2):
        two = np.load(tmp_path / "w2" / ("rank%d.npz" % rank))
        for key in ("msd", "msd_all", "aa", "j", "time"):
            np.testing.assert_array_equal(two[key], one[key], err_msg=key)
        np.testing.assert_allclose(two["msd_int"], one["msd_int"], rtol=1e-13)


def _visc_worker(rank, world, port, tmp_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from mdproptools_amd import backend
    from mdproptools_amd.dynamical.viscosity import Viscosity
    from oracle import cpu_ref as O

    def xcorr(a, b=None, method=0, n_lags=None, ctx=None):
        return np.stack([O.xcorr_fft(row, row) for row in np.atleast_2d(a)])

    def cumtrapz(y, dx, leading_zero=False, ctx=None):
        return np.stack([O.cumtrapz(row, dx, leading_zero) for row in np.atleast_2d(y)])

    def green_kubo(a, b=None, method=0, acf_scale=1.0, dx=1.0, integral_scale=1.0, leading_zero=False, want_acf=True,
                   want_mean=False, ctx=None):
        acf = xcorr(a) * acf_scale
        integral = np.multiply(integral_scale, cumtrapz(acf, dx, leading_zero))
        return (acf if want_acf else None), integral, (np.mean(integral, axis=0) if want_mean else None)

    backend.xcorr, backend.cumtrapz, backend.green_kubo = xcorr, cumtrapz, green_kubo
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    v = Viscosity("log.rep*", cutoff_time=20, volume=1000.0, temp=300.0, timestep=1, working_dir=tmp_dir)
    avg, data, acf, t = v.calc_avg_visc(output_all_data=True)
    np.savez(os.path.join(tmp_dir, "visc_w%d_r%d.npz" % (world, rank)), avg=np.stack(avg), data=np.stack(data),
             acf=np.stack(acf), t=t)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_dropin_viscosity_replicates_sharded_world2_gloo(tmp_path):
    """Viscosity.calc_avg_visc under torch.distributed: the replicate logs are dealt to the ranks, every rank returns
    all replicates in file order, equal to the single-process result."""
    import glob

    import torch.multiprocessing as mp

    from mdproptools_amd import io as mio

    rng = np.random.default_rng(4)
    for k in range(5):
        n = 300
        tbl = np.column_stack([np.arange(n) * 10, rng.normal(0, 50, (n, 3))])
        mio.write_log(str(tmp_path / ("log.rep%d" % k)), tbl, ["Step", "Pxy", "Pxz", "Pyz"])
    mp.spawn(_visc_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_visc_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    one = np.load(tmp_path / "visc_w1_r0.npz")
    assert one["avg"].shape[0] == len(glob.glob(str(tmp_path / "log.rep*"))) == 5
    for rank in range(2):
        two = np.load(tmp_path / ("visc_w2_r%d.npz" % rank))
        for key in ("avg", "data", "acf", "t"):
            np.testing.assert_array_equal(two[key], one[key], err_msg=key)


# ------------------------------------------------------------------ rank-local failures are agreed on collectively
def _raise_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from mdproptools_amd import dist as D

    dist.init_process_group("gloo", rank=rank, world_size=world)
    D.raise_together(None)  # nobody failed: goes through
    got = "none"
    try:
        D.raise_together(ValueError("atom masses change between frames") if rank == 1 else None, "parsing")
    except ValueError as e:
        got = "own:" + str(e)
    except RuntimeError as e:
        got = "other:" + str(e)
    # both ranks are still in step: the next collective completes
    counts = D.allgather_counts(rank + 5)
    with open(os.path.join(out_dir, "raise%d.txt" % rank), "w") as fh:
        fh.write(got + "|" + ",".join(map(str, counts)))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_local_failure_makes_every_rank_raise(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_raise_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = (tmp_path / "raise0.txt").read_text()
    r1 = (tmp_path / "raise1.txt").read_text()
    assert r0.startswith("other:rank(s) [1] failed on parsing") and r0.endswith("|5,6")
    assert r1 == "own:atom masses change between frames|5,6"
