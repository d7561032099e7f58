"""
BASELINE.json configs[2..4] at FULL size through the C-ABI (the bench legs' parity, as tests):

  C3  100 000 atoms x 1000 frames: per-frame == frame-summed == sum; partials add up to the full histogram; frames
      {0, 499, 999} against the C oracle bit for bit (RDF and CN; a frame split over the host cores by head rows);
      packed-f32 sweep == all-f64 sweep; the RDF+CN call in one sweep == the separate calls.
  C4  50 000 entities x 5000 frames: single origin against the oracle on the first frame pairs, fixed lag against numpy
      on an entity subset, the default full-lag path against the difference kernel within the reported bound and
      against the oracle on sampled lags of an entity group.
  C5  n = 1e6: FFT against direct on the first half of the lags, direct against the oracle on sampled lags, a lag
      range against the whole function.
Also here: the device-resident RDF sums (mdhip_rdf_atomic_dev) behind a one-rank RCCL group, and bench.py's own
N > 1 launch (two ranks sharing the GPU over gloo, strong scaling).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import cref as C

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THREADS = min(os.cpu_count() or 1, 32)


@pytest.fixture(scope="module")
def B():
    from mdproptools_amd import backend

    return backend


def cn_pairs_threaded(x0, ty, rel, L, rc2, n_threads):
    """oracle CN of one frame, head rows dealt to threads with equal pair counts."""
    from concurrent.futures import ThreadPoolExecutor

    n = x0.shape[1]
    total = n * (n - 1) / 2.0
    bounds = [0] + [int(n - (1.0 + np.sqrt(1.0 + 8.0 * total * (1.0 - k / n_threads))) / 2.0)
                    for k in range(1, n_threads)] + [n]
    with ThreadPoolExecutor(max_workers=n_threads) as pool:
        parts = list(pool.map(lambda k: C.cn_pairs(x0, ty, rel, L, rc2, rows=(bounds[k], bounds[k + 1])),
                              range(n_threads)))
    return sum(parts)


@pytest.fixture(scope="module")
def c3():
    import torch

    from mdproptools_amd import synth

    cfg = synth.rdf_config("C3")
    n, L, F = cfg["n_atoms"], cfg["box_len"], cfg["n_frames"]
    xyz = torch.empty((F, 3, n), dtype=torch.float64, device="cuda")
    for f0 in range(0, F, 50):
        xyz[f0:f0 + 50] = torch.from_numpy(synth.rdf_frames(n, range(f0, f0 + 50), L, cfg["seed_offset"])).cuda()
    yield dict(xyz=xyz, ty=synth.rdf_types(n), rel=np.array(synth.ALL_PAIRS_4, dtype=np.int32),
               box=np.full((F, 3), L), L=L, n=n, F=F, cuts=synth.cn_cutoffs(10))
    del xyz
    torch.cuda.empty_cache()


def test_c3_full_size_rdf(B, c3):
    from mdproptools_amd._lib import Context

    xyz, ty, rel, box, L, n, F = (c3[k] for k in ("xyz", "ty", "rel", "box", "L", "n", "F"))
    full, part, ov = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=False)
    pf, pp, ov2 = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=True)
    assert ov == 0 and ov2 == 0
    np.testing.assert_array_equal(pf.sum(axis=0), full)
    np.testing.assert_array_equal(pp.sum(axis=0), part)
    mult = np.array([1 if a == b else 2 for a, b in rel], dtype=np.uint64)
    np.testing.assert_array_equal((part * mult[:, None]).sum(axis=0), full)  # the 10 type pairs cover every pair
    for f in (0, 499, 999):
        cf, cp, _ = C.rdf_pairs_threaded(xyz[f].cpu().numpy(), ty, rel, [L] * 3, 400.0, 0.05, 400, THREADS)
        np.testing.assert_array_equal(pf[f], cf)
        np.testing.assert_array_equal(pp[f], cp)
    ctx = Context(0)
    ctx.set_option("rdf_pk", 0)  # the all-f64 sweep
    f64, p64, _ = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=False, ctx=ctx)
    assert "<2" in ctx.last_kernel_name()
    ctx.close()
    np.testing.assert_array_equal(f64, full)
    np.testing.assert_array_equal(p64, part)
    frac = float(full.sum()) / 2 / (F * n * (n - 1) / 2)
    assert abs(frac - 4 / 3 * np.pi * 20.0 ** 3 / L ** 3) < 1e-5  # ideal gas, 5e12 pairs


def test_c3_full_size_cn(B, c3):
    xyz, ty, rel, box, L, cuts = (c3[k] for k in ("xyz", "ty", "rel", "box", "L", "cuts"))
    cn = B.cn_loop(xyz, ty, box, rel, cuts, per_frame=False)
    cnf = B.cn_loop(xyz, ty, box, rel, cuts, per_frame=True)
    np.testing.assert_array_equal(cnf.sum(axis=0), cn)
    for f in (0, 499, 999):
        ref = cn_pairs_threaded(xyz[f].cpu().numpy(), ty, rel, [L] * 3, [c * c for c in cuts], THREADS)
        np.testing.assert_array_equal(cnf[f], ref)
    if hasattr(B, "rdf_cn_loop"):  # RDF and CN from one sweep: identical integers
        full, part, ov = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, per_frame=False)
        f2, p2, ov2, cn2 = B.rdf_cn_loop(xyz, ty, box, rel, 20.0, 0.05, 400, cuts, per_frame=False)
        np.testing.assert_array_equal(f2, full)
        np.testing.assert_array_equal(p2, part)
        np.testing.assert_array_equal(cn2, cn)
        f3, p3, ov3, cn3 = B.rdf_cn_loop(xyz[:40], ty, box[:40], rel, 20.0, 0.05, 400, cuts, per_frame=True)
        np.testing.assert_array_equal(cn3, cnf[:40])


def test_c3_device_resident_sums_behind_rccl(B, c3):
    """mdhip_rdf_atomic_dev + an RCCL all-reduce that reads the buffer the kernels wrote (a one-rank group: the
    collective is the identity, the code path is the N > 1 one) == the host-output call."""
    import socket

    import torch
    import torch.distributed as dist

    from mdproptools_amd import dist as D

    xyz, ty, rel, box = (c3[k] for k in ("xyz", "ty", "rel", "box"))
    sub, bsub = xyz[:64], box[:64]
    full, part, ov = B.rdf_loop(sub, ty, bsub, rel, 20.0, 0.05, 400, per_frame=False)
    out = torch.empty(11 * 400 + 1, dtype=torch.int64, device="cuda")
    B.rdf_loop_dev(sub, ty, bsub, rel, 20.0, 0.05, 400, out)
    flat = out.cpu().numpy().view(np.uint64)
    np.testing.assert_array_equal(flat[:400], full)
    np.testing.assert_array_equal(flat[400:4400].reshape(10, 400), part)
    assert int(flat[4400]) == ov
    # small frames take the dense kernels: the host fallback inside the same entry point
    small = torch.from_numpy(np.random.default_rng(3).uniform(0, 30, (3, 3, 900))).cuda()
    tys = (1 + np.arange(900) % 4).astype(np.int32)
    fs, ps, _ = B.rdf_loop(small, tys, np.full((3, 3), 30.0), rel, 10.0, 0.05, 200, per_frame=False)
    outs = torch.empty(11 * 200 + 1, dtype=torch.int64, device="cuda")
    B.rdf_loop_dev(small, tys, np.full((3, 3), 30.0), rel, 10.0, 0.05, 200, outs)
    np.testing.assert_array_equal(outs.cpu().numpy().view(np.uint64)[:200], fs)
    np.testing.assert_array_equal(outs.cpu().numpy().view(np.uint64)[200:2200].reshape(10, 200), ps)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert D._on_rccl(sub)
        a, p, o = D.rdf_sharded_async(sub, ty, bsub, rel, 20.0, 0.05, 400).wait()
        np.testing.assert_array_equal(a, full)
        np.testing.assert_array_equal(p, part)
        assert int(o[0]) == ov
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def c4():
    import torch

    from mdproptools_amd import synth

    E, F = 50_000, 5000
    g = torch.Generator(device="cuda")
    g.manual_seed(synth.BASE_SEED + 4)
    r = torch.empty((F, 3, E), dtype=torch.float64, device="cuda")
    r[0] = torch.rand((3, E), generator=g, device="cuda", dtype=torch.float64) * 82.8
    for f0 in range(1, F, 250):
        f1 = min(F, f0 + 250)
        st = torch.randn((f1 - f0, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1
        r[f0:f1] = r[f0 - 1] + torch.cumsum(st, dim=0)
        del st
    yield dict(r=r, E=E, F=F)
    del r
    torch.cuda.empty_cache()


def test_c4_full_size_msd(B, c4):
    from mdproptools_amd._lib import Context

    r, E, F = c4["r"], c4["E"], c4["F"]
    pairs = [(0, t) for t in range(F)]
    s1 = B.msd_pairs(r, pairs, [0, 64, E], scale=1.0)
    nsub = 120
    ref = C.msd_pairs(r[:nsub].cpu().numpy(), pairs[:nsub], [0, 64, E])
    np.testing.assert_allclose(s1[:nsub], ref, rtol=1e-10)
    tot = s1.sum(axis=1)[:, 3] / E
    assert abs(tot[-1] / (3 * 0.01 * (F - 1)) - 1.0) < 0.02  # random walk, sigma 0.1 per axis and frame
    # fixed lag (tao = 4), per entity, against numpy on the first 64 entities
    win = B.msd_windows(r, 4, scale=1.0)
    kept = r[::4, :, :64].cpu().numpy()
    d2 = (kept[1:] - kept[:-1]) ** 2
    np.testing.assert_allclose(win[:64, :3], d2.sum(axis=0).T, rtol=1e-12)
    np.testing.assert_allclose(win[:64, 3], d2.sum(axis=(0, 1)), rtol=1e-12)
    # full lag x origin average: the default path against the difference kernel, within the bound it reports
    goff = [0, 64, E]
    lag = B.lag_msd(r, F - 1, goff)
    from mdproptools_amd._lib import default_context

    bound = default_context().last_rel_bound()
    ctx = Context(0)
    ctx.set_option("lag_variant", 1)
    lagd = B.lag_msd(r, F - 1, goff, ctx=ctx)
    assert ctx.last_kernel_name() == "lag_msd_lds_kernel"
    ctx.close()
    assert bound <= 1e-10  # else the default falls back to the difference kernel itself
    rel_err = np.max(np.abs(lag[1:] - lagd[1:]) / lagd[1:])
    assert rel_err <= max(bound, 1e-13), (rel_err, bound)
    # sampled lags of the 64-entity group against the oracle
    lags = np.unique(np.concatenate([[0, 1, 2, 3, 7, 8, 9, 511, 512, 513], np.linspace(10, F - 1, 30).astype(int)]))
    refl = C.lag_msd(r[:, :, :64].contiguous().cpu().numpy(), lags, [0, 64])
    np.testing.assert_allclose(lagd[lags, 0, :], refl[:, 0, :], rtol=1e-10)
    np.testing.assert_allclose(lag[lags, 0, :], refl[:, 0, :], rtol=1e-10, atol=1e-300)
    theory = 3 * 0.01 * np.arange(1, F)
    np.testing.assert_allclose(lag[1:, 1, 3] / theory, 1.0, atol=0.2)


def test_full_lag_msd_at_twice_and_four_times_c4s_length(B):
    """Round 6: trajectories beyond the fused kernels at C4's atom count — 10 000 frames (12 GB resident; padded length 24 576,
    msd_power_w12p_kernel) and 20 000 frames x 20 000 atoms (9.6 GB; padded length 49 152, + msd_power_w12o_kernel): the
    default path on the device with its status word, a 64-entity group of the SAME call against the exact-difference kernel
    within the reported bound and against the C oracle on sampled lags, the diffusive slope of the whole system
    (diffusion.py:207-238's quantity, every origin)."""
    import torch

    from mdproptools_amd import synth
    from mdproptools_amd._lib import Context, default_context

    for F, E, name in ((10_000, 50_000, "msd_power_w12p_kernel"), (20_000, 20_000, "msd_power_w12p_kernel + msd_power_w12o_kernel")):
        g = torch.Generator(device="cuda")
        g.manual_seed(synth.BASE_SEED + 40 + F)
        r = torch.empty((F, 3, E), dtype=torch.float64, device="cuda")
        r[0] = torch.rand((3, E), generator=g, device="cuda", dtype=torch.float64) * 82.8
        for f0 in range(1, F, 250):
            f1 = min(F, f0 + 250)
            st = torch.randn((f1 - f0, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1
            r[f0:f1] = r[f0 - 1] + torch.cumsum(st, dim=0)
            del st
        goff = [0, 64, E]
        out = torch.empty((F, 2, 4), dtype=torch.float64, device="cuda")
        status = torch.full((1,), -1.0, dtype=torch.float64, device="cuda")
        # (20 000 frames: the bound of a 64-entity group passes 1e-10 and the DEFAULT would hand the call to the difference
        # kernel — seconds; the spectral path is asked for and held to the bound it reports)
        ctx0 = default_context()
        if F > 12288:
            ctx0.set_option("lag_variant", 2)
        try:
            B.lag_msd(r, F - 1, goff, out=out, async_=True, status_out=status).wait()
        finally:
            ctx0.set_option("lag_variant", -1)
        bound = ctx0.last_rel_bound()
        assert ctx0.last_kernel_name() == name and float(status.item()) == bound, (F, ctx0.last_kernel_name(), bound)
        assert 0.0 < bound <= (1e-10 if F <= 12288 else 1e-9), (F, bound)
        lag = out.cpu().numpy()
        rsub = r[:, :, :64].contiguous()
        del r
        torch.cuda.empty_cache()
        ctx = Context(0)
        ctx.set_option("lag_variant", 1)
        lagd = B.lag_msd(rsub, F - 1, [0, 64], ctx=ctx)
        assert ctx.last_kernel_name().startswith("lag_msd_")
        ctx.close()
        rel = np.max(np.abs(lag[1:, 0] - lagd[1:, 0]) / lagd[1:, 0])
        assert rel <= max(bound, 1e-13), (F, rel, bound)
        lags = np.unique(np.concatenate([[0, 1, 2, 3, 511, 512, 6143, 6144, 6145], np.linspace(10, F - 1, 20).astype(int)]))
        refl = C.lag_msd(rsub.cpu().numpy(), lags, [0, 64])
        np.testing.assert_allclose(lag[lags, 0, :], refl[:, 0, :], rtol=1e-9, atol=1e-300)
        theory = 3 * 0.01 * np.arange(1, F // 2)
        np.testing.assert_allclose(lag[1:F // 2, 1, 3] / theory, 1.0, atol=0.2)
        del out, rsub
        torch.cuda.empty_cache()


def test_c5_full_size_acf(B):
    import torch

    from mdproptools_amd import synth

    n = 1_000_000
    ph = synth.ar1_series(n)
    p = torch.from_numpy(ph).cuda()
    fft = B.xcorr(p, method=B.XCORR_FFT)
    direct = B.xcorr(p, method=B.XCORR_DIRECT)
    half = n // 2
    for k in range(3):
        # the FFT estimator's rounding error (~1e-16 n acf[0]) is divided by n - k: only the first half of the lags
        # is a meaningful comparison at n = 1e6, in numpy's FFT as well
        assert np.max(np.abs(fft[k][:half] - direct[k][:half])) <= 1e-10 * direct[k][0]
        assert np.max(np.abs(fft[k] - direct[k])) <= 1e-8 * direct[k][0]
        np.testing.assert_allclose(direct[k][0], np.mean(ph[k] ** 2), rtol=1e-12)
    ref = C.xcorr_direct(ph[1], ph[1], n_lags=64)
    np.testing.assert_allclose(direct[1][:64], ref, rtol=0, atol=1e-10 * ref[0])
    for k0 in (250_000, 999_000):  # late lags: the tail of the series only
        exact = float(np.dot(ph[1][k0:], ph[1][:n - k0])) / (n - k0)
        assert abs(direct[1][k0] - exact) <= 1e-10 * direct[1][0]
    # a lag range == the same lags of the whole function (the unit of the multi-GPU split)
    part = B.xcorr(p, method=B.XCORR_DIRECT, lag_begin=700_001, n_lags=12_345)
    # (time slabs are cut per launch, so the order of the partial sums differs: equal to rounding, not bit for bit)
    np.testing.assert_allclose(part, direct[:, 700_001:712_346], rtol=0, atol=1e-12 * direct[0][0])
    cross = B.xcorr(p[0], p[2], method=B.XCORR_DIRECT, lag_begin=5, n_lags=300)
    refc = C.xcorr_direct(ph[0], ph[2], n_lags=305)[5:]
    np.testing.assert_allclose(cross, refc, rtol=0, atol=1e-10 * abs(refc).max())
    integ = B.cumtrapz(fft, 1e-15)
    np.testing.assert_allclose(integ[:, -1], 1e-15 * (fft.sum(axis=1) - 0.5 * (fft[:, 0] + fft[:, -1])), rtol=1e-9)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself (here sharing the GPU over
    gloo) — strong scaling, C3's 1000 frames split 500 + 500, all-reduce inside the timed region."""
    env = dict(os.environ, MDHIP_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--scaling", "strong",
                        "--steps", "2", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["frames_per_gpu"] == 500
    assert line["config"]["pairs_per_step"] == 1000 * 100_000 * 99_999 // 2
    assert line["value"] > 1e12


def test_bench_c4_workload_two_ranks():
    """`python bench.py --gpus 2 --workload c4`: the MSD half of BASELINE.json's metric at N > 1 — frames dealt to the
    ranks for single-origin / fixed-lag, entities for the full lag average, collectives inside the timed region (two
    ranks sharing the GPU over gloo here); bench.py's own checks tie the two shardings together."""
    env = dict(os.environ, MDHIP_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "c4",
                        "--steps", "2", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["metric"] == "frame-pairs/s" and line["unit"].startswith("effective frame-pairs/s") and line["n_gpus"] == 2
    assert line["scaling"] == "strong" and line["config"]["frames_per_gpu"] == 2500
    assert line["config"]["entities_per_gpu"] == 25_000
    assert line["config"]["collectives"]["world_size"] == 2 and line["config"]["collectives"]["backend"] == "gloo"
    assert line["value"] > 1e6 and line["msd"]["single_origin"]["value_at_kernel_time"] > 1e4
    assert line["msd"]["kernel_ms_per_step"] > 0 and len(line["msd"]["step_ms"]["raw"]) == 2
    assert line["lib_build_id"]["match"] is True


def test_bench_through_rccl_with_one_rank():
    """Both bench workloads with a ONE-rank RCCL process group (MDHIP_BENCH_FORCE_DIST=1): every collective of the
    N > 1 path — histogram all-reduce on the device buffer, origin broadcast, row all-gathers, halo all-gather, the
    entity-sharded all-reduce, the max-over-ranks timing — goes through RCCL itself on device tensors (two ranks
    cannot share one GPU under RCCL; the two-rank tests above use gloo)."""
    env = dict(os.environ, MDHIP_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MDHIP_DIST_BACKEND"):
        env.pop(k, None)
    for port, extra in ((29541, ["--workload", "c4"]), (29542, ["--legs", "parity", "--no-cpu-baseline", "--msd-steps", "1"])):
        env["MASTER_PORT"] = str(port)
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1"] + extra,
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        coll = line["config"]["collectives"]
        assert coll["backend"] == "nccl" and coll["world_size"] == 1 and line["n_gpus"] == 1
        assert line["msd"]["value"] > 1e8 and line["value"] > 1e8
        if "--workload" not in extra:
            assert line["metric"] == "atom-pairs/s" and line["parity_checked"] is True


# ------------------------------------------------------------------ compute-bound paths sharded over two ranks
def _sharded_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist

    from mdproptools_amd import dist as D

    dist.init_process_group("gloo", rank=rank, world_size=world)  # two ranks share the one GPU of the test box
    rng = np.random.default_rng(31)
    F, E = 600, 3000
    r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(0, 40, (1, 3, E))
    goff = [0, 1000, 1900, E]  # group 1 straddles the rank boundary at entity 1500
    lo, hi = D.entity_shard(E)
    lag = D.lag_msd_sharded(r[:, :, lo:hi], (lo, hi), F - 1, goff, scale=1.0)
    series = np.cumsum(rng.normal(size=(3, 40_000)), axis=1)
    acf = D.xcorr_direct_sharded(series)
    ccf = D.xcorr_direct_sharded(series[0], series[1], n_lags=25_000)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), lag=lag, acf=acf, ccf=ccf)
    dist.barrier()
    dist.destroy_process_group()


def test_lag_msd_and_direct_acf_sharded_two_ranks(B, tmp_path):
    """lag_msd_sharded (entities split, a group straddling the boundary) and xcorr_direct_sharded (lag ranges of equal
    work) with the real kernels, two ranks sharing the GPU over gloo, against the single-process calls."""
    import socket

    import torch.multiprocessing as mp

    from mdproptools_amd._lib import Context

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(31)
    F, E = 600, 3000
    r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(0, 40, (1, 3, E))
    ctx = Context(0)
    ctx.set_option("lag_variant", 1)  # the exact-difference kernel as the reference
    one = B.lag_msd(r, F - 1, [0, 1000, 1900, E], ctx=ctx)
    ctx.close()
    series = np.cumsum(rng.normal(size=(3, 40_000)), axis=1)
    acf = B.xcorr(series, method=B.XCORR_DIRECT)
    ccf = B.xcorr(series[0], series[1], method=B.XCORR_DIRECT, n_lags=25_000)
    for rank in range(2):
        g = np.load(tmp_path / ("rank%d.npz" % rank))
        np.testing.assert_allclose(g["lag"][1:], one[1:], rtol=1e-10)
        assert np.all(g["lag"][0] == 0.0)
        np.testing.assert_allclose(g["acf"], acf, rtol=0, atol=1e-12 * abs(acf[:, 0]).max())
        np.testing.assert_allclose(g["ccf"], ccf, rtol=0, atol=1e-12 * abs(acf[:, 0]).max())


# ------------------------------------------------------------------ frame-sharded paths over two ranks, real kernels
def _frame_case():
    rng = np.random.default_rng(77)
    F, n = 9, 4000  # odd frame count: shards of 5 and 4
    L = np.array([34.0, 35.0, 36.0])
    xyz = rng.uniform(0, 1, (F, 3, n)) * L[None, :, None]
    ty = (1 + np.arange(n) % 3).astype(np.int32)
    rel = np.array([[1, 1], [1, 2], [2, 3], [3, 3]])
    box = np.tile(L, (F, 1)) * (1 + 0.01 * np.arange(F))[:, None]  # NPT: every frame its own box
    walk = np.cumsum(rng.normal(0, 0.1, (F, 3, n)), axis=0) + rng.uniform(0, 30, (1, 3, n))
    vel = rng.normal(0, 1e-3, (F, 3, n))
    mass = 1.0 + (np.arange(n) % 4)
    q = np.where(np.arange(n) < 2000, 0.25, -0.5)
    seg_off = np.concatenate([np.arange(0, 2000, 4), np.arange(2000, n + 1, 2)]).astype(np.int64)
    seg_type = np.concatenate([np.zeros(500, np.int32), np.ones(1000, np.int32)])
    return F, n, xyz, ty, rel, box, walk, vel, mass, q, seg_off, seg_type


def _frame_sharded_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch
    import torch.distributed as dist

    from mdproptools_amd import dist as D

    dist.init_process_group("gloo", rank=rank, world_size=world)  # two ranks share the one GPU of the test box
    F, n, xyz, ty, rel, box, walk, vel, mass, q, seg_off, seg_type = _frame_case()
    lo, hi = D.frame_shard(F)
    res = {}
    dev = torch.device("cuda", 0)
    for tag, put in (("h", lambda a: a), ("d", lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev))):
        # host arrays and device-resident shards go through the same functions (device: the *_dev entry points and
        # collectives on the buffers the kernels wrote — staged through the host only because this is gloo)
        x_l, w_l, v_l = put(xyz[lo:hi]), put(walk[lo:hi]), put(vel[lo:hi])
        res["cn_" + tag] = D.cn_sharded(x_l, ty, box[lo:hi], rel, [2.0, 3.0, 4.0, 5.5])
        res["s0_" + tag] = D.msd_single_origin_sharded(w_l, F, [0, 1500, n], scale=1e-10, origin_frame=0)
        res["s7_" + tag] = D.msd_single_origin_sharded(w_l, F, [0, 1500, n], scale=1e-10, origin_frame=7)
        res["w2_" + tag] = D.msd_windows_sharded(w_l, F, 2, scale=1e-10)
        res["w3_" + tag] = D.msd_windows_sharded(w_l, F, 3, scale=1e-10)
        res["j_" + tag] = D.charge_flux_sharded(v_l, F, mass, q, seg_off, seg_type, 2, 1e5, 1.602e-19)
    # (device-resident frames: the packed per-frame rows are gathered as a device tensor)
    pf, pp, ov = D.rdf_sharded_per_frame(torch.from_numpy(xyz[lo:hi]).to(dev), ty, box[lo:hi], rel, 8.0, 0.05, 160, F)
    full, part, ovs = D.rdf_sharded(torch.from_numpy(xyz[lo:hi]).to(dev), ty, box[lo:hi], rel, 8.0, 0.05, 160)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), pf=pf, pp=pp, ov=ov, full=full, part=part, **res)
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharded_paths_two_ranks_real_kernels(B, tmp_path):
    """cn_sharded, msd_single_origin_sharded (origin in either shard), msd_windows_sharded (one-frame halo),
    charge_flux_sharded and rdf_sharded_per_frame with the HIP kernels behind them — two ranks sharing the GPU over
    gloo, host-resident and device-resident shards — against the single-process calls and the oracle."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_frame_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    F, n, xyz, ty, rel, box, walk, vel, mass, q, seg_off, seg_type = _frame_case()
    cn = B.cn_loop(xyz, ty, box, rel, [2.0, 3.0, 4.0, 5.5], per_frame=False)
    assert np.array_equal(cn, sum(C.cn_pairs(xyz[f], ty, rel, box[f], [4.0, 9.0, 16.0, 30.25]) for f in range(F)))
    s0 = B.msd_pairs(walk, [(0, t) for t in range(F)], [0, 1500, n], scale=1e-10)
    s7 = B.msd_pairs(walk, [(7, t) for t in range(F)], [0, 1500, n], scale=1e-10)
    np.testing.assert_allclose(s0, C.msd_pairs(walk * 1e-10, [(0, t) for t in range(F)], [0, 1500, n]), rtol=1e-12)
    w2 = B.msd_windows(walk, 2, scale=1e-10)
    w3 = B.msd_windows(walk, 3, scale=1e-10)
    j = B.charge_flux(vel, mass, q, seg_off, seg_type, 2, 1e5, 1.602e-19)
    pf, pp, ov = B.rdf_loop(xyz, ty, box, rel, 8.0, 0.05, 160, per_frame=True)
    for f in (0, F - 1):
        cf, cp, _ = C.rdf_pairs(xyz[f], ty, rel, box[f], 64.0, 0.05, 160)
        assert np.array_equal(pf[f], cf) and np.array_equal(pp[f], cp)
    for rank in range(2):
        g = np.load(tmp_path / ("rank%d.npz" % rank))
        for tag in ("h", "d"):
            np.testing.assert_array_equal(g["cn_" + tag], cn)               # integers: exact
            np.testing.assert_array_equal(g["s0_" + tag], s0)               # a frame's sums come from one rank, same kernel
            np.testing.assert_array_equal(g["s7_" + tag], s7)
            np.testing.assert_allclose(g["w2_" + tag], w2, rtol=1e-13)      # windows summed rank by rank
            np.testing.assert_allclose(g["w3_" + tag], w3, rtol=1e-13)
            np.testing.assert_array_equal(g["j_" + tag], j)
        np.testing.assert_array_equal(g["pf"], pf)
        np.testing.assert_array_equal(g["pp"], pp)
        np.testing.assert_array_equal(g["full"], pf.sum(axis=0))
        np.testing.assert_array_equal(g["part"], pp.sum(axis=0))
