"""
The asynchronous entry points (include/mdhip.h "asynchronous calls"; SURVEY.md 8b: "an _async variant + mdhip_sync"):
every *_async call must leave, once it has completed, exactly what its synchronous twin leaves — bit for bit, since
both run the same kernels in the same order — whatever is in flight around it, and errors must surface at the wait.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from mdproptools_amd import backend, synth
    from mdproptools_amd._lib import Context

    ctx = Context(0)
    yield backend, synth, torch, ctx
    ctx.close()


def _frames(synth, torch, n, F, L, seed, device=True):
    x = synth.rdf_frames(n, range(F), L, seed)
    return torch.from_numpy(x).cuda() if device else x


def test_rdf_cn_async_equals_sync_with_several_calls_in_flight(env):
    B, synth, torch, ctx = env
    n, F, L = 6000, 24, 42.0
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    cuts = synth.cn_cutoffs(len(rel))
    xs = [_frames(synth, torch, n, F, L, 40 + k) for k in range(3)]
    for per_frame in (False, True):
        ref = [B.rdf_loop(x, ty, box, rel, 12.0, 0.05, 240, per_frame=per_frame, ctx=ctx) for x in xs]
        ref_cn = [B.cn_loop(x, ty, box, rel, cuts, per_frame=per_frame, ctx=ctx) for x in xs]
        ref_both = B.rdf_cn_loop(xs[0], ty, box, rel, 12.0, 0.05, 240, cuts, per_frame=per_frame, ctx=ctx)
        # three histogram calls, three CN calls and a one-sweep call queued behind each other, then ONE wait
        hs = [B.rdf_loop(x, ty, box, rel, 12.0, 0.05, 240, per_frame=per_frame, ctx=ctx, async_=True) for x in xs]
        hc = [B.cn_loop(x, ty, box, rel, cuts, per_frame=per_frame, ctx=ctx, async_=True) for x in xs]
        hb = B.rdf_cn_loop(xs[0], ty, box, rel, 12.0, 0.05, 240, cuts, per_frame=per_frame, ctx=ctx, async_=True)
        assert ctx.pending() == 7
        got_b = hb.wait()  # completes everything issued before it too
        assert ctx.pending() == 0
        for h, r in zip(hs, ref):
            g = h.wait()
            np.testing.assert_array_equal(g[0], r[0])
            np.testing.assert_array_equal(g[1], r[1])
            assert g[2] == r[2]
        for h, r in zip(hc, ref_cn):
            np.testing.assert_array_equal(h.wait(), r)
        for a, b in zip(got_b, ref_both):
            np.testing.assert_array_equal(a, b)
    # the stats of the calls completed last are remembered, newest first: the one-sweep kernel, then a CN sweep
    ms, aux, launches, name = ctx.call_stats(0)
    assert ms > 0 and launches >= 1 and "pair_hist" in name
    assert ctx.call_stats(1)[0] > 0


def test_double_buffered_steps_and_device_results(env):
    """The bench's pattern: issue step k + 1, then wait for step k (mdhip_wait(ctx, 1)) — with host results and with
    the sums left in a device buffer (the multi-GPU path's form)."""
    B, synth, torch, ctx = env
    n, F, L = 5000, 16, 40.0
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    xs = [_frames(synth, torch, n, F, L, 60 + k) for k in range(5)]
    ref = [B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=ctx) for x in xs]
    pending, got = None, []
    for x in xs:
        h = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=ctx, async_=True)
        if pending is not None:
            got.append(pending.wait())
            assert ctx.pending() == 1
        pending = h
    got.append(pending.wait())
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g[0], r[0])
        np.testing.assert_array_equal(g[1], r[1])
    words = (1 + len(rel)) * 200 + 1
    outs = [torch.empty(words, dtype=torch.int64, device="cuda") for _ in xs]
    hs = [B.rdf_loop_dev(x, ty, box, rel, 10.0, 0.05, 200, o, ctx=ctx, async_=True) for x, o in zip(xs, outs)]
    ctx.sync()
    for h, o, r in zip(hs, outs, ref):
        flat = h.wait().cpu().numpy().view(np.uint64)
        np.testing.assert_array_equal(flat[:200], r[0])
        np.testing.assert_array_equal(flat[200:-1].reshape(len(rel), 200), r[1])
        assert int(flat[-1]) == r[2]
    cn_ref = B.cn_loop(xs[0], ty, box, rel, synth.cn_cutoffs(len(rel)), per_frame=False, ctx=ctx)
    cn_dev = torch.empty(len(rel), dtype=torch.int64, device="cuda")
    B.cn_loop(xs[0], ty, box, rel, synth.cn_cutoffs(len(rel)), per_frame=False, ctx=ctx, out=cn_dev, async_=True).wait()
    np.testing.assert_array_equal(cn_dev.cpu().numpy().view(np.uint64), cn_ref)


def test_overflow_guard_rerun_inside_an_async_call(env):
    """A launch that raises the overflow guard of the 32-bit block histograms cannot be fixed while it is in flight:
    its completion step runs the batch again, synchronously, in halves. Same integers, host and device results."""
    B, synth, torch, ctx = env
    from mdproptools_amd._lib import Context

    n, F, L = 6000, 96, 42.0
    x = _frames(synth, torch, n, F, L, 11)
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    ref = B.rdf_loop(x, ty, box, rel, 12.0, 0.05, 240, per_frame=False, ctx=ctx)
    c2 = Context(0)
    try:
        hit = False
        for guard in (1024, 512, 256, 128, 64, 32):
            c2.set_option("rdf_guard", guard)
            try:
                g = B.rdf_loop(x, ty, box, rel, 12.0, 0.05, 240, per_frame=False, ctx=c2, async_=True).wait()
            except Exception as e:  # a threshold even one frame exceeds: the clean error, delivered by the wait
                assert "overflow the 32-bit" in str(e)
                break
            np.testing.assert_array_equal(g[0], ref[0])
            np.testing.assert_array_equal(g[1], ref[1])
            if c2.last_kernel_ms()[1] > 1:  # the re-run needed more than one launch
                hit = True
                out = torch.empty((1 + len(rel)) * 240 + 1, dtype=torch.int64, device="cuda")
                B.rdf_loop_dev(x, ty, box, rel, 12.0, 0.05, 240, out, ctx=c2, async_=True).wait()
                flat = out.cpu().numpy().view(np.uint64)
                np.testing.assert_array_equal(flat[:240], ref[0])
                np.testing.assert_array_equal(flat[240:-1].reshape(len(rel), 240), ref[1])
                break
        assert hit, "no threshold made the asynchronous call re-run its batch"
    finally:
        c2.close()


def test_inputs_the_async_path_does_not_take(env):
    """Frames in host memory (staged batch by batch), frames too small for the culled sweep (dense kernels), many
    classes (class passes): the *_async entry point completes that work before it returns — same results."""
    B, synth, torch, ctx = env
    rng = np.random.default_rng(5)
    cases = []
    n, F, L = 5000, 40, 40.0
    cases.append((synth.rdf_frames(n, range(F), L, 3), synth.rdf_types(n), np.array(synth.ALL_PAIRS_4), L, 10.0, 200))
    n = 700  # fewer than 8 tiles: dense LDS-tile kernel
    cases.append((torch.from_numpy(rng.uniform(0, 20.0, (6, 3, n))).cuda(), (1 + np.arange(n) % 3).astype(np.int32),
                  np.array([[1, 1], [1, 2], [2, 3]]), 20.0, 8.0, 160))
    n, T = 2600, 24  # 301 classes: several class passes
    rel24 = np.array([[a, b] for a in range(1, T + 1) for b in range(a, T + 1)], dtype=np.int32)
    cases.append((torch.from_numpy(rng.uniform(0, 30.0, (2, 3, n))).cuda(), rng.integers(1, T + 1, n).astype(np.int32),
                  rel24, 30.0, 9.0, 90))
    for x, ty, rel, L, rc, nb in cases:
        F = x.shape[0]
        box = np.full((F, 3), L)
        for per_frame in (False, True):
            r = B.rdf_loop(x, ty, box, rel, rc, rc / nb, nb, per_frame=per_frame, ctx=ctx)
            g = B.rdf_loop(x, ty, box, rel, rc, rc / nb, nb, per_frame=per_frame, ctx=ctx, async_=True).wait()
            np.testing.assert_array_equal(g[0], r[0])
            np.testing.assert_array_equal(g[1], r[1])
            assert g[2] == r[2]


def test_msd_step_async_equals_sync(env):
    """The three MSD calls of a C4 step, host and device results, queued back to back and waited for once."""
    B, synth, torch, ctx = env
    F, E = 400, 3000
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    r = torch.cumsum(torch.randn((F, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
    goff = [0, 1000, E]
    ref_o = B.msd_origin(r[1:], r[0], goff, scale=1e-10, ctx=ctx)
    ref_w = B.msd_windows(r, 4, scale=1e-10, ctx=ctx)
    ref_l = B.lag_msd(r, F - 1, goff, scale=1.0, ctx=ctx)
    bound = ctx.last_rel_bound()
    assert 0.0 < bound <= 1e-10
    for dev in (False, True):
        o = torch.empty((F - 1, 2, 4), dtype=torch.float64, device="cuda") if dev else None
        w = torch.empty((E, 4), dtype=torch.float64, device="cuda") if dev else None
        lg = torch.empty((F, 2, 4), dtype=torch.float64, device="cuda") if dev else None
        h1 = B.msd_origin(r[1:], r[0], goff, scale=1e-10, out=o, ctx=ctx, async_=True)
        h2 = B.msd_windows(r, 4, scale=1e-10, out=w, ctx=ctx, async_=True)
        h3 = B.lag_msd(r, F - 1, goff, scale=1.0, out=lg, ctx=ctx, async_=True)
        assert ctx.pending() == 3
        ctx.sync()
        assert ctx.last_rel_bound() == bound  # (the spectral path's bound, checked at completion)
        got = [h.wait() for h in (h1, h2, h3)]
        got = [t.cpu().numpy() if dev else t for t in got]
        np.testing.assert_array_equal(got[0], ref_o)
        np.testing.assert_array_equal(got[1], ref_w)
        np.testing.assert_array_equal(got[2], ref_l)
    # a trajectory whose bound is too loose for the spectral path (ballistic motion): the fallback to the difference
    # kernel happens inside the completion step
    t = torch.arange(F, device="cuda", dtype=torch.float64)[:, None, None]
    rb = (t * torch.rand((1, 3, E), generator=g, device="cuda", dtype=torch.float64) + 500.0).contiguous()
    ref_b = B.lag_msd(rb, F - 1, goff, scale=1.0, ctx=ctx)
    # (round 6: the difference form answers — for the whole call, or, when only a few lags at the ends of the range miss
    # the bound, for those lags: the reported bound is then that of the spectral rows that stood)
    assert ctx.last_rel_bound() <= 1e-10 and ("lag_msd" in ctx.last_kernel_name() or "lag_low_lags" in ctx.last_kernel_name())
    bound_b = ctx.last_rel_bound()
    got_b = B.lag_msd(rb, F - 1, goff, scale=1.0, ctx=ctx, async_=True).wait()
    np.testing.assert_array_equal(got_b, ref_b)
    assert ctx.last_rel_bound() == bound_b
    ctx.set_option("lag_variant", 1)
    try:
        exact_b = B.lag_msd(rb, F - 1, goff, scale=1.0, ctx=ctx)
    finally:
        ctx.set_option("lag_variant", -1)
    nzb = exact_b > 0
    assert (np.abs(ref_b[nzb] - exact_b[nzb]) / exact_b[nzb]).max() <= max(bound_b, 1e-13)


def test_full_lag_in_kernel_transposition_pipelined(env):
    """Trajectories long enough for the kernel that transposes its tiles itself (clusters of workgroups handing tiles to
    each other through a ring in device memory and per-tile counters that every call resets): four calls in flight on
    two different trajectories — ring, counters and result staging are shared by the calls and ordered by the stream —
    equal the synchronous calls bit for bit."""
    B, synth, torch, ctx = env
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    trajs = []
    for F, E, goff in ((2300, 500, [0, 200, 500]), (4100, 333, [0, 333])):
        r = torch.cumsum(torch.randn((F, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
        ref = B.lag_msd(r, F - 1, goff, scale=1.0, ctx=ctx)
        assert "msd_power_" in ctx.last_kernel_name() and "repeated" not in ctx.last_kernel_name()
        trajs.append((r, F, goff, ref))
    hs = [B.lag_msd(r, F - 1, goff, scale=1.0, ctx=ctx, async_=True) for _ in range(2) for r, F, goff, _ in trajs]
    assert ctx.pending() == 4
    got = [h.wait() for h in hs]
    for k, out in enumerate(got):
        np.testing.assert_array_equal(out, trajs[k % 2][3])


def test_flux_com_xcorr_cumtrapz_async_equal_sync(env):
    B, synth, torch, ctx = env
    rng = np.random.default_rng(9)
    F, M, per = 50, 600, 5
    N = M * per
    vel = torch.from_numpy(rng.standard_normal((F, 3, N))).cuda()
    mass = 1.0 + rng.random(N)
    q = rng.standard_normal(N)
    off = np.arange(0, N + 1, per)
    st = (np.arange(M) >= 200).astype(np.int32) + (np.arange(M) >= 400).astype(np.int32)
    ref_f = B.charge_flux(vel, mass, q, off, st, 3, 1e-5, 1.6e-19, ctx=ctx)
    ref_c = B.segment_com(vel, mass, off, atom_q=q, ctx=ctx)
    s = torch.from_numpy(rng.standard_normal((3, 20000))).cuda()
    ref_x = B.xcorr(s, method=B.XCORR_FFT, ctx=ctx)
    ref_d = B.xcorr(s, method=B.XCORR_DIRECT, ctx=ctx)
    ref_i = B.cumtrapz(ref_x, 2e-15, ctx=ctx)
    hf = B.charge_flux(vel, mass, q, off, st, 3, 1e-5, 1.6e-19, ctx=ctx, async_=True)
    hc = B.segment_com(vel, mass, off, atom_q=q, ctx=ctx, async_=True)
    hx = B.xcorr(s, method=B.XCORR_FFT, ctx=ctx, async_=True)
    hd = B.xcorr(s, method=B.XCORR_DIRECT, ctx=ctx, async_=True)
    hi = B.cumtrapz(ref_x, 2e-15, ctx=ctx, async_=True)
    fdev = torch.empty((3, 3, F), dtype=torch.float64, device="cuda")
    hfd = B.charge_flux(vel, mass, q, off, st, 3, 1e-5, 1.6e-19, ctx=ctx, out=fdev, async_=True)
    assert ctx.pending() == 6
    ctx.sync()
    np.testing.assert_array_equal(hf.wait(), ref_f)
    np.testing.assert_array_equal(hfd.wait().cpu().numpy(), ref_f)
    for a, b in zip(hc.wait(), ref_c):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(hx.wait(), ref_x)
    np.testing.assert_array_equal(hd.wait(), ref_d)
    np.testing.assert_array_equal(hi.wait(), ref_i)


def test_green_kubo_chain_equals_the_separate_calls(env, g_acf):
    """mdhip_green_kubo = xcorr -> x c^2 -> cumtrapz -> x V/(kB T) -> mean, every factor one rounding as on the host:
    bit-identical to the separate calls + numpy, both estimators, with and without the leading zero, large (page-locked
    results, DMA) and small; and the reference's own viscosity chain on the golden series."""
    B, synth, torch, ctx = env
    rng = np.random.default_rng(21)
    c2, dx, vk = 101325.0 ** 2, 2e-15, 3.7e9
    for n, method in ((300000, B.XCORR_FFT), (5000, B.XCORR_FFT), (5000, B.XCORR_DIRECT)):
        s = rng.standard_normal((3, n)) * 100.0
        for lead in (False, True):
            acf_ref = B.xcorr(s, method=method, ctx=ctx) * c2
            int_ref = np.multiply(vk, B.cumtrapz(acf_ref, dx, leading_zero=lead, ctx=ctx))
            mean_ref = np.mean(int_ref, axis=0)
            for dev in (False, True):
                a = torch.from_numpy(s).cuda() if dev else s
                acf, integral, mean = B.green_kubo(a, method=method, acf_scale=c2, dx=dx, integral_scale=vk,
                                                   leading_zero=lead, want_mean=True, ctx=ctx)
                np.testing.assert_array_equal(acf, acf_ref)
                np.testing.assert_array_equal(integral, int_ref)
                np.testing.assert_array_equal(mean, mean_ref)
        h = B.green_kubo(s, method=method, acf_scale=c2, dx=dx, integral_scale=vk, want_acf=False, ctx=ctx, async_=True)
        acf, integral, mean = h.wait()
        assert acf is None and mean is None
        np.testing.assert_array_equal(integral, np.multiply(vk, B.cumtrapz(B.xcorr(s, method=method, ctx=ctx) * c2, dx, ctx=ctx)))
    # cross-correlation with b != a (the conductivity chain's form)
    a, b = rng.standard_normal((4, 3000)), rng.standard_normal((4, 3000))
    acf, integral, _ = B.green_kubo(a, b, acf_scale=1.0, dx=1e-15, leading_zero=True, ctx=ctx)
    np.testing.assert_array_equal(acf, B.xcorr(a, b, ctx=ctx))
    np.testing.assert_array_equal(integral, B.cumtrapz(acf, 1e-15, leading_zero=True, ctx=ctx))
    # the reference's numbers (tests/golden/acf.npz: Viscosity._calc_3d_visc on the AR(1) series, units "real")
    from mdproptools_amd.common import constants as K

    ser = np.ascontiguousarray(g_acf["pressure"])
    dt = float(g_acf["visc_step"][1] - g_acf["visc_step"][0]) * float(g_acf["visc_timestep"]) * K.TIME_CONVERSION["real"]
    pre = float(g_acf["visc_volume"]) * K.DISTANCE_CONVERSION["real"] ** 3 / (K.BOLTZMANN * float(g_acf["visc_temp"]))
    acf, integral, mean = B.green_kubo(ser, acf_scale=K.PRESSURE_CONVERSION["real"] ** 2, dx=dt, integral_scale=pre,
                                       want_mean=True, ctx=ctx)
    np.testing.assert_allclose(acf, g_acf["visc_acf"], rtol=0, atol=1e-10 * g_acf["visc_acf"].max())
    scale = np.abs(g_acf["visc_data"]).max()
    np.testing.assert_allclose(integral, g_acf["visc_data"], rtol=1e-9, atol=1e-10 * scale)
    np.testing.assert_allclose(mean, g_acf["visc_avg"], rtol=1e-9, atol=1e-10 * scale)


def test_errors_and_mixing_sync_with_async(env):
    B, synth, torch, ctx = env
    n, F, L = 5000, 8, 40.0
    x = _frames(synth, torch, n, F, L, 77)
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    ref = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=ctx)
    # an argument error is reported by the call itself and leaves nothing in flight
    with pytest.raises(Exception):
        B.rdf_loop(x, ty, box, rel, 10.0, -0.05, 200, per_frame=False, ctx=ctx, async_=True)
    assert ctx.pending() == 0
    # a synchronous call completes what was issued before it
    h = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=ctx, async_=True)
    s = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=True, ctx=ctx)
    assert ctx.pending() == 0
    g = h.wait()
    np.testing.assert_array_equal(g[0], ref[0])
    np.testing.assert_array_equal(s[0].sum(axis=0), ref[0])
    # an error found at completion (a single frame that cannot pass the lowered guard) comes out of the wait
    from mdproptools_amd._lib import Context, MdhipError

    c2 = Context(0)
    try:
        c2.set_option("rdf_guard", 1)
        h = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=c2, async_=True)
        with pytest.raises(MdhipError, match="overflow the 32-bit"):
            h.wait()
        c2.set_option("rdf_guard", 0)
        g = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=c2, async_=True).wait()
        np.testing.assert_array_equal(g[0], ref[0])
    finally:
        c2.close()


def test_blocking_wait_option_gives_the_same_results(env):
    B, synth, torch, ctx = env
    n, F, L = 5000, 8, 40.0
    x = _frames(synth, torch, n, F, L, 78)
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    ref = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=ctx)
    ctx.set_option("sync_spin", 0)
    try:
        a = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=ctx)
        b = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=ctx, async_=True).wait()
    finally:
        ctx.set_option("sync_spin", 1)
    np.testing.assert_array_equal(a[0], ref[0])
    np.testing.assert_array_equal(b[1], ref[1])


def test_host_resident_frames_pipelined(env):
    """Asynchronous calls on frames in (page-locked or pageable) HOST memory: the copy of call k + 1 goes into the
    staging buffer call k does not use and runs under call k's kernels. Different data every call, several in flight,
    a synchronous call in between — every result equals the synchronous one."""
    B, synth, torch, ctx = env
    n, F, L = 5000, 16, 40.0
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    cuts = synth.cn_cutoffs(len(rel))
    hosts = []
    for k in range(6):
        x = synth.rdf_frames(n, range(F), L, 90 + k)
        if k % 2 == 0:  # page-locked
            t = torch.empty(x.shape, dtype=torch.float64, pin_memory=True)
            t.numpy()[...] = x
            hosts.append(t.numpy())
        else:
            hosts.append(x)
    ref = [B.rdf_loop(torch.from_numpy(np.ascontiguousarray(x)).cuda(), ty, box, rel, 10.0, 0.05, 200, per_frame=False,
                      ctx=ctx) for x in hosts]
    hs = [B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=ctx, async_=True) for x in hosts[:4]]
    mid = B.rdf_loop(hosts[4], ty, box, rel, 10.0, 0.05, 200, per_frame=True, ctx=ctx)  # synchronous: drains first
    assert ctx.pending() == 0
    hs += [B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=ctx, async_=True) for x in hosts[4:]]
    hc = B.cn_loop(hosts[5], ty, box, rel, cuts, per_frame=False, ctx=ctx, async_=True)
    for h, r in zip(hs, ref):
        g = h.wait()
        np.testing.assert_array_equal(g[0], r[0])
        np.testing.assert_array_equal(g[1], r[1])
    np.testing.assert_array_equal(mid[0].sum(axis=0), ref[4][0])
    np.testing.assert_array_equal(hc.wait(), B.cn_loop(hosts[5], ty, box, rel, cuts, per_frame=False, ctx=ctx))


def test_completion_error_goes_to_its_own_handle_whoever_completed_the_call(env):
    """a FAILED asynchronous call completed on behalf of a later one (a synchronous call that drains it, the
    wait of a later handle) is reported by ITS handle's wait() — not returned as zero-filled arrays — and the handle whose
    call went through returns its result. (The failing call: a pair sweep whose single frames cannot pass a lowered
    overflow guard — found at completion; the calls around it: cumulative trapezoids, which have no such guard.)"""
    import warnings

    B, synth, torch, _ctx = env
    from mdproptools_amd._lib import Context, MdhipError

    n, F, L = 5000, 8, 40.0
    x = _frames(synth, torch, n, F, L, 79)
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    y = np.random.default_rng(3).normal(size=(2, 50_000))
    c2 = Context(0)
    try:
        ref = B.cumtrapz(y, 0.5, ctx=c2)
        c2.set_option("rdf_guard", 1)
        # (a) drained by a synchronous call
        bad = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=c2, async_=True)
        good = B.cumtrapz(y, 0.5, ctx=c2)  # completes `bad` first
        assert c2.pending() == 0
        np.testing.assert_array_equal(good, ref)
        with pytest.raises(MdhipError, match="overflow the 32-bit"):
            bad.wait()
        c2.sync()  # its owner has been told: not reported a second time
        # (b) completed by the wait of a LATER handle
        bad = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=c2, async_=True)
        later = B.cumtrapz(y, 0.5, ctx=c2, async_=True)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            g = later.wait()
        np.testing.assert_array_equal(g, ref)
        assert any("earlier asynchronous mdhip call failed" in str(m.message) for m in w)
        with pytest.raises(MdhipError, match="overflow the 32-bit"):
            bad.wait()
        # (c) nobody holds the failed call's handle: the next sync reports it
        B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=c2, async_=True)
        with pytest.raises(MdhipError, match="overflow the 32-bit"):
            c2.sync()
        c2.set_option("rdf_guard", 0)
        np.testing.assert_array_equal(B.cumtrapz(y, 0.5, ctx=c2, async_=True).wait(), ref)
    finally:
        c2.close()


def test_dropped_handles_leave_nothing_dangling(env):
    """the arrays of an un-waited call belong to the context until the call has completed — dropping the
    handle and then completing the call through sync(), a later wait or close() writes into memory that is still there."""
    import gc

    B, synth, torch, _ctx = env
    from mdproptools_amd._lib import Context

    n, F, L = 5000, 8, 40.0
    x = _frames(synth, torch, n, F, L, 80)
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    c2 = Context(0)
    try:
        ref = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=c2)
        for _ in range(3):
            B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=c2, async_=True)  # handle dropped
        gc.collect()
        assert len(c2._live) == 3 and c2.pending() == 3
        junk = [np.full(200 * 11, 7, dtype=np.uint64) for _ in range(16)]  # (whatever freed result memory would be reused for)
        c2.sync()
        assert len(c2._live) == 0
        assert all(int(j.min()) == 7 and int(j.max()) == 7 for j in junk)
        h1 = B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=c2, async_=True)
        B.rdf_loop(x, ty, box, rel, 10.0, 0.05, 200, per_frame=False, ctx=c2, async_=True)  # dropped, behind h1
        gc.collect()
        np.testing.assert_array_equal(h1.wait()[0], ref[0])
        assert len(c2._live) == 1
    finally:
        c2.close()  # completes the dropped call into arrays the context still holds
    assert c2._live == {}


def test_lag_status_word_on_the_device(env):
    """mdhip_lag_msd_status_dev: the status of the lag call issued last, as a number in device memory behind the call's
    kernels — the spectral path's bound for diffusive data, a bound above the tolerance for ballistic data (where the
    library itself then answers with the exact kernel at completion), 0 when the exact kernel was asked for."""
    B, synth, torch, ctx = env
    F, E = 400, 3000
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    r = torch.cumsum(torch.randn((F, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
    t = torch.arange(F, device="cuda", dtype=torch.float64)[:, None, None]
    rb = (t * torch.rand((1, 3, E), generator=g, device="cuda", dtype=torch.float64) + 500.0).contiguous()
    out = torch.empty((F, 1, 4), dtype=torch.float64, device="cuda")
    st = torch.full((1,), -1.0, dtype=torch.float64, device="cuda")
    B.lag_msd(r, F - 1, [0, E], out=out, ctx=ctx, async_=True, status_out=st).wait()
    assert float(st.cpu()[0]) == ctx.last_rel_bound() and 0.0 < ctx.last_rel_bound() <= 1e-10
    B.lag_msd(rb, F - 1, [0, E], out=out, ctx=ctx, async_=True, status_out=st).wait()
    assert float(st.cpu()[0]) > 1e-10 and ctx.last_rel_bound() <= 1e-10  # (the difference form answered at completion: for the
    # whole call, or for the few lags that missed the bound)
    ctx.set_option("lag_variant", 1)
    try:
        B.lag_msd(r, F - 1, [0, E], out=out, ctx=ctx, async_=True, status_out=st).wait()
    finally:
        ctx.set_option("lag_variant", -1)
    assert float(st.cpu()[0]) == 0.0


def test_fused_step_redoes_the_lag_part_together_when_the_status_says_so():
    """dist.msd_step_sharded_async (one host wait): the lag call's status word travels through the all-reduce; a bound
    above the tolerance (ballistic motion) makes every rank repeat the lag part with the exact-difference kernel. One rank
    behind a real process group, in a process of its own; against the plain library calls."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %r)
from mdproptools_amd import backend as B, dist as D
from mdproptools_amd._lib import default_context
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = default_context(0)
F, E = 400, 3000
g = torch.Generator(device="cuda"); g.manual_seed(3)
walk = torch.cumsum(torch.randn((F, 3, E), generator=g, device="cuda", dtype=torch.float64) * 0.1, dim=0)
t = torch.arange(F, device="cuda", dtype=torch.float64)[:, None, None]
ball = (t * torch.rand((1, 3, E), generator=g, device="cuda", dtype=torch.float64) + 500.0).contiguous()
goff = [0, 1000, E]
for name, r in (("walk", walk), ("ballistic", ball)):
    ref_l = B.lag_msd(r, F - 1, goff, scale=1.0, ctx=ctx)
    ref_o = B.msd_origin(r, r[0], goff, scale=1e-10, ctx=ctx)
    single, win, lag, stats = D.msd_step_sharded(r, r, F, (0, E), goff, 4, scale=1e-10, lag_scale=1.0, ctx=ctx)
    assert ("lag_redone" in stats) == (name == "ballistic"), (name, stats.keys())
    np.testing.assert_allclose(single, ref_o, rtol=1e-12)
    np.testing.assert_allclose(lag, ref_l, rtol=1e-9 if name == "walk" else 1e-12, atol=0)
print("STEP OK")
dist.destroy_process_group()
''' % root
    r = subprocess.run([sys.executable, "-c", code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert r.returncode == 0 and "STEP OK" in r.stdout, r.stdout[-3000:]
