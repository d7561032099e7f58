"""
GPU end-to-end: the drop-in functions/classes (same names and arguments as the reference) run on
text dumps/logs rebuilt from the golden inputs and are compared with the DataFrames/arrays the real
reference produced for those inputs (oracle/make_golden.py).
"""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

MASS = [16.000, 12.010, 1.008, 14.010, 32.060, 16.000, 12.010, 19.000, 24.305]


def _write(tmp, g, columns=None):
    from mdproptools_amd import io as mio

    cols = list(g["columns"]) if columns is None else columns
    for s, b, t in zip(g["steps"], g["bounds"], g["frames"]):
        mio.write_dump(os.path.join(tmp, "dump.nvt.%d.dump" % s), s, b, cols, t)
    return os.path.join(tmp, "dump.nvt.*.dump")


@pytest.fixture(scope="module")
def c1_dir(tmp_path_factory):
    g = load_golden("c1_rdf.npz")
    tmp = str(tmp_path_factory.mktemp("c1"))
    return g, _write(tmp, g), tmp


def test_calc_atomic_rdf_and_cn(c1_dir):
    from mdproptools_amd.structural.rdf_cn import calc_atomic_cn, calc_atomic_rdf

    g, pat, tmp = c1_dir
    out = os.path.join(tmp, "rdf.csv")
    df = calc_atomic_rdf(20, 0.05, 9, MASS, g["rdf_def_rel"].tolist(), pat, path_or_buff=out)
    assert list(df.columns) == list(g["rdf_def_df_columns"])
    np.testing.assert_array_equal(df.to_numpy(), g["rdf_def_df"])  # bit-identical g(r)
    np.testing.assert_allclose(pd.read_csv(out).to_numpy(), g["rdf_def_df"], rtol=1e-13)  # csv text round trip
    df = calc_atomic_rdf(20, 0.05, 9, MASS, g["rdf_alt_rel"].tolist(), pat, num_mols=g["num_mols"].tolist(),
                         num_atoms_per_mol=g["num_atoms_per_mol"].tolist(), save_mode=False)
    np.testing.assert_array_equal(df.to_numpy(), g["rdf_alt_df"])
    assert list(df.columns)[2:] == ["g_32-17", "g_32-32"]
    cn = calc_atomic_cn(g["cn_def_cut"].tolist(), 0.05, 9, MASS, g["cn_def_rel"].tolist(), pat, save_mode=False)
    np.testing.assert_array_equal(cn.to_numpy(), g["cn_def_df"])
    assert list(cn.columns) == ["cn_9-1", "cn_9-4", "cn_9-6", "cn_9-9"]
    cn = calc_atomic_cn(g["cn_alt_cut"].tolist(), 0.05, 9, MASS, g["rdf_alt_rel"].tolist(), pat,
                        num_mols=g["num_mols"].tolist(), num_atoms_per_mol=g["num_atoms_per_mol"].tolist(),
                        save_mode=False)
    np.testing.assert_array_equal(cn.to_numpy(), g["cn_alt_df"])


def test_calc_molecular_rdf_and_cn(c1_dir):
    from mdproptools_amd.structural.rdf_cn import calc_molecular_cn, calc_molecular_rdf

    g, pat, tmp = c1_dir
    nm, na = g["num_mols"].tolist(), g["num_atoms_per_mol"].tolist()
    df = calc_molecular_rdf(20, 0.05, 9, MASS, g["mol_rel"].tolist(), pat, nm, na, save_mode=False)
    # COM sums run in a different order on the GPU (rtol 1e-13 on the sites); the histograms of these
    # frames are nevertheless identical, so g(r) is too
    np.testing.assert_allclose(df.to_numpy(), g["mol_rdf_df"], rtol=1e-12, atol=0)
    assert list(df.columns) == ["r ($\\AA$)", "g_9-1", "g_9-2", "g_4-3"]
    cn = calc_molecular_cn(g["mol_cn_cut"].tolist(), 0.05, 9, MASS, g["mol_rel"].tolist(), pat, nm, na,
                           save_mode=False)
    np.testing.assert_allclose(cn.to_numpy(), g["mol_cn_df"], rtol=1e-12, atol=0)


def test_rdf_errors(c1_dir):
    from mdproptools_amd.structural.rdf_cn import calc_atomic_rdf

    g, pat, tmp = c1_dir
    with pytest.raises(ValueError):
        calc_atomic_rdf(20, 0.05, 8, MASS[:8], [[9], [1]], pat, save_mode=False)  # wrong num_types


@pytest.fixture(scope="module")
def small_dir(tmp_path_factory):
    g = load_golden("small_md.npz")
    tmp = str(tmp_path_factory.mktemp("small"))
    _write(tmp, g)
    return g, tmp


def _frame_equal(df, values, columns, rtol):
    assert list(df.columns) == list(columns), (list(df.columns), list(columns))
    assert df.shape == values.shape
    np.testing.assert_allclose(df.to_numpy(dtype=np.float64), values, rtol=rtol, atol=0)


def test_diffusion_allatom(small_dir):
    from mdproptools_amd.dynamical.diffusion import Diffusion

    g, tmp = small_dir
    d = Diffusion(timestep=1, units="real", outputs_dir=tmp, diff_dir=tmp)
    msd, msd_all, msd_int = d.get_msd_from_dump("dump.nvt.*.dump", msd_type="allatom", avg_interval=True,
                                                tao_coeff=4)
    _frame_equal(msd, g["aa_msd"], g["aa_msd_cols"], 1e-10)
    _frame_equal(msd_all, g["aa_msd_all"], g["aa_msd_all_cols"], 1e-12)
    _frame_equal(msd_int, g["aa_msd_int"], g["aa_msd_int_cols"], 1e-10)
    assert msd_all["id"].dtype == np.int64 and msd_int["id"].dtype == np.int64  # as the reference's parser reads ids
    two = d.get_msd_from_dump("dump.nvt.*.dump", msd_type="allatom")
    assert len(two) == 2
    with pytest.raises(ValueError):
        d.get_msd_from_dump("dump.nvt.*.dump", msd_type="nonsense")


@pytest.mark.parametrize("tag,drift", [("com", False), ("comd", True)])
def test_diffusion_com(small_dir, tag, drift):
    from mdproptools_amd.dynamical.diffusion import Diffusion

    g, tmp = small_dir
    d = Diffusion(timestep=1, units="real", outputs_dir=tmp, diff_dir=tmp)
    msd, msd_all, msd_int = d.get_msd_from_dump(
        "dump.nvt.*.dump", msd_type="com", num_mols=g["num_mols"].tolist(),
        num_atoms_per_mol=g["num_atoms_per_mol"].tolist(), mass=MASS, com_drift=drift, avg_interval=True,
        tao_coeff=4)
    _frame_equal(msd, g[tag + "_msd"], g[tag + "_msd_cols"], 1e-9)
    _frame_equal(msd_all, g[tag + "_msd_all"], g[tag + "_msd_all_cols"], 1e-9)
    _frame_equal(msd_int, g[tag + "_msd_int"], g[tag + "_msd_int_cols"], 1e-9)
    if drift:
        table = d.calc_diff(msd, diff_names=["dme", "tfsi", "mg"])
        np.testing.assert_allclose(table.to_numpy(), g["comd_diff"], rtol=1e-9)
        assert os.path.exists(os.path.join(tmp, "diffusion.csv"))


@pytest.mark.parametrize("kw", [dict(msd_type="allatom"), dict(msd_type="com", mass=MASS),
                                dict(msd_type="com", mass=None), dict(msd_type="com", mass=MASS, com_drift=True)])
def test_diffusion_streamed_equals_load_all(small_dir, kw, monkeypatch):
    """get_msd_from_dump on the frame stream (text -> page-locked batches -> device-resident trajectory, several
    batches here) returns the same DataFrames, bit for bit, as the load-everything-first route."""
    from mdproptools_amd.dynamical import diffusion as dm

    g, tmp = small_dir
    if kw["msd_type"] == "com":
        kw = dict(kw, num_mols=g["num_mols"].tolist(), num_atoms_per_mol=g["num_atoms_per_mol"].tolist())
    d = dm.Diffusion(timestep=1, units="real", outputs_dir=tmp, diff_dir=tmp)
    n = g["frames"][0].shape[0]
    res = {}
    for on in (True, False):
        monkeypatch.setattr(dm, "STREAM", on)
        monkeypatch.setattr(dm, "STREAM_BATCH_BYTES", 2 * 24 * n)  # two frames per batch
        made = []
        orig = dm.Diffusion._entity_frames_streamed

        def spy(self, *a, **k):
            out = orig(self, *a, **k)
            made.append(out)
            return out

        monkeypatch.setattr(dm.Diffusion, "_entity_frames_streamed", spy)
        res[on] = d.get_msd_from_dump("dump.nvt.*.dump", avg_interval=True, tao_coeff=3, **kw)
        monkeypatch.setattr(dm.Diffusion, "_entity_frames_streamed", orig)
        if on:
            assert made and made[0] is not None and made[0][1].is_cuda  # the device-resident route was taken
        else:
            assert not made
    for a, b in zip(res[True], res[False]):
        assert list(a.columns) == list(b.columns)
        np.testing.assert_array_equal(a.to_numpy(), b.to_numpy())


def test_diffusion_streamed_time_order(tmp_path, monkeypatch):
    """Files whose numeric order is not the time order (the reference sorts its (time, id) index): the streamed route
    reorders the device-resident trajectory and returns what the load-everything route returns."""
    from mdproptools_amd import io as mio
    from mdproptools_amd.dynamical import diffusion as dm

    rng = np.random.default_rng(3)
    n, steps = 400, [300, 0, 200, 100, 400]  # file k holds timestep steps[k]
    walk = {s: rng.uniform(0, 20, (n, 3)) + 0.01 * s * rng.normal(0, 1, (n, 3)) for s in sorted(steps)}
    for k, s_ in enumerate(steps):
        perm = rng.permutation(n)
        tbl = np.column_stack([perm + 1, 1 + perm % 2, walk[s_][perm]])
        mio.write_dump(str(tmp_path / ("dump.nvt.%d.dump" % k)), s_, [[0, 20.0]] * 3, ["id", "type", "xu", "yu", "zu"], tbl)
    d = dm.Diffusion(timestep=1, units="real", outputs_dir=str(tmp_path), diff_dir=str(tmp_path))
    res = {}
    for on in (True, False):
        monkeypatch.setattr(dm, "STREAM", on)
        monkeypatch.setattr(dm, "STREAM_BATCH_BYTES", 2 * 24 * n)
        res[on] = d.get_msd_from_dump("dump.nvt.*.dump", msd_type="allatom", avg_interval=True, tao_coeff=2)
    for a, b in zip(res[True], res[False]):
        np.testing.assert_array_equal(a.to_numpy(), b.to_numpy())
    t = res[True][0]["Time (s)"].to_numpy()
    assert (np.diff(t) > 0).all() and res[True][0]["msd"].iloc[0] == 0.0


def test_calc_com_dataframe(small_dir):
    from mdproptools_amd import io as mio
    from mdproptools_amd.common.com_mols import calc_com

    g, tmp = small_dir
    (dump,) = list(mio.parse_lammps_dumps(os.path.join(tmp, "dump.nvt.0.dump")))
    dump.data = dump.data.sort_values(by=["id"]).reset_index()
    nm, na = g["num_mols"].tolist(), g["num_atoms_per_mol"].tolist()
    com = calc_com(dump, nm, na, MASS, atom_attributes=["xu", "yu", "zu"]).reset_index()
    _frame_equal(com, g["calc_com_xu"], g["calc_com_xu_cols"], 1e-13)
    comv = calc_com(dump, nm, na, None, atom_attributes=["vx", "vy", "vz"], calc_charge=True).reset_index()
    assert list(comv.columns) == list(g["calc_com_v_cols"])
    np.testing.assert_allclose(comv.to_numpy(dtype=np.float64), g["calc_com_v"], rtol=1e-12, atol=1e-12)


def test_conductivity_chain(small_dir):
    from mdproptools_amd.dynamical.conductivity import Conductivity

    g, tmp = small_dir
    c = Conductivity("dump.nvt.*.dump", g["num_mols"].tolist(), g["num_atoms_per_mol"].tolist(),
                     float(g["cond_volume"]), mass=MASS, temp=298.15, timestep=1, units="real", working_dir=tmp)
    j = c.get_charge_flux()
    np.testing.assert_allclose(j, g["cond_j"], rtol=1e-9, atol=1e-25)
    np.testing.assert_allclose(c.time, g["cond_time"], rtol=1e-15)
    tot = c.correlate_charge_flux(g["cond_j"])
    np.testing.assert_allclose(tot, g["cond_tot_flux"], rtol=0, atol=1e-10 * abs(g["cond_tot_flux"]).max())
    integ = c.integrate_charge_flux_correlation(g["cond_tot_flux"])
    np.testing.assert_allclose(integ, g["cond_integral"], rtol=1e-9, atol=1e-12 * abs(g["cond_integral"]).max())
    np.testing.assert_allclose(c.green_kubo(g["cond_integral"][:, -1]), g["cond_gk"], rtol=1e-14)
    a, b = g["cond_j"][0, 1], g["cond_j"][0, 2]
    ref = np.array([np.dot(a[k:], b[: len(b) - k]) / (len(a) - k) for k in range(len(a))])
    np.testing.assert_allclose(Conductivity.correlate(a, b), ref, rtol=0, atol=1e-10 * abs(ref).max())


def test_conductivity_flux_streamed_equals_load_all(small_dir, monkeypatch):
    """get_charge_flux on the frame stream (several batches) == the load-everything-first route, bit for bit."""
    from mdproptools_amd import stream as S
    from mdproptools_amd.dynamical import conductivity as cm

    g, tmp = small_dir
    n = g["frames"][0].shape[0]
    res = {}
    for on in (True, False):
        monkeypatch.setattr(cm, "STREAM", on)
        monkeypatch.setattr(S, "DEFAULT_BATCH_BYTES", 2 * 24 * n)
        orig = S.FrameStream.__init__

        def small_batches(self, *a, **k):
            k["batch_bytes"] = 2 * 24 * n  # two frames per batch
            orig(self, *a, **k)

        monkeypatch.setattr(S.FrameStream, "__init__", small_batches)
        for mass in (MASS, None):
            c = cm.Conductivity("dump.nvt.*.dump", g["num_mols"].tolist(), g["num_atoms_per_mol"].tolist(), 1000.0,
                                mass=mass, temp=300.0, timestep=1, units="real", working_dir=tmp)
            res[(on, mass is None)] = (c.get_charge_flux(), np.asarray(c.time))
        monkeypatch.setattr(S.FrameStream, "__init__", orig)
    for key in (False, True):
        np.testing.assert_array_equal(res[(True, key)][0], res[(False, key)][0])
        np.testing.assert_array_equal(res[(True, key)][1], res[(False, key)][1])
    assert np.abs(res[(True, False)][0]).max() > 0


def test_conductivity_fit_curve():
    from mdproptools_amd.dynamical.conductivity import Conductivity

    g = load_golden("host_logic.npz")
    c = Conductivity.__new__(Conductivity)
    c.num_mols = [1]
    c.time = list(np.arange(g["dtr_flux"].shape[1]) * 2e-15)
    c.temp, c.volume = float(g["fc_temp"]), float(g["fc_volume"])
    integ = c.integrate_charge_flux_correlation(g["dtr_flux"])
    np.testing.assert_allclose(integ, g["fc_integral"], rtol=1e-9, atol=1e-12 * abs(g["fc_integral"]).max())
    ave, tr = c.fit_curve(g["dtr_flux"], integ, float(g["dtr_tol"]))
    np.testing.assert_allclose(ave, g["fc_ave"], rtol=1e-9)
    np.testing.assert_allclose(np.array([list(r) for r in tr]), g["fc_time_range"], rtol=1e-15)
    np.testing.assert_allclose(c.green_kubo(ave), g["fc_cond"], rtol=1e-9)


def test_viscosity_3d_and_replicates(tmp_path):
    from mdproptools_amd import io as mio
    from mdproptools_amd.dynamical.viscosity import Viscosity

    g = load_golden("acf.npz")
    p = g["pressure"]
    v = Viscosity("log.*", 0, float(g["visc_volume"]), temp=float(g["visc_temp"]), timestep=int(g["visc_timestep"]),
                  acf_method="wkt", units="real")
    log_df = pd.DataFrame({"Step": g["visc_step"], "Pxy": p[0], "Pxz": p[1], "Pyz": p[2]})
    avg, data, acf = v._calc_3d_visc(log_df)
    scale = abs(g["visc_avg"]).max()
    np.testing.assert_allclose(acf, g["visc_acf"], rtol=0, atol=1e-10 * g["visc_acf"].max())
    np.testing.assert_allclose(data, g["visc_data"], rtol=1e-9, atol=1e-10 * scale)
    np.testing.assert_allclose(avg, g["visc_avg"], rtol=1e-9, atol=1e-10 * scale)
    v.acf_method = "brute_force"
    avg_b, _, _ = v._calc_3d_visc(log_df)
    np.testing.assert_allclose(avg_b, g["visc_avg_brute"], rtol=1e-9, atol=1e-10 * scale)
    for k in range(3):
        np.testing.assert_allclose(Viscosity.autocorrelate(p[k], "wkt"), g["acf_wkt"][k], rtol=0,
                                   atol=1e-10 * g["acf_wkt"][k][0])
    with pytest.raises(ValueError):
        Viscosity.autocorrelate(p[0], "fourier")
    with pytest.raises(KeyError):
        Viscosity("log.*", 0, 1.0, units="furlongs")._calc_3d_visc(log_df)

    h = load_golden("host_logic.npz")
    for r in range(h["log_press"].shape[0]):
        tbl = np.column_stack([h["log_step"], h["log_press"][r].T])
        mio.write_log(tmp_path / ("log.rep%d" % r), tbl, ["Step", "Pxy", "Pxz", "Pyz"])
    v = Viscosity("log.rep*", int(h["visc_cutoff"]), float(h["visc_volume"]), temp=float(h["visc_temp"]),
                  timestep=1, acf_method="wkt", units="real", working_dir=str(tmp_path))
    visc_avg, visc_data, acf_data, tm = v.calc_avg_visc(output_all_data=True)
    import glob

    order = [int(os.path.basename(f)[7:]) for f in glob.glob(str(tmp_path / "log.rep*"))]
    expect = {int(o): row for o, row in zip(h["visc_rep_order"], h["visc_avg"])}
    for o, row in zip(order, visc_avg):
        np.testing.assert_allclose(row, expect[o], rtol=1e-8, atol=1e-10 * abs(expect[o]).max())
    np.testing.assert_array_equal(tm, h["visc_time"])
    fit = v.fit_avg_visc([expect[o] for o in sorted(expect)], initial_guess=[1e-8, 0.5, 50.0, 500.0])
    ref_fit = float(h["visc_fit"])
    assert np.isfinite(ref_fit) and abs(fit - ref_fit) <= 1e-6 * abs(ref_fit)


def test_native_and_pandas_readers_give_identical_results(small_dir, tmp_path):
    """The drop-in layer reads dumps through the native reader by default; the pandas-based reader (what
    the reference sees) must give the same frames. Also covers dumps without xu/yu/zu (x + ix*L)."""
    from mdproptools_amd import io as mio
    from mdproptools_amd.dynamical.diffusion import Diffusion

    g, tmp = small_dir
    cols = list(g["columns"])
    keep = ["id", "type", "mass", "q", "x", "y", "z", "ix", "iy", "iz"]
    for s, b, t in zip(g["steps"], g["bounds"], g["frames"]):
        L = b[:, 1] - b[:, 0]
        img = np.rint((t[:, [cols.index(c) for c in ("xu", "yu", "zu")]]
                       - t[:, [cols.index(c) for c in ("x", "y", "z")]]) / L)
        tbl = np.column_stack([t[:, [cols.index(c) for c in keep[:7]]], img])
        mio.write_dump(str(tmp_path / ("dump.nvt.%d.dump" % s)), s, b, keep, tbl)
    d = Diffusion(outputs_dir=str(tmp_path), diff_dir=str(tmp_path))
    res = {}
    for native in (True, False):
        mio.USE_NATIVE_READER = native
        try:
            res[native] = d.get_msd_from_dump("dump.nvt.*.dump", msd_type="com", num_mols=g["num_mols"].tolist(),
                                              num_atoms_per_mol=g["num_atoms_per_mol"].tolist(), avg_interval=True)
        finally:
            mio.USE_NATIVE_READER = True
    for a, b in zip(res[True], res[False]):
        assert list(a.columns) == list(b.columns)
        np.testing.assert_array_equal(a.to_numpy(), b.to_numpy())
    # the rebuilt unwrapped coordinates differ from the dumped xu only by the 6-digit rounding of the dump text
    np.testing.assert_allclose(res[True][0].to_numpy(), g["com_msd"], rtol=1e-3)


# ------------------------------------------------------------------ drop-in under torch.distributed, on the GPU
def _gpu_dist_worker(rank, world, port, tmp_dir):
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist

    from mdproptools_amd.structural import rdf_cn

    if world > 1:  # two ranks share the one GPU of the test box: gloo for the (host-side) collectives
        dist.init_process_group("gloo", rank=rank, world_size=world)
    out = os.path.join(tmp_dir, "w%d" % world)
    os.makedirs(out, exist_ok=True)
    pattern = os.path.join(tmp_dir, "dump.nvt.*.dump")
    g = rdf_cn.calc_atomic_rdf(6.0, 0.05, 3, [1.0, 2.0, 3.0], [[1, 1, 2], [1, 2, 3]], pattern,
                               path_or_buff=os.path.join(out, "rdf.csv"))
    c = rdf_cn.calc_atomic_cn([2.0, 3.0, 4.5], 0.05, 3, [1.0, 2.0, 3.0], [[1, 1, 2], [1, 2, 3]], pattern,
                              path_or_buff=os.path.join(out, "cn.csv"))
    # the streamed Diffusion route (device-resident trajectory; the ranks' blocks are all-gathered)
    from mdproptools_amd.dynamical.diffusion import Diffusion

    d = Diffusion(timestep=1, units="real", outputs_dir=tmp_dir, diff_dir=out)
    aa = d.get_msd_from_dump("msd.*.dump", msd_type="allatom", avg_interval=True, tao_coeff=2)
    cm = d.get_msd_from_dump("msd.*.dump", msd_type="com", num_mols=[500, 500], num_atoms_per_mol=[4, 2],
                             mass=[1.0, 2.0, 3.0], avg_interval=True, tao_coeff=2)
    cd = d.get_msd_from_dump("msd.*.dump", msd_type="com", num_mols=[500, 500], num_atoms_per_mol=[4, 2],
                             mass=[1.0, 2.0, 3.0], com_drift=True, avg_interval=True, tao_coeff=3)
    from mdproptools_amd.dynamical import diffusion as dmod

    dmod.MSD_ALL_ON_EVERY_RANK = False  # every rank keeps the msd_all rows of ITS frames
    try:
        own = d.get_msd_from_dump("msd.*.dump", msd_type="allatom")[1]
    finally:
        dmod.MSD_ALL_ON_EVERY_RANK = True
    np.savez(os.path.join(out, "rank%d.npz" % rank), g=g.to_numpy(), c=c.to_numpy(), own_all=own.to_numpy(),
             **{"aa%d" % k: v.to_numpy() for k, v in enumerate(aa)}, **{"cm%d" % k: v.to_numpy() for k, v in enumerate(cm)},
             **{"cd%d" % k: v.to_numpy() for k, v in enumerate(cd)})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_dropin_rdf_cn_two_ranks_on_gpu(tmp_path):
    """calc_atomic_rdf / calc_atomic_cn with two processes (gloo) sharing the GPU: each parses its own files and
    runs its own frames through libmdhip.so; both return bit for bit the single-process DataFrames."""
    import socket

    import torch.multiprocessing as mp

    from mdproptools_amd import io as mio

    rng = np.random.default_rng(12)
    n = 3000
    for k in range(5):
        L = 30.0 + 0.2 * k
        tbl = np.column_stack([rng.permutation(n) + 1, 1 + (np.arange(n) % 3), rng.uniform(0, L, (n, 3))])
        mio.write_dump(str(tmp_path / ("dump.nvt.%d.dump" % (k * 100))), k * 100, [[0, L]] * 3,
                       ["id", "type", "x", "y", "z"], tbl)
    walk = rng.uniform(0, 30, (n, 3))
    for k in range(7):  # unwrapped coordinates of a random walk, one frame per file, seven files over two ranks
        perm = rng.permutation(n)  # rows in a different order in every file; an atom keeps its id, type and walk
        tbl = np.column_stack([perm + 1, 1 + (perm % 3), walk[perm]])
        mio.write_dump(str(tmp_path / ("msd.%d.dump" % (k * 100))), k * 100, [[0, 30.0]] * 3,
                       ["id", "type", "xu", "yu", "zu"], tbl)
        walk = walk + rng.normal(0, 0.3, (n, 3))

    def port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return s.getsockname()[1]

    mp.spawn(_gpu_dist_worker, args=(1, port(), str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_gpu_dist_worker, args=(2, port(), str(tmp_path)), nprocs=2, join=True)
    one = np.load(tmp_path / "w1" / "rank0.npz")
    assert abs(one["g"][60:, 1].mean() - 1.0) < 0.05  # ideal gas
    for rank in range(2):
        two = np.load(tmp_path / "w2" / ("rank%d.npz" % rank))
        np.testing.assert_array_equal(two["g"], one["g"])
        np.testing.assert_array_equal(two["c"], one["c"])
        # Diffusion under torch.distributed reduces every rank's frames where they are (origin broadcast, sums
        # gathered): msd and msd_all are the single-process frames bit for bit; msd_int's windows are summed rank by
        # rank (one all-reduce), so its last bits may differ
        for key in ("aa0", "aa1", "cm0", "cm1", "cd0", "cd1"):
            np.testing.assert_array_equal(two[key], one[key])
        for key in ("aa2", "cm2", "cd2"):
            np.testing.assert_allclose(two[key], one[key], rtol=1e-12)
        np.testing.assert_array_equal(two["own_all"], one["aa1"][{0: slice(0, 4 * 3000), 1: slice(4 * 3000, None)}[rank]])
    assert one["aa0"][-1, 4] > 0
    assert open(tmp_path / "w2" / "rdf.csv").read() == open(tmp_path / "w1" / "rdf.csv").read()


def test_calc_atomic_rdf_cn_one_pass(c1_dir, tmp_path):
    """calc_atomic_rdf_cn == (calc_atomic_rdf, calc_atomic_cn) on the mg_tfsi_dme frames, DataFrames bit for bit,
    default and altered ids; the two CSV files are the ones the separate calls write."""
    from mdproptools_amd.structural.rdf_cn import calc_atomic_cn, calc_atomic_rdf, calc_atomic_rdf_cn

    g, pat, tmp = c1_dir
    rel = [[9, 9, 9, 9, 1], [1, 4, 6, 9, 3]]
    cuts = [2.3, 2.3, 3.1, 6.0, 1.5]
    for kw in ({}, dict(num_mols=g["num_mols"].tolist(), num_atoms_per_mol=g["num_atoms_per_mol"].tolist())):
        r = rel if not kw else [[32, 32], [17, 32]]
        c = cuts if not kw else [2.4, 5.5]
        a = calc_atomic_rdf(20, 0.05, 9, MASS, r, pat, path_or_buff=str(tmp_path / "a.csv"), **kw)
        b = calc_atomic_cn(c, 0.05, 9, MASS, r, pat, path_or_buff=str(tmp_path / "b.csv"), **kw)
        g2, c2 = calc_atomic_rdf_cn(20, c, 0.05, 9, MASS, r, pat, rdf_path_or_buff=str(tmp_path / "g.csv"),
                                    cn_path_or_buff=str(tmp_path / "c.csv"), **kw)
        assert list(g2.columns) == list(a.columns) and list(c2.columns) == list(b.columns)
        np.testing.assert_array_equal(g2.to_numpy(), a.to_numpy())
        np.testing.assert_array_equal(c2.to_numpy(), b.to_numpy())
        assert open(tmp_path / "g.csv").read() == open(tmp_path / "a.csv").read()
        assert open(tmp_path / "c.csv").read() == open(tmp_path / "b.csv").read()


def test_calc_intermolecular_rdf(c1_dir):
    """Molecule-COM to molecule-COM g(r) (rdf_cn.py:857-903) against the real reference on two mg_tfsi_dme
    frames: num_types counts molecule types there while `mass` still lists atom masses."""
    from mdproptools_amd.structural.rdf_cn import calc_intermolecular_rdf

    g, pat, tmp = c1_dir
    ref = load_golden("inter_rdf.npz")
    df = calc_intermolecular_rdf(20, 0.05, 3, MASS, ref["rel"].tolist(), pat, g["num_mols"].tolist(),
                                 g["num_atoms_per_mol"].tolist(), save_mode=False)
    assert list(df.columns) == [str(c) for c in ref["columns"]]
    # COM sums run in another order on the GPU (rtol 1e-13 on the sites): a COM pair sitting within that of a bin
    # edge could move one count; on these frames none does
    np.testing.assert_allclose(df.to_numpy(), ref["df"], rtol=1e-12, atol=0)
