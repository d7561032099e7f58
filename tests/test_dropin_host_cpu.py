"""CPU-only: the host-side logic of the drop-in layer (no kernel calls) against the golden vectors."""
import inspect

import numpy as np
import pandas as pd
import pytest

from conftest import load_golden


def test_calc_atom_type_vectorised(g_synth):
    from mdproptools_amd.structural.rdf_cn import _calc_atom_type

    nm, na = g_synth["atom_type_num_mols"], g_synth["atom_type_num_atoms"]
    ids = np.arange(1, int(np.dot(nm, na)) + 1, dtype=np.float64)
    np.testing.assert_array_equal(_calc_atom_type(ids, nm, na), g_synth["atom_type_out"])
    # ids beyond the last molecule type are left untouched, as in the reference loop
    np.testing.assert_array_equal(_calc_atom_type(np.array([1000.0]), nm, na), [1000.0])


def test_normalisation_matches_reference_frame(g_c1):
    from mdproptools_amd.structural import rdf_cn as M

    g = g_c1
    rel = g["rdf_def_rel"].tolist()
    gf_sum = gp_sum = 0.0
    for f in range(2):
        fr = g["frames"][f]
        L = tuple((g["bounds"][f][:, 1] - g["bounds"][f][:, 0]).tolist())
        rho, rho_pairs, atom_types, _ = M._calc_props(L, fr[:, 1], fr[:, 1], 9, g["mass"], rel, False)
        gf, gp = M._normalize_rdf(0.05, rho_pairs, atom_types, rel, 5, 400, g["rdf_def_part"][f].astype(float),
                                  g["rdf_def_full"][f].astype(float), len(fr), rho)
        gf_sum, gp_sum = gf_sum + gf, gp_sum + gp
    df = M._save_rdf((np.arange(400) + 0.5) * 0.05, np.asarray(rel).T, None, False, gp_sum / 2, gf_sum / 2)
    np.testing.assert_array_equal(df.to_numpy(), g["rdf_def_df"])
    assert list(df.columns) == list(g["rdf_def_df_columns"])


def test_calc_props_errors():
    from mdproptools_amd.structural import rdf_cn as M

    types = np.array([1.0, 1.0, 2.0, 3.0])
    with pytest.raises(ValueError):  # wrong number of atom types (rdf_cn.py:266-271)
        M._calc_props((10.0, 10.0, 10.0), types, types, 2, [1.0, 1.0], [[1], [2]], False)
    with pytest.raises(KeyError):  # relation names a type that does not occur
        M._calc_props((10.0, 10.0, 10.0), types, types, 3, [1.0, 1.0, 1.0], [[1], [7]], False)


def test_public_signatures_match_reference_surface():
    """Names, positional order and defaults of the drop-in surface."""
    from mdproptools_amd.dynamical.conductivity import Conductivity
    from mdproptools_amd.dynamical.diffusion import Diffusion
    from mdproptools_amd.dynamical.viscosity import Viscosity
    from mdproptools_amd.structural import rdf_cn

    def sig(f):
        return [(p.name, p.default if p.default is not inspect._empty else None)
                for p in inspect.signature(f).parameters.values()]

    assert sig(rdf_cn.calc_atomic_rdf) == [
        ("r_cut", None), ("bin_size", None), ("num_types", None), ("mass", None), ("partial_relations", None),
        ("filename", None), ("num_mols", None), ("num_atoms_per_mol", None), ("path_or_buff", "rdf.csv"),
        ("save_mode", True)]
    assert sig(rdf_cn.calc_atomic_cn)[-2:] == [("path_or_buff", "cn.csv"), ("save_mode", True)]
    assert sig(rdf_cn.calc_molecular_rdf)[6:] == [("num_mols", None), ("num_atoms_per_mol", None),
                                                  ("path_or_buff", "rdf_mol.csv"), ("save_mode", True)]
    assert sig(rdf_cn.calc_molecular_cn)[-2:] == [("path_or_buff", "cn_mol.csv"), ("save_mode", True)]
    assert sig(Diffusion.__init__)[1:] == [("timestep", 1), ("units", "real"), ("outputs_dir", None),
                                           ("diff_dir", None)]
    assert sig(Diffusion.get_msd_from_dump)[1:] == [
        ("filename", None), ("msd_type", "com"), ("num_mols", None), ("num_atoms_per_mol", None), ("mass", None),
        ("com_drift", False), ("avg_interval", False), ("tao_coeff", 4)]
    assert sig(Diffusion.calc_diff)[1:] == [("msd", None), ("initial_time", None), ("final_time", None),
                                            ("dimension", 3), ("diff_names", None), ("save", False), ("plot", False)]
    assert sig(Conductivity.__init__)[1:] == [
        ("filename", None), ("num_mols", None), ("num_atoms_per_mol", None), ("volume", None), ("mass", None),
        ("temp", 298.15), ("timestep", 1), ("units", "real"), ("working_dir", None)]
    assert sig(Viscosity.__init__)[1:] == [
        ("log_pattern", None), ("cutoff_time", None), ("volume", None), ("temp", 298.15), ("timestep", 1),
        ("acf_method", "wkt"), ("units", "real"), ("working_dir", None)]
    for name in ("correlate", "detect_time_range", "get_charge_flux", "correlate_charge_flux",
                 "integrate_charge_flux_correlation", "fit_curve", "green_kubo", "calc_cond", "einstein", "nernst"):
        assert hasattr(Conductivity, name)
    for name in ("autocorrelate", "exp_func", "calc_visc", "_calc_3d_visc", "calc_avg_visc", "fit_avg_visc",
                 "bootstrapping"):
        assert hasattr(Viscosity, name)
    with pytest.raises(KeyError):
        Diffusion(units="furlongs")
    from mdproptools_amd.dynamical.residence_time import ResidenceTime

    assert sig(ResidenceTime.__init__)[1:] == [
        ("r_cut", None), ("partial_relations", None), ("filename", None), ("dt", 1), ("num_mols", None),
        ("num_atoms_per_mol", None), ("working_dir", None)]  # residence_time.py:40-49
    assert sig(ResidenceTime.fit_auto_correlation)[1:] == [("cut_percent", 0.9), ("plot", True)]
    assert hasattr(ResidenceTime, "calc_auto_correlation")
    assert sig(rdf_cn.calc_intermolecular_rdf)[6:] == [("num_mols", None), ("num_atoms_per_mol", None),
                                                       ("path_or_buff", "rdf_mol.csv"), ("save_mode", True)]


def test_detect_time_range_and_green_kubo():
    from mdproptools_amd.dynamical.conductivity import Conductivity

    g = load_golden("host_logic.npz")
    for f, expect in zip(g["dtr_flux"], g["dtr_range"]):
        assert tuple(Conductivity.detect_time_range(f, float(g["dtr_tol"]))) == tuple(expect)
    c = Conductivity.__new__(Conductivity)
    c.temp, c.volume = float(g["fc_temp"]), float(g["fc_volume"])
    np.testing.assert_allclose(c.green_kubo(g["fc_ave"]), g["fc_cond"], rtol=1e-15)
    with pytest.raises(TypeError):
        Conductivity.detect_time_range(np.random.default_rng(0).standard_normal(200), 1e-9)


def test_ols_through_origin_vs_reference_table(g_small):
    from mdproptools_amd.dynamical.diffusion import Diffusion

    g = g_small
    msd = pd.DataFrame(g["comd_msd"], columns=list(g["comd_msd_cols"]))
    import tempfile

    d = Diffusion(diff_dir=tempfile.mkdtemp())
    table = d.calc_diff(msd, diff_names=["dme", "tfsi", "mg"], save=True)
    np.testing.assert_allclose(table.to_numpy(), g["comd_diff"], rtol=1e-10)
    assert list(table.columns) == ["diffusion (m2/s)", "std", "R2"]
    dist = d.get_diff_dist(pd.DataFrame(g["comd_msd_int"], columns=list(g["comd_msd_int_cols"])), dump_freq=50000)
    np.testing.assert_allclose(dist.to_numpy(), g["comd_diff_dist"], rtol=1e-14)


def test_lammps_readers_round_trip(tmp_path):
    from mdproptools_amd import io as mio

    rng = np.random.default_rng(1)
    cols = ["id", "type", "x", "y", "z"]
    for step in (0, 10, 200, 30):
        tbl = np.column_stack([rng.permutation(50) + 1, rng.integers(1, 4, 50), rng.random((50, 3)).round(5) * 9])
        mio.write_dump(tmp_path / f"d.{step}.dump", step, [[0.5, 9.5]] * 3, cols, tbl)
    dumps = list(mio.parse_lammps_dumps(str(tmp_path / "d.*.dump")))
    assert [d.timestep for d in dumps] == [0, 10, 30, 200]  # numeric, not lexical, order
    assert dumps[0].natoms == 50 and list(dumps[0].data.columns) == cols
    assert dumps[0].box.to_lattice().lengths == (9.0, 9.0, 9.0)
    steps, bounds, planes = mio.read_dump_arrays(str(tmp_path / "d.*.dump"), ["x", "y", "z"])
    assert planes.shape == (4, 3, 50) and np.all(np.diff(steps) > 0)
    tab = np.column_stack([np.arange(5) * 10, rng.random((5, 2)).round(6)])  # short mantissas parse exactly
    mio.write_log(tmp_path / "log.a", tab, ["Step", "Pxy", "msd_1"])
    (df,) = mio.parse_lammps_log(str(tmp_path / "log.a"))
    np.testing.assert_array_equal(df.to_numpy(), tab)


def test_concat_log_segments(tmp_path):
    """Restart segments: numeric order, every segment but the last drops its final (repeated) row."""
    from mdproptools_amd import io as mio
    from mdproptools_amd.utilities.log import concat_log

    for seg, steps in [(1, [0, 10, 20]), (10, [40, 50]), (2, [20, 30, 40])]:
        tab = np.column_stack([steps, np.asarray(steps) * 0.5])
        mio.write_log(tmp_path / ("log.%d" % seg), tab, ["Step", "c_msd[4]"])
    full = concat_log("log.*", working_dir=str(tmp_path))
    assert list(full["Step"]) == [0, 10, 20, 30, 40, 50]
    assert list(full.index) == list(range(6))


def test_get_msd_from_log(tmp_path):
    """diffusion.py:241-265: msd columns x DISTANCE_CONVERSION**2 and a 'Time (s)' column (host only)."""
    from mdproptools_amd import io as mio
    from mdproptools_amd.dynamical.diffusion import Diffusion

    steps = np.arange(0, 50, 10)
    tab = np.column_stack([steps, steps * 0.25, steps * 1.5, 300.0 + 0 * steps])
    mio.write_log(tmp_path / "log.mixture_nvt", tab, ["Step", "c_msd1[4]", "c_msd2[4]", "Temp"])
    d = Diffusion(timestep=2, units="real", outputs_dir=str(tmp_path), diff_dir=str(tmp_path))
    msd = d.get_msd_from_log("log.mixture_nvt")
    assert list(msd.columns) == ["c_msd1[4]", "c_msd2[4]", "Time (s)"]
    np.testing.assert_allclose(msd["c_msd2[4]"], steps * 1.5 * 1e-20, rtol=1e-15)
    np.testing.assert_allclose(msd["Time (s)"], steps * 2 * 1e-15, rtol=1e-15)


def test_residence_time_fit_host_logic(tmp_path):
    """fit_auto_correlation (residence_time.py:148-200): stretched-exponential fit, table layout and CSV, on
    the table the real reference fitted (no GPU involved)."""
    import pandas as pd
    from conftest import load_golden
    from mdproptools_amd.dynamical.residence_time import ResidenceTime

    g = load_golden("residence.npz")
    rt = ResidenceTime([[0, 1]], [[9], [1]], "unused", working_dir=str(tmp_path))
    assert rt.dt == 1e-3 and rt.relation_matrix.tolist() == [[9, 1]]
    rt.corr_df = pd.DataFrame({"Time (ps)": g["fit_t"], "9-1": g["fit_y"]})
    res = rt.fit_auto_correlation(cut_percent=0.9, plot=False)
    np.testing.assert_allclose(res["9-1"], g["fit_res"], rtol=1e-6)
    assert list(rt.res_time_df.index) == ["a", "tau_res", "tau_short", "beta", "r (ps)"]
    assert (tmp_path / "residence_time.csv").exists()
    y = ResidenceTime._stretched_exp_function(np.array([0.0, 1.0]), 0.8, 40.0, 1.5, 0.7)
    assert y[0] == 1.0 and 0 < y[1] < 1


def test_batch_normalisation_sum_and_csv_are_the_per_frame_code_bit_for_bit(tmp_path):
    """Round 3 host-share cut: the per-frame normalisation of a batch in one expression, the frame-ordered sum and the
    hand-written CSV writer give exactly what the per-frame code / pandas gave (rdf_cn.py:297-329, 502-521, 341-365)."""
    import pandas as pd

    from mdproptools_amd.structural import rdf_cn as R

    rng = np.random.default_rng(3)
    B, nb, rel = 7, 160, [[1, 1, 2, 3], [1, 2, 3, 3]]
    n = 3000
    full = rng.integers(0, 50_000, (B, nb)).astype(np.uint64)
    part = rng.integers(0, 50_000, (B, 4, nb)).astype(np.uint64)
    ty = (1 + np.arange(n) % 3).astype(np.float64)
    props = [R._calc_props((30.0 + 0.1 * k, 31.0, 32.0 - 0.05 * k), ty, ty, 3, [1.0, 2.0, 3.0], rel, False) for k in range(B)]
    memo = [R._calc_props_memo((30.0 + 0.1 * (k // 2), 31.0, 32.0), ty, ty, 3, [1.0, 2.0, 3.0], rel, False) for k in range(B)]
    for k in range(B):  # the memo returns what a fresh call returns (boxes repeat in pairs here)
        fresh = R._calc_props((30.0 + 0.1 * (k // 2), 31.0, 32.0), ty, ty, 3, [1.0, 2.0, 3.0], rel, False)
        assert fresh[0] == memo[k][0] and np.array_equal(fresh[1], memo[k][1]) and fresh[2] == memo[k][2]
    rows = R._normalize_rdf_batch(0.05, props, rel, 4, nb, part, full, [n] * B)
    want = []
    for k in range(B):
        gf, gp = R._normalize_rdf(0.05, props[k][1], props[k][2], rel, 4, nb, part[k].astype(np.float64),
                                  full[k].astype(np.float64), n, props[k][0])
        want.append(np.concatenate([gf, np.ravel(gp)]))
    np.testing.assert_array_equal(rows, np.stack(want))
    only_part = R._normalize_rdf_batch(0.05, props, rel, 4, nb, part)
    np.testing.assert_array_equal(only_part, np.stack(want)[:, nb:])
    acc = np.zeros(rows.shape[1])
    for r in want:
        acc += r
    np.testing.assert_array_equal(R._sum_frames(rows), acc)
    vals = np.column_stack([(np.arange(nb) + 0.5) * 0.05, acc[:nb] / B, (acc[nb:].reshape(4, nb) / B).T])
    vals[3, 2], vals[5, 3], vals[6, 1], vals[7, 4] = np.nan, np.inf, 1e-300, 1e22
    df = pd.DataFrame(vals, columns=[R._R_LABEL, "g_full(r)"] + ["g_%d-%d" % (a, b) for a, b in zip(*rel)])
    R._write_csv(df, str(tmp_path / "fast.csv"))
    df.to_csv(str(tmp_path / "pandas.csv"), index=False)
    assert open(tmp_path / "fast.csv", "rb").read() == open(tmp_path / "pandas.csv", "rb").read()
    import io

    buf = io.StringIO()  # not a path: pandas writes it
    R._write_csv(df, buf)
    assert buf.getvalue() == open(tmp_path / "pandas.csv").read()


def test_diffusion_private_drift_helpers_match_the_array_path():
    """Diffusion._calculate_type_com / _modify_dump_coordinates (the reference's private helpers, diffusion.py:83-96, kept
    for callers that subclass): on a (Time, type, mol_id)-indexed frame they remove the same per-type drift as the array
    path the product runs (`_remove_drift`), to rounding."""
    import pandas as pd

    from mdproptools_amd.dynamical.diffusion import Diffusion

    rng = np.random.default_rng(12)
    F, counts = 6, [5, 3, 4]
    E = sum(counts)
    types = np.repeat([1, 2, 3], counts)
    mol_id = np.concatenate([np.arange(1, c + 1) for c in counts])
    mass = rng.uniform(1.0, 30.0, E)
    r = rng.normal(0, 1, (1, 3, E)) + np.cumsum(rng.normal(0, 0.3, (F, 3, E)), axis=0)
    times = np.arange(F) * 2.0e-15
    rows = []
    for f in range(F):
        rows.append(pd.DataFrame({"Time (s)": times[f], "type": types, "mol_id": mol_id, "mass": mass,
                                  "xu": r[f, 0], "yu": r[f, 1], "zu": r[f, 2]}))
    msd_df = pd.concat(rows).set_index(["Time (s)", "type", "mol_id"]).sort_index()
    d = Diffusion()
    com = d._calculate_type_com(msd_df, ["Time (s)", "type"])
    assert list(com.columns) == ["xu", "yu", "zu"] and com.index.names == ["Time (s)", "type"]
    goff = np.concatenate([[0], np.cumsum(counts)])
    want_com = Diffusion._type_com(r, mass, goff)  # [F,3,G]
    np.testing.assert_allclose(com.to_numpy().reshape(F, 3, 3).transpose(0, 2, 1), want_com, rtol=1e-13)
    out = d._modify_dump_coordinates(msd_df.copy())
    want = Diffusion._remove_drift(r, mass, goff)
    got = out[["xu", "yu", "zu"]].to_numpy().reshape(F, E, 3).transpose(0, 2, 1)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(got[0], r[0])  # time 0 is the reference: no drift there
