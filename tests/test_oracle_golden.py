"""
CPU-only: pin the oracle (oracle/cpu_ref.py numpy + oracle/cpu_ref.c) to the
golden vectors recorded from the real reference (oracle/make_golden.py).
Integer work must match bit for bit; floating point within the stated rtol.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REPO, load_golden, sorted_frame
from oracle import cpu_ref as O
from oracle import cref as C


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rdf_cn_synth_numpy_and_c(g_synth, tag):
    g = g_synth
    data, L, rel = g[tag + "_data"], g[tag + "_lengths"], g[tag + "_rel"]
    r_cut, ddr = float(g[tag + "_r_cut"]), float(g[tag + "_ddr"])
    nb = int(r_cut / ddr)
    full, part, ov = O.rdf_pairs(data, rel, L, r_cut, ddr, nb)
    assert ov == 0
    np.testing.assert_array_equal(full, g[tag + "_full"])
    np.testing.assert_array_equal(part, g[tag + "_part"])
    xyz = np.ascontiguousarray(data[:, 1:4].T)
    ty = data[:, 0].astype(np.int32)
    cfull, cpart, cov = C.rdf_pairs(xyz, ty, rel, L, r_cut ** 2, ddr, nb)
    assert cov == 0
    np.testing.assert_array_equal(cfull.astype(np.int64), g[tag + "_full"])
    np.testing.assert_array_equal(cpart.astype(np.int64), g[tag + "_part"])
    cuts = g[tag + "_cn_cut"]
    np.testing.assert_array_equal(O.cn_pairs(data, rel, L, list(cuts)), g[tag + "_cn"])
    np.testing.assert_array_equal(
        C.cn_pairs(xyz, ty, rel, L, [c ** 2 for c in cuts]).astype(np.int64), g[tag + "_cn"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rdf_cn_mol_synth(g_synth, tag):
    g = g_synth
    data, mol, L, rel = g[tag + "_data"], g[tag + "_mol"], g[tag + "_lengths"], g[tag + "_mol_rel"]
    r_cut, ddr = float(g[tag + "_r_cut"]), float(g[tag + "_ddr"])
    nb = int(r_cut / ddr)
    part, ov = O.rdf_mol_pairs(data, mol, rel, L, r_cut, ddr, nb)
    assert ov == 0
    np.testing.assert_array_equal(part, g[tag + "_mol_part"])
    cuts = g[tag + "_mol_cn_cut"]
    np.testing.assert_array_equal(O.cn_mol_pairs(data, mol, rel, L, list(cuts)), g[tag + "_mol_cn"])
    xyz = np.ascontiguousarray(data[:, 1:4].T)
    sxyz = np.ascontiguousarray(mol[:, 1:4].T)
    cpart, cov = C.rdf_rect(xyz, data[:, 0].astype(np.int32), sxyz, mol[:, 0].astype(np.int32),
                            rel, L, r_cut ** 2, ddr, nb)
    np.testing.assert_array_equal(cpart.astype(np.int64), g[tag + "_mol_part"])
    ccn = C.cn_rect(xyz, data[:, 0].astype(np.int32), sxyz, mol[:, 0].astype(np.int32),
                    rel, L, [c ** 2 for c in cuts])
    np.testing.assert_array_equal(ccn.astype(np.int64), g[tag + "_mol_cn"])


def test_calc_atom_type(g_synth):
    g = g_synth
    nm, na = g["atom_type_num_mols"], g["atom_type_num_atoms"]
    ids = np.arange(1, int(np.dot(nm, na)) + 1, dtype=np.float64)
    np.testing.assert_array_equal(O.calc_atom_type(ids, nm, na), g["atom_type_out"])


def test_c1_known_answers(g_c1):
    """SURVEY.md §8a known integers for mg_tfsi_dme frame 0 (reference run in the build container)."""
    g = g_c1
    assert int(g["rdf_def_full"][0].sum()) == 30926986
    np.testing.assert_array_equal(g["rdf_def_part"][0].sum(axis=1), [10670, 782, 3127, 358, 1975213])
    np.testing.assert_array_equal(g["cn_def_raw"][0], [141, 41, 57, 120])
    np.testing.assert_array_equal(g["rdf_def_full"][0][18:25], [0, 14, 948, 6376, 4172, 310, 4])


def test_c1_rdf_c_oracle_frame0(g_c1):
    g = g_c1
    fr = sorted_frame(g["frames"][0])
    L = g["bounds"][0][:, 1] - g["bounds"][0][:, 0]
    xyz = np.ascontiguousarray(fr[:, 2:5].T)
    ty = fr[:, 1].astype(np.int32)
    rel = g["rdf_def_rel"].T
    full, part, ov = C.rdf_pairs(xyz, ty, rel, L, 400.0, 0.05, 400)
    assert ov == 0
    np.testing.assert_array_equal(full.astype(np.int64), g["rdf_def_full"][0])
    np.testing.assert_array_equal(part.astype(np.int64), g["rdf_def_part"][0])
    cn = C.cn_pairs(xyz, ty, g["cn_def_rel"].T, L, g["cn_def_cut"] ** 2)
    np.testing.assert_array_equal(cn.astype(np.int64), g["cn_def_raw"][0])
    # altered ids (rdf_cn.py:197-215) then the same loop
    alt = O.calc_atom_type(fr[:, 0], g["num_mols"], g["num_atoms_per_mol"]).astype(np.int32)
    full, part, ov = C.rdf_pairs(xyz, alt, g["rdf_alt_rel"].T, L, 400.0, 0.05, 400)
    np.testing.assert_array_equal(full.astype(np.int64), g["rdf_alt_full"][0])
    np.testing.assert_array_equal(part.astype(np.int64), g["rdf_alt_part"][0])
    cn = C.cn_pairs(xyz, alt, g["rdf_alt_rel"].T, L, g["cn_alt_cut"] ** 2)
    np.testing.assert_array_equal(cn.astype(np.int64), g["cn_alt_raw"][0])


def test_c1_normalisation_and_frame_average(g_c1):
    """R8: per-frame normalisation then mean over frames reproduces the reference DataFrame exactly."""
    g = g_c1
    rel = g["rdf_def_rel"]
    gf_sum, gp_sum = 0.0, 0.0
    for f in range(2):
        fr = sorted_frame(g["frames"][f])
        L = g["bounds"][f][:, 1] - g["bounds"][f][:, 0]
        counts = {int(t): int(c) for t, c in zip(*np.unique(fr[:, 1].astype(np.int64), return_counts=True))}
        gf, gp = O.normalize_rdf(g["rdf_def_full"][f].astype(np.float64),
                                 g["rdf_def_part"][f].astype(np.float64),
                                 len(fr), np.prod(L), counts, counts, rel.tolist(), 0.05)
        gf_sum, gp_sum = gf_sum + gf, gp_sum + gp
    df = g["rdf_def_df"]
    np.testing.assert_array_equal(df[:, 0], (np.arange(400) + 0.5) * 0.05)
    np.testing.assert_array_equal(df[:, 1], gf_sum / 2)
    np.testing.assert_array_equal(df[:, 2:], (gp_sum / 2).T)


def test_c1_molecular(g_c1):
    g = g_c1
    fr = sorted_frame(g["frames"][0])
    L = g["bounds"][0][:, 1] - g["bounds"][0][:, 0]
    mt, mid, off, seg_type = O.molecule_layout(g["num_mols"], g["num_atoms_per_mol"])
    amass = g["mass"][fr[:, 1].astype(np.int64) - 1]
    com = O.mol_com_dot(fr[:, 2:5], amass, off)
    ref_com = g["mol_com"][0]
    np.testing.assert_array_equal(ref_com[:, 0], seg_type)
    np.testing.assert_allclose(com, ref_com[:, 1:4], rtol=1e-13, atol=0)
    # integer parity of the rectangular loops given the reference's own COM sites
    xyz = np.ascontiguousarray(fr[:, 2:5].T)
    part, ov = C.rdf_rect(xyz, fr[:, 1].astype(np.int32), np.ascontiguousarray(ref_com[:, 1:4].T),
                          ref_com[:, 0].astype(np.int32), g["mol_rel"].T, L, 400.0, 0.05, 400)
    np.testing.assert_array_equal(part.astype(np.int64), g["mol_rdf_part"][0])
    cn = C.cn_rect(xyz, fr[:, 1].astype(np.int32), np.ascontiguousarray(ref_com[:, 1:4].T),
                   ref_com[:, 0].astype(np.int32), g["mol_rel"].T, L, g["mol_cn_cut"] ** 2)
    np.testing.assert_array_equal(cn.astype(np.int64), g["mol_cn_raw"][0])


def _small_arrays(g):
    cols = list(g["columns"])
    frames = np.stack([sorted_frame(fr, cols.index("id")) for fr in g["frames"]])
    return cols, frames


def test_msd_allatom(g_small):
    g = g_small
    cols, fr = _small_arrays(g)
    r = fr[:, :, [cols.index(c) for c in ("xu", "yu", "zu")]] * 1e-10  # diffusion.py:201-203
    all4 = O.msd_single_origin(r)
    F, E = r.shape[:2]
    np.testing.assert_allclose(all4.reshape(F * E, 4), g["aa_msd_all"][:, 2:6], rtol=1e-12, atol=0)
    mean = O.msd_group_mean(all4, [0, E])[:, 0, :]
    np.testing.assert_allclose(mean, g["aa_msd"][:, 1:5], rtol=1e-10, atol=0)
    np.testing.assert_allclose(g["aa_msd"][:, 0], g["steps"] * 1e-15, rtol=1e-15)
    np.testing.assert_allclose(O.msd_fixed_lag(r, 4), g["aa_msd_int"][:, 1:5], rtol=1e-10, atol=0)


@pytest.mark.parametrize("tag,drift", [("com", False), ("comd", True)])
def test_msd_com(g_small, tag, drift):
    g = g_small
    cols, fr = _small_arrays(g)
    nm, na = g["num_mols"], g["num_atoms_per_mol"]
    _, _, off, seg_type = O.molecule_layout(nm, na)
    amass = g["mass"][fr[0][:, cols.index("type")].astype(np.int64) - 1]
    xyz = fr[:, :, [cols.index(c) for c in ("xu", "yu", "zu")]]
    coms, seg_mass = [], None
    for f in range(len(fr)):
        com, seg_mass, _ = O.calc_com(xyz[f], amass, off)
        coms.append(com * 1e-10)
    r = np.stack(coms)
    goff = np.concatenate([[0], np.cumsum(nm)])
    if drift:
        r = O.remove_type_drift(r, seg_mass * (1e-3 / 6.02214076e23), goff)
    all4 = O.msd_single_origin(r)
    F, E = r.shape[:2]
    np.testing.assert_allclose(all4.reshape(F * E, 4), g[tag + "_msd_all"][:, 3:7], rtol=1e-9, atol=0)
    mean = O.msd_group_mean(all4, goff)  # [F,G,4]
    # columns: Time, dx21 dy21 dz21 msd1, dx22 ... (diffusion.py:220-222)
    np.testing.assert_allclose(mean.reshape(F, -1), g[tag + "_msd"][:, 1:], rtol=1e-9, atol=0)
    np.testing.assert_allclose(O.msd_fixed_lag(r, 4), g[tag + "_msd_int"][:, 2:6], rtol=1e-9, atol=0)
    if drift:
        t = g[tag + "_msd"][:, 0]
        for k in range(3):
            slope, bse, r2 = O.ols_origin(t, mean[:, k, 3])
            np.testing.assert_allclose([slope / 6, bse / 6, r2], g["comd_diff"][k], rtol=1e-10)


def test_calc_com_and_charge_flux(g_small):
    g = g_small
    cols, fr = _small_arrays(g)
    nm, na = g["num_mols"], g["num_atoms_per_mol"]
    _, _, off, seg_type = O.molecule_layout(nm, na)
    f0 = fr[0]
    amass = g["mass"][f0[:, cols.index("type")].astype(np.int64) - 1]
    com, seg_mass, _ = O.calc_com(f0[:, [cols.index(c) for c in ("xu", "yu", "zu")]], amass, off)
    ref = g["calc_com_xu"]  # type, mol_id, xu, yu, zu, mass
    np.testing.assert_allclose(com, ref[:, 2:5], rtol=1e-13)
    np.testing.assert_allclose(seg_mass, ref[:, 5], rtol=1e-14)
    # velocities with masses taken from the dump column + charge
    dmass = f0[:, cols.index("mass")]
    q = f0[:, cols.index("q")]
    vcom, _, segq = O.calc_com(f0[:, [cols.index(c) for c in ("vx", "vy", "vz")]], dmass, off, q)
    refv = g["calc_com_v"]  # type, mol_id, mass, vx, vy, vz, q
    cv = list(g["calc_com_v_cols"])
    np.testing.assert_allclose(vcom, refv[:, [cv.index(c) for c in ("vx", "vy", "vz")]], rtol=1e-12)
    np.testing.assert_allclose(segq, refv[:, cv.index("q")], rtol=0, atol=1e-12)
    # charge flux for every frame (mass list given -> masses from the list)
    for f in range(len(fr)):
        vel = fr[f][:, [cols.index(c) for c in ("vx", "vy", "vz")]]
        j = O.charge_flux(vel, fr[f][:, cols.index("q")], amass, off, seg_type, 3,
                          1e-10 / 1e-15, 1.602176634e-19)
        np.testing.assert_allclose(j, g["cond_j"][:, :, f], rtol=1e-9, atol=1e-25)


def test_xcorr_and_integrals(g_acf):
    g = g_acf
    p = g["pressure"]
    for k in range(3):
        np.testing.assert_allclose(O.xcorr_fft(p[k], p[k]), g["acf_wkt"][k], rtol=1e-12,
                                   atol=1e-12 * g["acf_wkt"][k][0])
        d = O.xcorr_direct(p[k], p[k])
        np.testing.assert_allclose(d, g["acf_brute"][k], rtol=1e-12, atol=1e-12 * d[0])
        dc = C.xcorr_direct(p[k], p[k])
        np.testing.assert_allclose(dc, g["acf_brute"][k], rtol=0, atol=1e-10 * dc[0])
    j = g["flux"]
    np.testing.assert_allclose(O.xcorr_fft(j[0, 0], j[0, 1]), g["corr_01"], rtol=1e-12,
                               atol=1e-12 * abs(g["corr_01"]).max())
    tot = np.zeros_like(g["tot_flux"])
    for a in range(3):
        for b in range(3):
            for k in range(3):
                c = O.xcorr_fft(j[k, a], j[k, b])
                tot[a] += c
                tot[-1] += c
    np.testing.assert_allclose(tot, g["tot_flux"], rtol=1e-12, atol=1e-12 * abs(g["tot_flux"]).max())
    dt = g["flux_time"][1] - g["flux_time"][0]
    for a in range(4):
        np.testing.assert_allclose(O.cumtrapz(g["tot_flux"][a], dt, leading_zero=True),
                                   g["integral"][a], rtol=1e-13, atol=0)
    # _calc_3d_visc: acf * P^2, cumtrapz * V/(kB T), mean of the three (viscosity.py:172-190)
    vol = float(g["visc_volume"]) * 1e-30
    dts = (g["visc_step"][1] - g["visc_step"][0]) * int(g["visc_timestep"]) * 1e-15
    visc = []
    for k in range(3):
        acf = O.xcorr_fft(p[k], p[k]) * 101325 ** 2
        visc.append(vol / (1.380649e-23 * float(g["visc_temp"])) * O.cumtrapz(acf, dts))
    np.testing.assert_allclose(np.mean(visc, axis=0), g["visc_avg"], rtol=1e-9,
                               atol=1e-12 * abs(g["visc_avg"]).max())


def test_cell17_published_table():
    """The reference's only published known answer (notebook cell 17), as reproduced here."""
    with open(os.path.join(GOLDEN, "cell17.json")) as fh:
        t = json.load(fh)
    for col in ("diffusion (m2/s)", "std", "R2"):
        for pub, here in zip(t["published"][col], t["reference_here"][col]):
            assert float("%.6e" % here) == pytest.approx(pub, rel=1e-6) or float("%.6f" % here) == pub


@pytest.mark.skipif(not os.path.isdir("/root/reference/data/mg_tfsi_dme"),
                    reason="full mg_tfsi_dme trajectory only exists in the build container")
def test_cell17_oracle_chain_full_trajectory():
    """Oracle restatement of calc_com -> drift -> MSD -> OLS on all 101 frames vs cell 17."""
    from mdproptools_amd.io import read_dump_arrays
    steps, bounds, planes = read_dump_arrays(
        "/root/reference/data/mg_tfsi_dme/dump.nvt.*.dump", ["type", "xu", "yu", "zu"])
    mass = np.array([16.000, 12.010, 1.008, 14.010, 32.060, 16.000, 12.010, 19.000, 24.305])
    nm, na = [591, 66, 33], [16, 15, 1]
    _, _, off, _ = O.molecule_layout(nm, na)
    amass = mass[planes[0, 0].astype(np.int64) - 1]
    r, seg_mass = [], None
    for f in range(len(steps)):
        com, seg_mass, _ = O.calc_com(planes[f, 1:4].T, amass, off)
        r.append(com * 1e-10)
    goff = np.concatenate([[0], np.cumsum(nm)])
    r = O.remove_type_drift(np.stack(r), seg_mass * (1e-3 / 6.02214076e23), goff)
    mean = O.msd_group_mean(O.msd_single_origin(r), goff)
    t = steps * 1e-15
    with open(os.path.join(GOLDEN, "cell17.json")) as fh:
        tab = json.load(fh)
    for k in range(3):
        slope, bse, r2 = O.ols_origin(t, mean[:, k, 3])
        assert "%.6e" % (slope / 6) == "%.6e" % tab["published"]["diffusion (m2/s)"][k]
        assert "%.6e" % (bse / 6) == "%.6e" % tab["published"]["std"][k]
        assert "%.6f" % r2 == "%.6f" % tab["published"]["R2"][k]


# ------------------------------------------------------------------ residence autocorrelation
def _residence_inputs(g):
    frames = [fr[np.argsort(fr[:, 0], kind="stable")] for fr in g["frames"]]
    labels = O.calc_atom_type(frames[0][:, 0], g["num_mols"], g["num_atoms_per_mol"])
    xyz = np.stack([np.ascontiguousarray(fr[:, 2:5].T) for fr in frames])
    box = g["bounds"][:, :, 1] - g["bounds"][:, :, 0]
    return xyz, labels, box


def test_residence_oracle_matches_reference():
    """oracle shell indicator + exact autocovariance numerators against ResidenceTime.calc_auto_correlation of
    the real reference (30 frames, altered ids; one relation without any neighbour gives NaN in both)."""
    g = load_golden("residence.npz")
    xyz, labels, box = _residence_inputs(g)
    assert str(g["default_ids_error"]) == "ValueError"  # the reference's default-id mode does not run
    np.testing.assert_allclose(g["corr"][:, 0], g["steps"] * 2e-3)
    for kl, (k, l) in enumerate(g["rel"].T):
        lo2, hi2 = g["r_cut"][kl][0] ** 2, g["r_cut"][kl][1] ** 2
        h = np.array([O.shell_indicator(xyz[f][:, labels == k].T, xyz[f][:, labels == l].T, box[f], lo2, hi2, k == l)
                      for f in range(len(xyz))])
        with np.errstate(invalid="ignore"):
            corr = O.residence_autocorr(h)
        ref = g["corr"][:, 1 + kl]
        assert np.array_equal(np.isnan(corr), np.isnan(ref))
        np.testing.assert_allclose(corr[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=1e-12, atol=1e-15)


def test_acovf_shim_equals_its_definition():
    """oracle/shims/statsmodels acovf (the stand-in the reference's ResidenceTime ran on when the goldens were made,
    residence_time.py:128-130) against the direct-sum definition it claims, sum_t x[t] x[t+k] / (n - k): 0/1 indicator
    series as the reference feeds it, and real-valued ones."""
    import importlib.util

    path = os.path.join(REPO, "oracle", "shims", "statsmodels", "tsa", "stattools.py")
    spec = importlib.util.spec_from_file_location("_acovf_shim", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 30, 101, 256):
        for x in (rng.integers(0, 2, n).astype(float), rng.normal(size=n)):
            direct = np.array([np.dot(x[k:], x[: n - k]) / (n - k) for k in range(n)])
            got = mod.acovf(x, demean=False, unbiased=True, fft=True)
            np.testing.assert_allclose(got, direct, rtol=0, atol=1e-12 * max(1.0, abs(direct).max()))
            np.testing.assert_allclose(mod.acovf(x, demean=False, unbiased=True, fft=False), direct, rtol=1e-13,
                                       atol=1e-15)
        # indicator series: numerators are integers, so the FFT route must round to them exactly
        h = rng.integers(0, 2, n).astype(float)
        num = mod.acovf(h, demean=False, unbiased=True, fft=True) * (n - np.arange(n))
        assert np.array_equal(np.rint(num), [np.dot(h[k:], h[: n - k]) for k in range(n)])
