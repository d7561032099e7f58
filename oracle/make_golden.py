#!/usr/bin/env python
"""
oracle/make_golden.py — generate tests/golden/*.npz by running the REAL reference.

TEST INFRASTRUCTURE, build container only: /root/reference does not exist on
the GPU box, so what travels is the output of this script (inputs + expected
outputs as data), never the reference itself.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference is imported read-only from /root/reference with the stand-ins in
oracle/shims/ for its four missing dependencies (see oracle/shims/README.md).
Each block below names the reference entry point it records.
"""

import contextlib
import io
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shims"), REPO, REF]

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import scipy.integrate as _si  # noqa: E402

if not hasattr(_si, "cumtrapz"):  # removed upstream; viscosity.py:151 still calls it
    _si.cumtrapz = _si.cumulative_trapezoid

from mdproptools.structural import rdf_cn as ref_rdf  # noqa: E402
from mdproptools.dynamical.diffusion import Diffusion  # noqa: E402
from mdproptools.dynamical.conductivity import Conductivity  # noqa: E402
from mdproptools.dynamical.viscosity import Viscosity  # noqa: E402
from mdproptools.common.com_mols import calc_com  # noqa: E402

from mdproptools_amd import io as mio  # noqa: E402

OUT = os.environ.get("MDHIP_GOLDEN_OUT") or os.path.join(REPO, "tests", "golden")  # (override: regenerate elsewhere and compare)
DATA = os.path.join(REF, "data", "mg_tfsi_dme")
MASS = [16.000, 12.010, 1.008, 14.010, 32.060, 16.000, 12.010, 19.000, 24.305]
NUM_MOLS = [591, 66, 33]
NUM_ATOMS = [16, 15, 1]
COLS = ["id", "type", "mass", "q", "x", "y", "z", "xu", "yu", "zu", "vx", "vy", "vz"]


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


class Recorder:
    """Wrap a reference kernel so that every call's integer output is kept."""

    def __init__(self, module, name):
        self.module, self.name, self.calls = module, name, []
        self.orig = getattr(module, name)

    def __enter__(self):
        def wrapped(*args):
            res = self.orig(*args)
            self.calls.append(
                tuple(np.array(r, copy=True) for r in (res if isinstance(res, tuple) else (res,)))
            )
            return res

        setattr(self.module, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.module, self.name, self.orig)


def write_frames(tmp, steps, bounds, tables, columns):
    for s, b, t in zip(steps, bounds, tables):
        mio.write_dump(os.path.join(tmp, "dump.nvt.%d.dump" % s), s, b, columns, t)


def load_real_frames(steps):
    """The reference's own dump files through the INDEPENDENT pandas route of oracle/shims (what the reference reads
    when this script runs it); the product's readers — pandas route and native mmap route — must return the very same
    doubles for the same files, or this script stops: golden inputs are never 'parsed by the code under test'."""
    from pymatgen.io.lammps.outputs import parse_lammps_dumps as shim_parse  # oracle/shims, no product import

    frames, bounds = [], []
    for s in steps:
        path = os.path.join(DATA, "dump.nvt.%d.dump" % s)
        (d,) = list(shim_parse(path))
        arr = d.data[COLS].to_numpy(dtype=np.float64)
        (own,) = list(mio.parse_lammps_dumps(path))
        assert own.timestep == d.timestep and own.natoms == d.natoms
        assert np.array_equal(np.asarray(own.box.bounds), np.asarray(d.box.bounds))
        assert list(own.data.columns) == list(d.data.columns)
        assert np.array_equal(own.data[COLS].to_numpy(dtype=np.float64), arr), "product pandas route differs"
        by_id = d.data.sort_values("id")[COLS].to_numpy(dtype=np.float64).T
        ((ts, nb, _len, _names, planes),) = list(mio.iter_native_frames(path, COLS, sort_by="id"))
        assert ts == d.timestep and np.array_equal(np.asarray(nb), np.asarray(d.box.bounds))
        assert np.array_equal(np.asarray(planes), by_id), "product native route differs"
        _st, _bd, pl = mio.read_dump_arrays(path, COLS)
        assert np.array_equal(pl[0], by_id)
        frames.append(arr)
        bounds.append(d.box.bounds)
    return np.stack(frames), np.asarray(bounds)


# ----------------------------------------------------------------------------
def golden_c1_rdf():
    """calc_atomic_rdf/cn, calc_molecular_rdf/cn on two full mg_tfsi_dme frames."""
    steps = [0, 50000]
    frames, bounds = load_real_frames(steps)
    keep = [COLS.index(c) for c in ("id", "type", "x", "y", "z")]
    out = {
        "steps": np.asarray(steps),
        "bounds": bounds,
        "columns": np.array(["id", "type", "x", "y", "z"]),
        "frames": frames[:, :, keep],
        "mass": np.asarray(MASS),
        "num_mols": np.asarray(NUM_MOLS),
        "num_atoms_per_mol": np.asarray(NUM_ATOMS),
    }
    with tempfile.TemporaryDirectory() as tmp:
        write_frames(tmp, steps, bounds, frames[:, :, keep], ["id", "type", "x", "y", "z"])
        pat = os.path.join(tmp, "dump.nvt.*.dump")

        rel_def = [[9, 9, 9, 9, 1], [1, 4, 6, 9, 3]]
        with quiet(), Recorder(ref_rdf, "_rdf_loop") as rec:
            df = ref_rdf.calc_atomic_rdf(20, 0.05, 9, MASS, rel_def, pat, save_mode=False)
        out["rdf_def_rel"] = np.asarray(rel_def)
        out["rdf_def_full"] = np.stack([c[0] for c in rec.calls]).astype(np.int64)
        out["rdf_def_part"] = np.stack([c[1] for c in rec.calls]).astype(np.int64)
        out["rdf_def_df"] = df.to_numpy()
        out["rdf_def_df_columns"] = np.array(list(df.columns))

        rel_alt = [[32, 32], [17, 32]]
        with quiet(), Recorder(ref_rdf, "_rdf_loop") as rec:
            df = ref_rdf.calc_atomic_rdf(
                20, 0.05, 9, MASS, rel_alt, pat,
                num_mols=NUM_MOLS, num_atoms_per_mol=NUM_ATOMS, save_mode=False,
            )
        out["rdf_alt_rel"] = np.asarray(rel_alt)
        out["rdf_alt_full"] = np.stack([c[0] for c in rec.calls]).astype(np.int64)
        out["rdf_alt_part"] = np.stack([c[1] for c in rec.calls]).astype(np.int64)
        out["rdf_alt_df"] = df.to_numpy()

        rel_cn = [[9, 9, 9, 9], [1, 4, 6, 9]]
        cn_cut = [2.325, 4.375, 2.375, 13.0]
        with quiet(), Recorder(ref_rdf, "_cn_loop") as rec:
            df = ref_rdf.calc_atomic_cn(cn_cut, 0.05, 9, MASS, rel_cn, pat, save_mode=False)
        out["cn_def_rel"] = np.asarray(rel_cn)
        out["cn_def_cut"] = np.asarray(cn_cut)
        out["cn_def_raw"] = np.stack([c[0] for c in rec.calls]).astype(np.int64)
        out["cn_def_df"] = df.to_numpy()

        cn_cut_alt = [4.375, 13.0]
        with quiet(), Recorder(ref_rdf, "_cn_loop") as rec:
            df = ref_rdf.calc_atomic_cn(
                cn_cut_alt, 0.05, 9, MASS, rel_alt, pat,
                num_mols=NUM_MOLS, num_atoms_per_mol=NUM_ATOMS, save_mode=False,
            )
        out["cn_alt_cut"] = np.asarray(cn_cut_alt)
        out["cn_alt_raw"] = np.stack([c[0] for c in rec.calls]).astype(np.int64)
        out["cn_alt_df"] = df.to_numpy()

        rel_mol = [[9, 9, 4], [1, 2, 3]]
        with quiet(), Recorder(ref_rdf, "_rdf_mol_loop") as rec, Recorder(
            ref_rdf, "_define_mol_cols"
        ) as rec_com:
            df = ref_rdf.calc_molecular_rdf(
                20, 0.05, 9, MASS, rel_mol, pat, NUM_MOLS, NUM_ATOMS, save_mode=False
            )
        out["mol_rel"] = np.asarray(rel_mol)
        out["mol_rdf_part"] = np.stack([c[0] for c in rec.calls]).astype(np.int64)
        out["mol_rdf_df"] = df.to_numpy()
        out["mol_com"] = np.stack([c[0] for c in rec_com.calls]).astype(np.float64)

        mol_cut = [2.325, 3.775, 4.375]
        with quiet(), Recorder(ref_rdf, "_cn_mol_loop") as rec:
            df = ref_rdf.calc_molecular_cn(
                mol_cut, 0.05, 9, MASS, rel_mol, pat, NUM_MOLS, NUM_ATOMS, save_mode=False
            )
        out["mol_cn_cut"] = np.asarray(mol_cut)
        out["mol_cn_raw"] = np.stack([c[0] for c in rec.calls]).astype(np.int64)
        out["mol_cn_df"] = df.to_numpy()
    np.savez_compressed(os.path.join(OUT, "c1_rdf.npz"), **out)
    print("c1_rdf: full sum frame0 =", out["rdf_def_full"][0].sum(),
          "part sums", out["rdf_def_part"][0].sum(axis=1), "cn raw", out["cn_def_raw"][0])


# ----------------------------------------------------------------------------
def golden_inter_rdf():
    """calc_intermolecular_rdf (rdf_cn.py:857-903) on the two frames of c1_rdf.npz: molecule-COM to molecule-COM
    g(r); inputs are c1_rdf.npz's frames, only the outputs are stored here."""
    steps = [0, 50000]
    frames, bounds = load_real_frames(steps)
    keep = [COLS.index(c) for c in ("id", "type", "x", "y", "z")]
    rel = [[1, 2, 3, 3], [1, 3, 3, 1]]
    with tempfile.TemporaryDirectory() as tmp:
        write_frames(tmp, steps, bounds, frames[:, :, keep], ["id", "type", "x", "y", "z"])
        with quiet(), Recorder(ref_rdf, "_rdf_mol_loop") as rec:
            df = ref_rdf.calc_intermolecular_rdf(20, 0.05, 3, MASS, rel,
                                                 os.path.join(tmp, "dump.nvt.*.dump"), NUM_MOLS, NUM_ATOMS,
                                                 save_mode=False)
    np.savez_compressed(os.path.join(OUT, "inter_rdf.npz"), rel=np.asarray(rel), df=df.to_numpy(),
                        columns=np.array(list(df.columns)), part=np.stack([c[0] for c in rec.calls]).astype(np.int64))
    print("inter_rdf:", list(df.columns), df.to_numpy()[100:103])


# ----------------------------------------------------------------------------
def reduced_system(frames, n_keep=(60, 12, 6)):
    """First n_keep molecules of each type, re-numbered 1..n in type-major order."""
    sel, off = [], 0
    for nm, na, nk in zip(NUM_MOLS, NUM_ATOMS, n_keep):
        sel.extend(range(off, off + nk * na))
        off += nm * na
    sel = np.asarray(sel)
    out = []
    for fr in frames:
        fr = fr[np.argsort(fr[:, 0], kind="stable")][sel].copy()
        fr[:, 0] = np.arange(1, len(sel) + 1)
        out.append(fr)
    return np.stack(out), list(n_keep)


def golden_small_md():
    """Diffusion.get_msd_from_dump / calc_diff / get_diff_dist, calc_com, Conductivity chain
    on a 1146-atom sub-system of 9 mg_tfsi_dme frames."""
    steps = [50000 * k for k in range(9)]
    frames, bounds = load_real_frames(steps)
    frames, num_mols = reduced_system(frames)
    rng = np.random.default_rng(20250328)
    shuffled = np.stack([fr[rng.permutation(len(fr))] for fr in frames])  # dumps are unsorted
    out = {
        "steps": np.asarray(steps), "bounds": bounds, "columns": np.array(COLS),
        "frames": shuffled, "mass": np.asarray(MASS),
        "num_mols": np.asarray(num_mols), "num_atoms_per_mol": np.asarray(NUM_ATOMS),
    }
    with tempfile.TemporaryDirectory() as tmp:
        write_frames(tmp, steps, bounds, shuffled, COLS)
        d = Diffusion(timestep=1, units="real", outputs_dir=tmp, diff_dir=tmp)
        with quiet():
            msd, msd_all, msd_int = d.get_msd_from_dump(
                "dump.nvt.*.dump", msd_type="allatom", avg_interval=True, tao_coeff=4)
        out["aa_msd"], out["aa_msd_cols"] = msd.to_numpy(), np.array(list(msd.columns))
        out["aa_msd_all"], out["aa_msd_all_cols"] = msd_all.to_numpy(), np.array(list(msd_all.columns))
        out["aa_msd_int"], out["aa_msd_int_cols"] = msd_int.to_numpy(), np.array(list(msd_int.columns))
        for tag, drift in (("com", False), ("comd", True)):
            with quiet():
                msd, msd_all, msd_int = d.get_msd_from_dump(
                    "dump.nvt.*.dump", msd_type="com", num_mols=num_mols,
                    num_atoms_per_mol=NUM_ATOMS, mass=MASS, com_drift=drift,
                    avg_interval=True, tao_coeff=4)
            out[tag + "_msd"], out[tag + "_msd_cols"] = msd.to_numpy(), np.array(list(msd.columns))
            out[tag + "_msd_all"] = msd_all.to_numpy()
            out[tag + "_msd_all_cols"] = np.array(list(msd_all.columns))
            out[tag + "_msd_int"] = msd_int.to_numpy()
            out[tag + "_msd_int_cols"] = np.array(list(msd_int.columns))
        with quiet():
            diff = d.calc_diff(msd, diff_names=["dme", "tfsi", "mg"])
            dist = d.get_diff_dist(msd_int.copy(), dump_freq=50000, tao_coeff=4)
        out["comd_diff"] = diff.to_numpy()
        out["comd_diff_dist"] = dist.to_numpy()
        # calc_com on frame 0: unwrapped coordinates, and velocities + charge
        (dump0,) = list(mio.parse_lammps_dumps(os.path.join(tmp, "dump.nvt.0.dump")))
        dump0.data = dump0.data.sort_values(by=["id"]).reset_index()
        com = calc_com(dump0, num_mols, NUM_ATOMS, MASS, atom_attributes=["xu", "yu", "zu"])
        out["calc_com_xu"] = com.reset_index().to_numpy(dtype=np.float64)
        out["calc_com_xu_cols"] = np.array(list(com.reset_index().columns))
        comv = calc_com(dump0, num_mols, NUM_ATOMS, None, atom_attributes=["vx", "vy", "vz"],
                        calc_charge=True)
        out["calc_com_v"] = comv.reset_index().to_numpy(dtype=np.float64)
        out["calc_com_v_cols"] = np.array(list(comv.reset_index().columns))
        # Conductivity chain
        vol = float(np.prod(bounds[0][:, 1] - bounds[0][:, 0]))
        c = Conductivity("dump.nvt.*.dump", num_mols, NUM_ATOMS, vol, mass=MASS,
                         temp=298.15, timestep=1, units="real", working_dir=tmp)
        j = c.get_charge_flux()
        tot = c.correlate_charge_flux(j)
        integ = c.integrate_charge_flux_correlation(tot)
        out["cond_volume"] = np.asarray(vol)
        out["cond_time"] = np.asarray(c.time)
        out["cond_j"], out["cond_tot_flux"], out["cond_integral"] = j, tot, integ
        out["cond_gk"] = c.green_kubo(integ[:, -1])
    np.savez_compressed(os.path.join(OUT, "small_md.npz"), **out)
    print("small_md: aa msd last", out["aa_msd"][-1], "diff", out["comd_diff"][:, 0])


# ----------------------------------------------------------------------------
def golden_residence():
    """ResidenceTime.calc_auto_correlation / fit_auto_correlation (dynamical/residence_time.py:70-200) on the
    1146-atom sub-system, 30 frames: Mg-O(DME), Mg-O(TFSI), a shell with a lower bound, and a same-type
    relation; default ids and altered ids."""
    from mdproptools.dynamical.residence_time import ResidenceTime

    steps = [50000 * k for k in range(30)]
    frames, bounds = load_real_frames(steps)
    frames, num_mols = reduced_system(frames)
    cols = ["id", "type", "x", "y", "z"]
    keep = [COLS.index(c) for c in cols]
    rng = np.random.default_rng(20250328 + 7)
    shuffled = np.stack([fr[rng.permutation(len(fr))][:, keep] for fr in frames])
    out = {"steps": np.asarray(steps), "bounds": bounds, "columns": np.array(cols), "frames": shuffled,
           "num_mols": np.asarray(num_mols), "num_atoms_per_mol": np.asarray(NUM_ATOMS)}
    # Only the altered-ids mode of the reference runs: with default ids it hands `_calc_rsq(..., num_of_ids=0)`
    # rows of [id, x, y, z] and stops with a broadcast ValueError (residence_time.py:96-101, rdf_cn.py:43).
    # Pseudo-types (rdf_cn.py:197-215): 1..16 DME atoms, 17..31 TFSI atoms, 32 Mg.
    rel = [[32, 32, 32, 1, 32], [1, 19, 32, 1, 4]]
    r_cut = [[0, 4.5], [0, 4.5], [0, 14.0], [2.0, 9.0], [2.0, 3.2]]
    with tempfile.TemporaryDirectory() as tmp:
        write_frames(tmp, steps, bounds, shuffled, cols)
        with quiet():
            rt = ResidenceTime(r_cut, rel, os.path.join(tmp, "dump.nvt.*.dump"), dt=2,
                               num_mols=list(num_mols), num_atoms_per_mol=NUM_ATOMS, working_dir=tmp)
            rt.calc_auto_correlation()
        out["rel"], out["r_cut"] = np.asarray(rel), np.asarray(r_cut, dtype=np.float64)
        out["corr"], out["corr_cols"] = rt.corr_df.to_numpy(), np.array(list(rt.corr_df.columns))
        try:
            with quiet():
                ResidenceTime(r_cut[:1], [[9], [1]], os.path.join(tmp, "dump.nvt.*.dump"),
                              working_dir=tmp).calc_auto_correlation()
            out["default_ids_error"] = np.array("")
        except Exception as e:  # recorded so the deliberate difference is documented by data
            out["default_ids_error"] = np.array(type(e).__name__)
        # the fit on a synthetic stretched-exponential table (the 30-frame correlations are too short to fit)
        t = np.arange(200) * 0.5
        y = ResidenceTime._stretched_exp_function(t, 0.8, 40.0, 1.5, 0.7)
        rt.corr_df = pd.DataFrame({"Time (ps)": t, "9-1": y})
        with quiet():
            fit = rt.fit_auto_correlation(cut_percent=0.9, plot=False)
        out["fit_t"], out["fit_y"], out["fit_res"] = t, y, np.asarray(fit["9-1"])
    np.savez_compressed(os.path.join(OUT, "residence.npz"), **out)
    print("residence:", out["corr_cols"], out["corr"][:3], out["default_ids_error"], out["fit_res"])


# ----------------------------------------------------------------------------
def golden_synth_rdf():
    """_rdf_loop/_cn_loop/_rdf_mol_loop/_cn_mol_loop on seeded synthetic frames with edge cases:
    non-cubic box with lo != 0, atoms outside the box by more than one box length,
    repeated / reversed / absent-type relations, coincident atoms."""
    rng = np.random.default_rng(20250328 + 1)
    out = {}
    cases = {
        "a": dict(n=512, lengths=(30.0, 31.5, 29.25), r_cut=10.0, ddr=0.1),
        "b": dict(n=700, lengths=(26.0, 26.0, 26.0), r_cut=13.0, ddr=0.05),
        "c": dict(n=300, lengths=(24.0, 25.0, 26.0), r_cut=12.0, ddr=0.02),
    }
    for tag, cfg in cases.items():
        n, L = cfg["n"], np.asarray(cfg["lengths"])
        xyz = rng.uniform(0.0, 1.0, size=(n, 3)) * L + 3.25
        far = rng.choice(n, size=n // 16, replace=False)
        xyz[far] += rng.integers(-2, 3, size=(len(far), 3)) * L  # unwrapped strays
        xyz[5] = xyz[4]  # coincident pair -> rsq == 0 -> bin 0
        xyz[7] = xyz[6] + np.array([L[0] / 2, 0.0, 0.0])  # |d| == L/2 exactly is NOT wrapped
        types = 1 + (np.arange(n) % 4)
        data = np.column_stack([types.astype(np.float64), xyz])
        rel = np.array([[1, 2], [2, 1], [1, 1], [3, 4], [4, 4], [2, 3], [1, 2], [5, 1]])
        nb = int(cfg["r_cut"] / cfg["ddr"])
        full = np.zeros(nb)
        part = np.zeros((len(rel), nb))
        ref_rdf._rdf_loop(data, rel, len(rel), tuple(L), cfg["r_cut"], cfg["ddr"], full, part)
        cuts = [2.325 + 0.9 * k for k in range(len(rel))]
        cn = np.zeros(len(rel))
        ref_rdf._cn_loop(data, rel, len(rel), tuple(L), cuts, cfg["ddr"], cn)
        m = n // 8
        mol = np.column_stack([1.0 + (np.arange(m) % 3), rng.uniform(0, 1, (m, 3)) * L + 3.25])
        rel_m = np.array([[1, 1], [1, 2], [4, 3], [2, 2], [1, 1]])
        mpart = np.zeros((len(rel_m), nb))
        ref_rdf._rdf_mol_loop(data, mol, rel_m, len(rel_m), tuple(L), cfg["r_cut"], cfg["ddr"], mpart)
        mcuts = [3.0, 5.5, 7.25, 9.0, 4.0]
        mcn = np.zeros(len(rel_m))
        ref_rdf._cn_mol_loop(data, mol, rel_m, len(rel_m), tuple(L), mcuts, cfg["ddr"], mcn)
        out.update({
            tag + "_data": data, tag + "_lengths": L, tag + "_r_cut": np.asarray(cfg["r_cut"]),
            tag + "_ddr": np.asarray(cfg["ddr"]), tag + "_rel": rel,
            tag + "_full": full.astype(np.int64), tag + "_part": part.astype(np.int64),
            tag + "_cn_cut": np.asarray(cuts), tag + "_cn": cn.astype(np.int64),
            tag + "_mol": mol, tag + "_mol_rel": rel_m, tag + "_mol_part": mpart.astype(np.int64),
            tag + "_mol_cn_cut": np.asarray(mcuts), tag + "_mol_cn": mcn.astype(np.int64),
        })
        print("synth", tag, "full sum", int(full.sum()), "cn", cn.astype(int))
    # _calc_atom_type on ids 1..N (float64 table, as the reference passes it)
    num_mols, num_atoms = [7, 3, 5], [4, 6, 1]
    ntot = int(np.dot(num_mols, num_atoms))
    tbl = np.column_stack([np.arange(1, ntot + 1, dtype=np.float64), np.zeros((ntot, 4))])
    out["atom_type_num_mols"], out["atom_type_num_atoms"] = np.asarray(num_mols), np.asarray(num_atoms)
    out["atom_type_out"] = ref_rdf._calc_atom_type(tbl.copy(), num_mols, num_atoms)[:, 0]
    np.savez_compressed(os.path.join(OUT, "synth_rdf.npz"), **out)


# ----------------------------------------------------------------------------
def golden_acf():
    """Viscosity.autocorrelate (wkt, brute_force), _calc_3d_visc, Conductivity.correlate,
    correlate_charge_flux, integrate_charge_flux_correlation on seeded AR(1) series."""
    rng = np.random.default_rng(20250328 + 5)
    n = 4096

    def ar1(size):
        e = rng.standard_normal(size)
        x = np.empty(size)
        x[0] = e[0]
        for t in range(1, size):
            x[t] = 0.99 * x[t - 1] + e[t]
        return x

    p = np.stack([ar1(n) for _ in range(3)]) * 100.0
    out = {"pressure": p}
    out["acf_wkt"] = np.stack([Viscosity.autocorrelate(s, "wkt") for s in p])
    out["acf_brute"] = np.stack([Viscosity.autocorrelate(s, "brute_force") for s in p])
    v = Viscosity("log.*", 0, 118969.0, temp=298.15, timestep=2, acf_method="wkt", units="real")
    log_df = pd.DataFrame({"Step": np.arange(n) * 5, "Pxy": p[0], "Pxz": p[1], "Pyz": p[2]})
    avg, visc, acf = v._calc_3d_visc(log_df)
    out["visc_avg"], out["visc_data"], out["visc_acf"] = avg, visc, acf
    v.acf_method = "brute_force"
    avg_b, _, _ = v._calc_3d_visc(log_df)
    out["visc_avg_brute"] = avg_b
    out["visc_volume"], out["visc_temp"], out["visc_timestep"] = (
        np.asarray(118969.0), np.asarray(298.15), np.asarray(2))
    out["visc_step"] = log_df["Step"].to_numpy()
    nj = 2048
    j = np.stack([[ar1(nj) for _ in range(3)] for _ in range(3)]) * 1e-16
    out["flux"] = j
    out["corr_01"] = Conductivity.correlate(j[0, 0], j[0, 1])
    c = Conductivity.__new__(Conductivity)
    c.num_mols = [1, 1, 1]
    c.time = list(np.arange(nj) * 5e-11)
    tot = c.correlate_charge_flux(j)
    out["tot_flux"] = tot
    out["integral"] = c.integrate_charge_flux_correlation(tot)
    out["flux_time"] = np.asarray(c.time)
    np.savez_compressed(os.path.join(OUT, "acf.npz"), **out)
    print("acf: acf0", out["acf_wkt"][:, 0], "visc last", avg[-1])


# ----------------------------------------------------------------------------
def golden_cell17():
    """examples/mg_tfsi_dme_analysis.ipynb cell 17 on all 101 frames (only the table is kept)."""
    d = Diffusion(timestep=1, units="real", outputs_dir=DATA, diff_dir=tempfile.mkdtemp())
    with quiet():
        msd, _, _ = d.get_msd_from_dump(
            "dump.nvt.*.dump", msd_type="com", num_mols=NUM_MOLS, num_atoms_per_mol=NUM_ATOMS,
            mass=MASS, com_drift=True, avg_interval=True, tao_coeff=4)
        diff = d.calc_diff(msd, diff_names=["dme", "tfsi", "mg"])
    table = {
        "published": {  # printed in the notebook output
            "diffusion (m2/s)": [1.330522e-09, 1.976415e-10, 1.585219e-10],
            "std": [2.164493e-12, 2.162102e-12, 1.821829e-12],
            "R2": [0.999735, 0.988174, 0.986964],
        },
        "reference_here": {c: [float(x) for x in diff[c]] for c in diff.columns},
        "msd_columns": list(msd.columns),
        "msd_first_last": [msd.to_numpy()[0].tolist(), msd.to_numpy()[-1].tolist()],
    }
    with open(os.path.join(OUT, "cell17.json"), "wt") as fh:
        json.dump(table, fh, indent=1)
    print("cell17:", table["reference_here"])


# ----------------------------------------------------------------------------
def golden_host_logic():
    """Conductivity.detect_time_range / fit_curve / green_kubo and Viscosity.calc_avg_visc /
    fit_avg_visc on designed inputs (host-side logic around the kernels)."""
    rng = np.random.default_rng(20250328 + 6)
    out = {}
    n = 3000
    t = np.arange(n)
    flux = np.stack([
        np.exp(-t / 40.0) * np.cos(t / 9.0) + 1e-3 * rng.standard_normal(n) * (t > 2400),
        np.exp(-t / 90.0) + 2e-3 * rng.standard_normal(n) * ((t > 1500) & (t < 1700)),
    ])
    out["dtr_flux"] = flux
    out["dtr_tol"] = np.asarray(1e-2)
    out["dtr_range"] = np.array([Conductivity.detect_time_range(f, 1e-2) for f in flux])
    c = Conductivity.__new__(Conductivity)
    c.num_mols = [1]
    c.time = list(t * 2e-15)
    c.temp, c.volume = 298.15, 1.2e-25
    integ = c.integrate_charge_flux_correlation(flux)
    ave, rng_t = c.fit_curve(flux, integ, 1e-2)
    out["fc_integral"], out["fc_ave"] = integ, ave
    out["fc_time_range"] = np.array([list(r) for r in rng_t])
    out["fc_cond"] = c.green_kubo(ave)
    out["fc_temp"], out["fc_volume"] = np.asarray(c.temp), np.asarray(c.volume)
    # three replicate logs -> calc_avg_visc -> fit_avg_visc
    nlog, reps = 12000, 3
    press = np.zeros((reps, 3, nlog))
    for r in range(reps):
        e = rng.standard_normal((3, nlog))
        x = np.zeros((3, nlog))
        for k in range(1, nlog):
            x[:, k] = 0.9 * x[:, k - 1] + e[:, k]
        press[r] = x * 300.0
    out["log_press"] = press
    out["log_step"] = np.arange(nlog) * 2
    with tempfile.TemporaryDirectory() as tmp:
        for r in range(reps):
            tbl = np.column_stack([out["log_step"], press[r].T])
            mio.write_log(os.path.join(tmp, "log.rep%d" % r), tbl, ["Step", "Pxy", "Pxz", "Pyz"])
        v = Viscosity("log.rep*", 400, 64000.0, temp=300.0, timestep=1, acf_method="wkt", units="real",
                      working_dir=tmp)
        import glob as _glob
        order = [int(os.path.basename(f)[7:]) for f in _glob.glob(os.path.join(tmp, "log.rep*"))]
        with quiet():
            visc_avg, visc_data, acf_data, tm = v.calc_avg_visc(output_all_data=True)
        out["visc_rep_order"] = np.asarray(order)
        out["visc_avg"] = np.stack(visc_avg)
        out["visc_time"] = tm
        out["visc_cutoff"], out["visc_volume"], out["visc_temp"] = (
            np.asarray(400), np.asarray(64000.0), np.asarray(300.0))
        try:
            out["visc_fit"] = np.asarray(v.fit_avg_visc(visc_avg, initial_guess=[1e-8, 0.5, 50.0, 500.0]))
        except Exception as exc:  # data dependent; record that the reference could not fit
            print("fit_avg_visc failed in the reference:", repr(exc))
            out["visc_fit"] = np.asarray(np.nan)
    np.savez_compressed(os.path.join(OUT, "host_logic.npz"), **out)
    print("host_logic: dtr", out["dtr_range"].tolist(), "cond", out["fc_cond"], "visc_fit", out["visc_fit"])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["synth", "acf", "small", "c1", "cell17", "host", "residence", "inter"]
    if "residence" in which:
        golden_residence()
    if "inter" in which:
        golden_inter_rdf()
    if "host" in which:
        golden_host_logic()
    if "synth" in which:
        golden_synth_rdf()
    if "acf" in which:
        golden_acf()
    if "small" in which:
        golden_small_md()
    if "c1" in which:
        golden_c1_rdf()
    if "cell17" in which:
        golden_cell17()
