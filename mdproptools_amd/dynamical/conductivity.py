"""
Green-Kubo ionic conductivity from LAMMPS dumps — drop-in for
/root/reference/mdproptools/dynamical/conductivity.py (class `Conductivity`, same method names,
argument order, defaults and files written: conductivity.py:51-62, 98, 117, 167, 197, 216, 234, 259, 276).

What runs where
  GPU (libmdhip.so): the per-frame molecular charge flux for ALL frames in one call
      (`conductivity_loop`, _conductivity.py:7-36 — the step the reference marks as its slowest and
      spreads over a process pool), every flux cross-correlation (conductivity.py:97-114, batched)
      and the running integrals (conductivity.py:216-232).
  Host: parsing, plateau detection (pandas, conductivity.py:116-165), the final Green-Kubo factor.
"""

import glob
import os

import numpy as np
import pandas as pd

from .. import backend
from ..common import constants
from ..common.com_mols import molecule_layout
from ..io import parse_lammps_dumps

# True: get_charge_flux parses into page-locked staging batches and runs the fused flux kernel on each while the next
# one is being parsed (mdproptools_amd/stream.py). False: every frame is parsed first (round-1 route).
STREAM = True


class Conductivity:
    """Green-Kubo ionic conductivity (total and per molecule type) following 10.1063/1.4890741."""

    def __init__(self, filename, num_mols, num_atoms_per_mol, volume, mass=None, temp=298.15, timestep=1,
                 units="real", working_dir=None):
        """
        filename: dump file pattern; num_mols / num_atoms_per_mol: molecules per type and atoms per
        molecule in dump order; volume: box volume in `units`; mass: per-atom-type masses (or None to
        read the dump's mass column); temp [K]; timestep in `units`; working_dir: where the dumps are.
        """
        self.working_dir = working_dir or os.getcwd()
        self.filename = filename
        self.dumps = parse_lammps_dumps(f"{self.working_dir}/{self.filename}")
        self.mass = mass
        self.num_mols = num_mols
        self.num_atoms_per_mol = num_atoms_per_mol
        self.units = units
        self.volume = volume * constants.DISTANCE_CONVERSION[self.units] ** 3  # m^3
        self.temp = temp
        self.timestep = timestep
        self.time = []  # seconds, one entry per frame, filled by get_charge_flux

    @staticmethod
    def correlate(a, b):
        """c[k] = sum_t a[t+k] b[t] / (n-k), evaluated with zero-padded FFTs on the GPU."""
        return backend.xcorr(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64),
                             method=backend.XCORR_FFT)

    @staticmethod
    def detect_time_range(flux, tol):
        """
        (start, end) indices of the longest stretch where the correlation function is flat: the series
        is cut into blocks, the block standard deviations are scaled by their own spread, compared
        with `tol`, smoothed with a centred rolling median and the longest run of ones is returned
        (conductivity.py:116-165).
        """
        flux = pd.Series(flux, name="flux")
        block = max(int(len(flux) / 10000), 5)
        labels = np.arange(len(flux)) // block
        block_std = flux.groupby(labels).transform("std")
        spread = block_std.std()
        flat = ((block_std / (spread if spread else 1)) < tol).astype("int").to_frame()
        smooth = (flat.rolling(window=4 * block + 1, min_periods=3 * block + 1, center=True)
                  .median().fillna(0)["flux"].to_numpy())
        on = smooth == 1
        runs, start = [], None
        for k in range(len(smooth)):
            if on[k] and start is None:
                start = k
            elif smooth[k] < 1 and start is not None:
                runs.append((start, k))
                start = None
        if start is not None:
            runs.append((start, len(smooth) - 1))
        best, best_len = None, 0
        for run in runs:
            if run[1] - run[0] > best_len:
                best, best_len = run, run[1] - run[0]
        if best is None:
            raise TypeError("list indices must be integers or slices, not NoneType")  # as the reference
        return best

    def get_charge_flux(self):
        """Charge flux J[3, n_types, n_frames] in SI units; also fills `self.time` (conductivity.py:167-195)."""
        n_expected = len(glob.glob(f"{self.working_dir}/{self.filename}"))
        seg_off, mol_type, _ = molecule_layout(self.num_mols, self.num_atoms_per_mol)
        from .. import io as mio

        vel, steps = [], []
        m = q = None
        from .. import dist as D

        files = None
        if mio.USE_NATIVE_READER and STREAM:
            # text -> page-locked batches -> fused flux kernel, the next batch being parsed meanwhile (stream.py)
            files = D.my_files(f"{self.working_dir}/{self.filename}")
            got = self._flux_streamed(files, seg_off, mol_type)
            if got is not None:
                flux, steps = got
                return self._finish_flux(flux, steps, files, n_expected)
        if mio.USE_NATIVE_READER:
            def wanted(names):
                return ["vx", "vy", "vz", "q"] + (["type"] if self.mass else ["mass"])

            # under torch.distributed every rank parses and reduces its own share of the files; the per-frame
            # flux vectors (3 x n_types doubles) are all-gathered below
            files = D.my_files(f"{self.working_dir}/{self.filename}")
            frames = ((ts, planes[0:3], planes[3], planes[4]) for ts, _b, _l, _n, planes in
                      mio.iter_native_frames(f"{self.working_dir}/{self.filename}", wanted, sort_by="id",
                                             files=files))
        else:
            def from_pandas():
                for dump in self.dumps:
                    data = dump.data.sort_values(by=["id"])
                    yield (dump.timestep,
                           np.ascontiguousarray(data[["vx", "vy", "vz"]].to_numpy(dtype=np.float64).T),
                           data["q"].to_numpy(dtype=np.float64),
                           data["type" if self.mass else "mass"].to_numpy(dtype=np.float64))

            frames = from_pandas()
        for ts, v, qcol, tm in frames:
            if seg_off[-1] != v.shape[1]:
                raise ValueError(f"Length of values ({int(seg_off[-1])}) does not match length of index "
                                 f"({v.shape[1]})")
            if m is None:
                m = np.asarray(self.mass, dtype=np.float64)[tm.astype(np.int64) - 1] if self.mass else tm
                q = qcol
            vel.append(np.ascontiguousarray(v))
            steps.append(ts * constants.TIME_CONVERSION[self.units])
        flux = None
        if vel:
            flux = backend.charge_flux(np.stack(vel), m, q, seg_off, (mol_type - 1).astype(np.int32),
                                       len(self.num_mols), constants.VELOCITY_CONVERSION[self.units],
                                       constants.CHARGE_CONVERSION[self.units])
        return self._finish_flux(flux, steps, files, n_expected)

    def _flux_streamed(self, files, seg_off, mol_type):
        """(flux [3, n_types, F_local] or None, steps) through the frame stream; None when the dumps need the general
        route (compressed text, a column missing: the general route raises the reference's message)."""
        from .. import io as mio
        from .. import stream as S

        pattern = f"{self.working_dir}/{self.filename}"
        mine = files if files is not None else mio._sorted_matches(pattern)
        if not mine or any(str(f).endswith(".gz") for f in mine):
            return None
        nd = mio.NativeDumpFile(mine[0])
        try:
            names = nd.header(0)[4] if nd.n_frames else []
        finally:
            nd.close()
        second = "type" if self.mass else "mass"
        if not {"id", "q", second, "vx", "vy", "vz"} <= set(names):
            return None
        m = q = None
        parts, steps = [], []
        # the staging batch carries the charge, ONE more per-atom attribute and the three velocity planes
        for batch in S.FrameStream(pattern, files=mine, columns=("q", second, "vx", "vy", "vz")):
            n = batch.xyz.shape[2]
            if seg_off[-1] != n:
                raise ValueError(f"Length of values ({int(seg_off[-1])}) does not match length of index ({n})")
            if m is None:  # masses and charges of the first frame, as the general route takes them
                tm = batch.types[0]
                m = np.asarray(self.mass, dtype=np.float64)[tm.astype(np.int64) - 1] if self.mass else tm.copy()
                q = batch.ids[0].copy()
            parts.append(backend.charge_flux(batch.xyz, m, q, seg_off, (mol_type - 1).astype(np.int32),
                                             len(self.num_mols), constants.VELOCITY_CONVERSION[self.units],
                                             constants.CHARGE_CONVERSION[self.units]))
            steps.extend((batch.timesteps * constants.TIME_CONVERSION[self.units]).tolist())
        if not parts:
            return None if files is None else (None, steps)
        return np.concatenate(parts, axis=2), steps

    def _finish_flux(self, flux, steps, files, n_expected):
        from .. import dist as D

        if files is not None:
            if flux is None:
                raise ValueError("this rank holds no frame: use at most as many ranks as there are dump files")
            flux = np.moveaxis(D.allgather_var(np.ascontiguousarray(np.moveaxis(flux, 2, 0))), 0, 2)
            steps = list(D.allgather_var(np.asarray(steps, dtype=np.float64)))
        n_frames = 0 if flux is None else flux.shape[2]
        j = np.zeros((3, len(self.num_mols), max(n_expected, n_frames)))
        if flux is not None:
            j[:, :, :n_frames] = flux
        for s in steps:
            self.time.append(s * self.timestep)
        return j

    def correlate_charge_flux(self, flux):
        """tot_flux[i] = sum_j sum_k corr(J[k,i], J[k,j]); last row = sum over i (conductivity.py:197-214)."""
        n_types = len(self.num_mols)
        a = np.stack([flux[k, i] for i in range(n_types) for jj in range(n_types) for k in range(flux.shape[0])])
        b = np.stack([flux[k, jj] for i in range(n_types) for jj in range(n_types) for k in range(flux.shape[0])])
        corr = backend.xcorr(a, b, method=backend.XCORR_FFT).reshape(n_types, n_types * flux.shape[0], -1)
        tot_flux = np.zeros((n_types + 1, flux.shape[2]))
        for i in range(n_types):
            for c in corr[i]:  # same accumulation order as the reference's triple loop
                tot_flux[i, :] += c
                tot_flux[-1, :] += c
        return tot_flux

    def integrate_charge_flux_correlation(self, tot_flux):
        """Running trapezoid integral of every row, first value 0 (conductivity.py:216-232)."""
        delta = self.time[1] - self.time[0]
        return backend.cumtrapz(np.asarray(tot_flux, dtype=np.float64), delta, leading_zero=True)

    def fit_curve(self, tot_flux, integral, tol):
        """Average of each integral over its detected plateau, and the plateau's time range."""
        ave = np.zeros(len(integral))
        time_range = np.zeros(len(integral), dtype=object)
        for i in range(len(integral)):
            lo, hi = self.detect_time_range(tot_flux[i], tol=tol)
            ave[i] = np.average(integral[i][lo:hi])
            time_range[i] = (self.time[lo], self.time[hi])
        return ave, time_range

    def green_kubo(self, ave):
        """sigma = <integral> / (3 kB T V) (conductivity.py:259-274)."""
        return np.array([a / 3 / constants.BOLTZMANN / self.temp / self.volume for a in ave])

    def calc_cond(self, tol=1e-4, plot=False, save=False):
        """
        Whole chain: charge flux, correlation, integral, plateau average, conductivity [S/m] per molecule
        type followed by the total. save writes charge_flux.csv, integral.csv, conductivity.csv; plot
        writes conductivity.png, all into working_dir (conductivity.py:276-397).
        """
        j = self.get_charge_flux()
        tot_flux = self.correlate_charge_flux(j)
        integral = self.integrate_charge_flux_correlation(tot_flux)
        ave, time_range = self.fit_curve(tot_flux, integral, tol)
        cond = self.green_kubo(ave)
        if plot:
            self._plot(tot_flux, integral, time_range)
        if save:
            t = np.array([self.time])
            header = "t," + ",".join(str(i + 1) for i in range(len(tot_flux) - 1)) + ",tot"
            np.savetxt(f"{self.working_dir}/charge_flux.csv", np.append(t, tot_flux, axis=0).T, delimiter=",",
                       header=header, comments="")
            np.savetxt(f"{self.working_dir}/integral.csv", np.append(t, integral, axis=0).T, delimiter=",",
                       header=header, comments="")
            cond = np.asarray([[r[0] for r in time_range], [r[1] for r in time_range], cond])
            np.savetxt(f"{self.working_dir}/conductivity.csv", cond.T, delimiter=",",
                       header="start_t,end_t,cond", comments="")
        return cond

    def _plot(self, tot_flux, integral, time_range):
        import matplotlib

        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt

        from ..utilities.plots import set_axis

        t_ns = np.array(self.time) * 10 ** 9
        cmap = plt.get_cmap("Paired")
        fig, (ax1, ax2) = plt.subplots(1, 2, figsize=(20, 5))
        for i in range(len(tot_flux) - 1):
            ax1.plot(t_ns, tot_flux[i], linewidth=2, color=cmap(i / 10))
            ax2.plot(t_ns, integral[i], linewidth=2, color=cmap(i / 10), label=i + 1)
        ax1.plot(t_ns, tot_flux[-1], linewidth=2, color="black")
        ax2.plot(t_ns, integral[-1], linewidth=2, color="black", label="total")
        ax1.set_ylabel(r"$\mathrm{\langle J(t)\cdot J(0)\rangle}$", fontsize=18)
        ax2.set_ylabel(r"$\mathrm{\int_{0}^{t}\langle J(t')\cdot J(0)\rangle dt'}$", fontsize=18)
        ax2.legend(fontsize=16, loc="center left", bbox_to_anchor=(1, 0.5), frameon=False)
        for ax in (ax1, ax2):
            set_axis(ax, axis="both")
            for edge in time_range[-1]:
                ax.axvline(edge * 10 ** 9, linewidth=2, color="black", linestyle="--")
            ax.set_xscale("log")
            ax.set_xlabel(r"$\mathrm{Time, 10^9 (s)}$", fontsize=18)
        fig.tight_layout(pad=3)
        fig.savefig(f"{self.working_dir}/conductivity.png", bbox_inches="tight", pad_inches=0.1)
        plt.close(fig)

    def einstein(self):
        pass

    def nernst(self):
        pass
