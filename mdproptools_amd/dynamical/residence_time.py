"""
Residence time of neighbours in a coordination shell — drop-in for
/root/reference/mdproptools/dynamical/residence_time.py (class `ResidenceTime`: residence_time.py:40-58,
70-146, 148-200; same constructor arguments, method names, defaults, files written).

What runs where
  GPU (libmdhip.so, `mdhip_shell_residence`): for every relation the shell indicator h_ij(t) of all
      (central atom, shell atom) pairs of every frame with the reference's exact single-wrap distance
      (residence_time.py:96-106) and the sum over pairs of the autocovariance numerators
      sum_t h_ij(t) h_ij(t+k) (residence_time.py:111-131) as exact integers.
  Host: parsing, the pseudo-type relabelling (`_calc_atom_type`), normalisation (1/(n-k), 1/columns, /C(0)),
      the stretched-exponential fit (scipy.optimize.curve_fit, as upstream) and the CSV / PNG files.

Deliberate difference: with default ids (no num_mols / num_atoms_per_mol) the reference stops with a
broadcast ValueError (it hands `_calc_rsq(..., num_of_ids=0)` rows of [id, x, y, z]:
residence_time.py:96-101, rdf_cn.py:43); here that mode works and selects atoms by their LAMMPS type.
`Displacement` (residence_time.py:203-254, unfinished upstream) is not provided.
"""

import os

import numpy as np
import pandas as pd
from scipy.optimize import curve_fit
from scipy.special import gamma

from .. import backend
from ..structural.rdf_cn import _calc_atom_type, _load_frames

VERBOSE = False


def _say(*args):
    if VERBOSE:
        print(*args)


def _is_writer():
    """Only rank 0 writes files under torch.distributed (every rank computes the same tables)."""
    from .. import dist as D

    return D.is_writer()


class ResidenceTime:
    def __init__(self, r_cut, partial_relations, filename, dt=1, num_mols=None, num_atoms_per_mol=None,
                 working_dir=None):
        """
        r_cut: [lo, hi] per relation (a neighbour is counted when lo < r <= hi); partial_relations:
        [[central types...], [shell types...]]; filename: dump file or '*' pattern; dt: timestep in fs.
        """
        self.r_cut = r_cut
        self.relation_matrix = np.asarray(partial_relations).transpose()
        self.atom_pairs = []
        self.filename = filename
        self.dt = dt * 10 ** -3  # input dt in fs - convert to ps (residence_time.py:54)
        self.corr_df = None
        self.res_time_df = None
        self.num_mols = num_mols
        self.num_atoms_per_mol = num_atoms_per_mol
        self.working_dir = working_dir or os.getcwd()

    @staticmethod
    def _stretched_exp_function(x, a, tau_res, tau_short, beta):
        return a * np.exp(-((x / tau_res) ** beta)) + (1 - a) * np.exp(-x / tau_short)

    @staticmethod
    def _integrate_sum_exp(a, tau_res, tau_short, beta):
        return (a * tau_res * gamma(1 + 1 / beta)) + (1 - a) * tau_short

    def _labels(self, frame):
        """Type label of every (id-sorted) atom of one frame: LAMMPS type, or the index of the atom inside
        its molecule type when num_mols / num_atoms_per_mol are given (residence_time.py:85-93)."""
        if self.num_mols and self.num_atoms_per_mol:
            return _calc_atom_type(frame.ids, self.num_mols, self.num_atoms_per_mol)
        return frame.types

    def calc_auto_correlation(self):
        frames = _load_frames(self.filename)
        n = len(frames)
        correlation = {"Time (ps)": [fr.timestep * self.dt for fr in frames]}
        if n == 0:
            self.corr_df = pd.DataFrame.from_dict(correlation)
            return
        labels = self._labels(frames[0])
        for fr in frames[1:]:
            if fr.xyz.shape != frames[0].xyz.shape or not np.array_equal(self._labels(fr), labels):
                raise ValueError("every frame must hold the same atoms with the same types")
        xyz = np.stack([fr.xyz for fr in frames])  # [F,3,N], id-sorted
        box = np.asarray([fr.lengths for fr in frames], dtype=np.float64)
        lag_weight = (n - np.arange(n)).astype(np.float64)
        for kl in range(len(self.relation_matrix)):
            k, l = self.relation_matrix[kl]
            atom_pair = f"{k}-{l}"
            self.atom_pairs.extend([atom_pair] * n)  # the reference appends the label once per frame
            sel_k = np.flatnonzero(labels == k)
            sel_l = np.flatnonzero(labels == l)
            _say("relation", atom_pair, ":", len(sel_k), "central atoms,", len(sel_l), "shell atoms")
            xk = np.ascontiguousarray(xyz[:, :, sel_k])  # the library stages the two selections to the device
            xl = xk if k == l else np.ascontiguousarray(xyz[:, :, sel_l])
            counts, _ = backend.shell_residence(
                xk, xl, box, self.r_cut[kl][0] ** 2, self.r_cut[kl][1] ** 2, exclude_diagonal=bool(k == l))
            total_columns = float(len(sel_k) * len(sel_l))
            with np.errstate(invalid="ignore", divide="ignore"):
                corr = counts.astype(np.float64) / lag_weight / total_columns  # mean unbiased autocovariance
                corr = corr / corr[0]                                           # residence_time.py:142
            correlation[atom_pair] = corr
        self.corr_df = pd.DataFrame.from_dict(correlation)
        if _is_writer():
            self.corr_df.to_csv(self.working_dir + "/auto_correlation.csv")

    def fit_auto_correlation(self, cut_percent=0.9, plot=True):
        residence_time = {}
        corr_data = self.corr_df.head(int(len(self.corr_df) * cut_percent))  # first part of the data
        for col in corr_data:
            if col == "Time (ps)":
                continue
            x = corr_data["Time (ps)"].values
            y = corr_data[col].values
            popt, _ = curve_fit(self._stretched_exp_function, x, y,
                                bounds=([0, 0, 0, 0.1], [np.inf, np.inf, np.inf, 1]), maxfev=5000)
            a, tau_res, tau_short, beta = popt
            residence_time[col] = [a, tau_res, tau_short, beta, self._integrate_sum_exp(a, tau_res, tau_short, beta)]
            if plot:
                self._plot_fit(corr_data, col, popt)
        print("Finished computing residence time")
        self.res_time_df = pd.DataFrame(residence_time)
        self.res_time_df.index = ["a", "tau_res", "tau_short", "beta", "r (ps)"]
        if _is_writer():
            self.res_time_df.to_csv(self.working_dir + "/residence_time.csv")
        return residence_time

    def _plot_fit(self, corr_data, col, popt):
        import matplotlib

        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt

        from ..utilities.plots import set_axis

        fig, ax = plt.subplots(figsize=(8, 6))
        set_axis(ax)
        t = corr_data["Time (ps)"]
        ax.scatter(t, corr_data[col], color="red", label="original")
        ax.plot(t, self._stretched_exp_function(t.values, *popt), color="black", label="fit")
        ax.legend(frameon=False, fontsize=20)
        ax.set_xlabel("Time (ps)", fontsize=20)
        ax.set_ylabel("C(t)", fontsize=20)
        if _is_writer():
            fig.savefig(self.working_dir + f"/{col}_fit.png", bbox_inches="tight", pad_inches=0.1)
        plt.close()
