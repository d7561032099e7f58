"""
Green-Kubo shear viscosity from LAMMPS logs — drop-in for
/root/reference/mdproptools/dynamical/viscosity.py (class `Viscosity`, same method names, argument
order, defaults: viscosity.py:45-55, 87, 123, 139, 155, 193, 239, 382).

What runs where
  GPU (libmdhip.so): the pressure-tensor autocorrelation, both "wkt" (zero-padded FFT,
      viscosity.py:110-115) and "brute_force" (direct lag sums, viscosity.py:103-108), and the
      cumulative trapezoid behind the running integral (viscosity.py:151).
  Host: log parsing, unit factors, replicate averaging, the double-exponential fit and the bootstrap
      (scipy.optimize.curve_fit, as upstream).
"""

import glob
import os
import random

import numpy as np
from scipy import optimize

from .. import backend
from ..common import constants
from ..io import parse_lammps_log

TENSOR_LABELS = ["Pxy", "Pxz", "Pyz"]


class Viscosity:
    """Green-Kubo viscosity with replicate averaging, double-exponential extrapolation and bootstrap."""

    def __init__(self, log_pattern, cutoff_time, volume, temp=298.15, timestep=1, acf_method="wkt",
                 units="real", working_dir=None):
        """
        log_pattern: LAMMPS log file pattern (replicates); cutoff_time: step before which data is
        ignored; volume in `units`; temp [K]; timestep in `units`; acf_method 'wkt' or 'brute_force'.
        """
        self.log_pattern = log_pattern
        self.cutoff_time = cutoff_time
        self.units = units
        self.volume = volume * constants.DISTANCE_CONVERSION[self.units] ** 3
        self.temp = temp
        self.timestep = timestep
        self.acf_method = acf_method
        self.working_dir = working_dir or os.getcwd()
        self.time = None
        self.step_to_s = self.timestep * constants.TIME_CONVERSION[self.units]

    @staticmethod
    def autocorrelate(series, method):
        """acf[k] = sum_t s[t+k] s[t] / (n-k) by 'wkt' (FFT) or 'brute_force' (direct), on the GPU."""
        if method == "brute_force":
            how = backend.XCORR_DIRECT
        elif method == "wkt":
            how = backend.XCORR_FFT
        else:
            raise ValueError("Method string input not recognized")
        return backend.xcorr(np.ascontiguousarray(series, dtype=np.float64), method=how)

    @staticmethod
    def exp_func(t, A, alpha, tau1, tau2):
        """Double exponential A a t1 (1 - e^{-t/t1}) + A (1-a) t2 (1 - e^{-t/t2}) (10.1021/jp062885s)."""
        return A * alpha * tau1 * (1 - np.exp(-t / tau1)) + A * (1 - alpha) * tau2 * (1 - np.exp(-t / tau2))

    def calc_visc(self, acf, dt):
        """Running Green-Kubo integral: V / (kB T) * cumulative trapezoid of the acf (viscosity.py:139-153)."""
        integral = backend.cumtrapz(np.ascontiguousarray(acf, dtype=np.float64), dt)
        return np.multiply(self.volume / (constants.BOLTZMANN * self.temp), integral)

    def _calc_3d_visc(self, log_df):
        """(mean viscosity, per-component viscosity [3, n-1], acf [3, n]) of one thermo table."""
        if self.units not in constants.SUPPORTED_UNITS:
            raise KeyError("Unit type not supported. Supported units are: " + str(constants.SUPPORTED_UNITS))
        time_data = log_df["Step"] * self.step_to_s
        delta_t = time_data.iloc[1] - time_data.iloc[0]
        series = np.stack([log_df[label].to_numpy(dtype=np.float64) for label in TENSOR_LABELS])
        how = {"wkt": backend.XCORR_FFT, "brute_force": backend.XCORR_DIRECT}.get(self.acf_method)
        if how is None:
            raise ValueError("Method string input not recognized")
        # one library call: the series go to the GPU once, acf -> x conv^2 -> cumtrapz -> x V/(kB T) -> mean over
        # the three components happen there (each factor one multiplication of the finished value, as upstream's
        # `acf * conv ** 2` and `np.multiply(V / (kB T), integral)`: viscosity.py:152, 182), three arrays come back
        acf_data, viscosity_data, viscosity_average = backend.green_kubo(
            series, method=how, acf_scale=constants.PRESSURE_CONVERSION[self.units] ** 2, dx=delta_t,
            integral_scale=self.volume / (constants.BOLTZMANN * self.temp), want_mean=True)
        return viscosity_average, viscosity_data, acf_data

    def calc_avg_visc(self, output_all_data=False):
        """Viscosity of every replicate log after `cutoff_time` (viscosity.py:193-237)."""
        from .. import dist as D

        files = glob.glob(f"{self.working_dir}/{self.log_pattern}")
        # under torch.distributed the replicates (independent logs) are dealt to the ranks: every rank parses
        # and correlates its share — plus the first log, which fixes the cutoff row and the time axis —
        # and the per-replicate arrays are all-gathered in file order
        sharded = D.is_distributed() and len(files) >= D.rank_world()[1]
        mine = D.shard_items(files) if sharded else files
        first = parse_lammps_log(files[0])[0]
        logs = [first if f == files[0] else parse_lammps_log(f)[0] for f in mine]
        start = first.index.get_loc(first[first["Step"] == self.cutoff_time].index[0])
        visc_avg, visc_data, acf_data = [], [], []
        for k, log_df in enumerate(logs):
            print(f"Processing replicate number {k + 1} out of {len(logs)}")
            avg, data, acf = self._calc_3d_visc(log_df.iloc[start:])
            visc_avg.append(avg)
            visc_data.append(data)
            acf_data.append(acf)
        if sharded:
            visc_avg = list(D.allgather_var(np.stack(visc_avg)))
            visc_data = list(D.allgather_var(np.stack(visc_data)))
            acf_data = list(D.allgather_var(np.stack(acf_data)))
        self.time = np.array(first["Step"][: len(visc_avg[0]) - 1]) * self.timestep
        if output_all_data:
            return visc_avg, visc_data, acf_data, self.time
        return visc_avg

    def fit_avg_visc(self, visc_avg, initial_guess=[1e-10, 0.8, 1.1e4, 1.1e4], plot=False,
                     plot_file="viscosity.png"):
        """
        Average the replicates, fit the double exponential between t > 2000 (time units) and the first
        time the replicate spread reaches 0.4 of the mean (weights 1/std**0.5), return its
        infinite-time value A a t1 + A (1-a) t2 (viscosity.py:239-380).
        """
        visc = np.average(visc_avg, axis=0)
        std = np.std(visc_avg, axis=0)
        i0 = np.where(self.time > 2000)[0][0]
        i1 = np.where(std >= 0.4 * visc)[0][0]
        t_fit, v_fit = self.time[i0:i1], visc[i0:i1]
        popt, _ = optimize.curve_fit(
            self.exp_func, t_fit, v_fit, sigma=1 / std[i0:i1] ** 0.5,
            bounds=(0, [max(v_fit), 1, 5 * self.time[i1], 5 * self.time[i1]]), p0=initial_guess,
            maxfev=1000000)
        viscosity = popt[0] * popt[1] * popt[2] + popt[0] * (1 - popt[1]) * popt[3]
        if plot:
            self._plot(visc_avg, visc, std, i0, i1, [self.exp_func(t, *popt) for t in t_fit], plot_file)
        return viscosity

    def _plot(self, visc_avg, visc, std, i0, i1, fit, plot_file):
        import matplotlib

        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt

        from ..utilities.plots import set_axis

        t_ns = self.time * self.step_to_s * 10 ** 9
        cmap = plt.get_cmap("Paired")
        fig, (ax1, ax2, ax3) = plt.subplots(1, 3, figsize=[20, 5])
        for k, arr in enumerate(visc_avg):
            ax1.plot(t_ns, arr[0:-1], linewidth=2, color=cmap(k / max(1, len(visc_avg))))
        ax1.plot(t_ns, visc[0:-1], linewidth=2, color="black")
        ax1.axvline(t_ns[i1], linewidth=2, color="black", linestyle="--")
        ax1.set_ylabel(r"$\mathrm{\mu \ (Pa.s)}$", fontsize=18)
        ax2.plot(t_ns, std[0:-1], linewidth=2, color="black")
        ax2.set_ylabel(r"$\mathrm{\sigma \ (Pa.s)}$", fontsize=18)
        ax3.plot(t_ns[i0:i1], visc[i0:i1], linewidth=2, color="red", label="data")
        ax3.plot(t_ns[i0:i1], fit, linewidth=2, color="black", label="fit")
        ax3.legend(fontsize=16, loc="lower right", frameon=False)
        ax3.set_ylabel(r"$\mathrm{\mu \ (Pa.s)}$", fontsize=18)
        for ax in (ax1, ax2, ax3):
            set_axis(ax, axis="both")
            ax.set_xlabel(r"$\mathrm{Time, 10^9 (s)}$", fontsize=18)
        fig.tight_layout(pad=3)
        fig.savefig(f"{self.working_dir}/{plot_file}", bbox_inches="tight", pad_inches=0.1)
        plt.close(fig)

    def bootstrapping(self, visc_avg, num_replicates, tot_replicates, initial_guess=[1e-10, 0.8, 1.1e4, 1.1e4],
                      plot=True):
        """
        `tot_replicates` fits of `num_replicates` replicates drawn without repetition; returns the mean
        and standard deviation of the fitted viscosities (viscosity.py:382-434).
        """
        picks = np.zeros((tot_replicates, num_replicates), dtype=int)
        for i in range(tot_replicates):
            picks[i] = random.sample(range(len(visc_avg)), num_replicates)
        samples = np.array(visc_avg)[picks]
        fitted = []
        for k, sample in enumerate(samples):
            print(f"Fitting viscosity sample {k + 1} out of {len(samples)}")
            fitted.append(self.fit_avg_visc(visc_avg=sample, initial_guess=initial_guess, plot=plot,
                                            plot_file=f"viscosity_{k + 1}.png"))
        return np.average(fitted), np.std(fitted)
