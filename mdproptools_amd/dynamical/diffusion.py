"""
Mean-square displacement and Einstein-relation diffusion coefficients — drop-in for
/root/reference/mdproptools/dynamical/diffusion.py (class `Diffusion`, same method names,
argument order, defaults, returned DataFrames and files written: diffusion.py:38, 101-111, 241,
267-276, 410-412).

What runs where
  GPU (libmdhip.so): molecule centres of mass (`calc_com`), the single-origin displacement
      reduction behind `msd` / `msd_all` (diffusion.py:212-218, `mdhip_msd_pairs`) and the
      fixed-lag per-entity reduction behind `msd_int` (diffusion.py:225-237, `mdhip_msd_windows`).
  Host: parsing, the per-type drift correction (small: F x n_types x 3), assembling the pandas
      objects, the through-origin least-squares fit (closed form of what statsmodels' OLS returns).

Tolerance: means are tree sums on the GPU vs pandas' compensated sums: rtol 1e-10 (observed ~1e-15).
"""

import os

import numpy as np
import pandas as pd

from .. import backend
from ..common import constants
from ..common.com_mols import atom_masses, molecule_layout
from ..io import parse_lammps_dumps, parse_lammps_log  # noqa: F401  (parse_lammps_log: API parity)
from ..utilities.log import concat_log


def _writer():
    """Only rank 0 writes result files under torch.distributed."""
    from .. import dist as D

    return D.is_writer()

_COORDS = ["xu", "yu", "zu"]
_DISPS = ["dx2", "dy2", "dz2"]

# True: get_msd_from_dump parses into page-locked staging buffers and moves every batch of frames to the GPU while
# the next one is being parsed (mdproptools_amd/stream.py); the trajectory is never stacked on the host and the
# reductions read ONE device copy. False: every frame is parsed first (round-1 route).
STREAM = True
STREAM_BATCH_BYTES = None  # coordinates per staging batch (None: stream.DEFAULT_BATCH_BYTES)
# Under torch.distributed: True = every rank returns the whole `msd_all` (its F x E value columns are gathered, device to
# device on RCCL) — the frame one process would return; False = every rank returns the rows of ITS frames only.
MSD_ALL_ON_EVERY_RANK = True


def _is_device(r):
    return hasattr(r, "is_cuda")


def _take_frames(r, idx):
    """r[idx] as a contiguous block, host array or device tensor alike."""
    if _is_device(r):
        import torch

        return r[torch.as_tensor(np.asarray(idx), device=r.device)].contiguous()
    return np.ascontiguousarray(r[idx])


class _OlsThroughOrigin:
    """y = b t without intercept: what `sm.OLS(y, t).fit()` exposes and calc_diff reads (diffusion.py:323-329)."""

    def __init__(self, y, t):
        self.y = np.asarray(y, dtype=np.float64)
        self.t = np.asarray(t, dtype=np.float64)
        sxx = float(self.t @ self.t)
        self.slope = float(self.t @ self.y) / sxx
        resid = self.y - self.slope * self.t
        self.rss = float(resid @ resid)
        self.nobs = len(self.t)
        self.bse = np.sqrt(self.rss / (self.nobs - 1) / sxx)
        self.rsquared = 1.0 - self.rss / float(self.y @ self.y)  # uncentred: no constant in the model

    def predict(self):
        return self.slope * self.t

    def summary(self):
        return ("OLS through the origin (closed form)\n"
                f"  observations : {self.nobs}\n  slope        : {self.slope!r}\n"
                f"  std err      : {self.bse!r}\n  R-squared    : {self.rsquared!r} (uncentred)\n"
                f"  residual SS  : {self.rss!r}\n")


class Diffusion:
    """
    Diffusion coefficients from the mean square displacement of a LAMMPS trajectory (dumps) or from
    msd columns of a LAMMPS log, via the Einstein relation.
    """

    def __init__(self, timestep=1, units="real", outputs_dir=None, diff_dir=None):
        """
        timestep: MD timestep in the units of `units`; units: LAMMPS unit style; outputs_dir: where the
        dump/log files are; diff_dir: where results (.csv, .txt, .png) go. Both default to the cwd.
        """
        self.units = units
        if self.units not in constants.SUPPORTED_UNITS:
            raise KeyError("Unit type not supported. Supported units are: " + str(constants.SUPPORTED_UNITS))
        self.outputs_dir = outputs_dir or os.getcwd()
        self.diff_dir = diff_dir or os.getcwd()
        self.timestep = timestep

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _prepare_unwrapped_coords(dump):
        """Make xu, yu, zu available, from x + ix*L when they were not dumped (diffusion.py:62-81)."""
        cols = dump.data.columns
        if "zu" not in cols:  # the reference's `"xu" and "yu" and "zu" not in ...` tests only zu
            assert "z" in cols, "Missing wrapped and unwrapped coordinates (x y z xu yu zu)"
            assert "iz" in cols, (
                "Missing unwrapped coordinates (xu yu zu) and box location (ix iy iz) for converting "
                "wrapped coordinates (x y z) into unwrapped coordinates. ")
            for axis, (lo, hi) in zip("xyz", dump.box.bounds):
                length = hi - lo
                dump.data[axis + "u"] = dump.data[axis].add(dump.data["i" + axis].multiply(length))
        return dump

    def _calculate_type_com(self, df, ind):
        """Mass-weighted mean of xu, yu, zu per group of `ind` (index levels or columns of `df`): a DataFrame indexed by
        `ind` with the three coordinate columns (diffusion.py:83-89). The product path computes the same per-type centres
        on its arrays (`_type_com`, mdhip_segment_com); this host form exists for callers that subclass and call it — the
        sums run per group in pandas' order instead of a BLAS dot, so it agrees with the reference to rounding."""
        cols = ["xu", "yu", "zu"]
        weighted = df[["mass"]].copy()
        weighted[cols] = df[cols].to_numpy() * df["mass"].to_numpy()[:, None]
        sums = weighted.groupby(ind).sum()
        return sums[cols].div(sums["mass"], axis=0)

    def _modify_dump_coordinates(self, msd_df):
        """Subtract from every molecule the drift of its TYPE's centre of mass since time 0 (diffusion.py:91-96);
        `msd_df` is indexed by ("Time (s)", "type", "mol_id") with columns mass, xu, yu, zu."""
        cols = ["xu", "yu", "zu"]
        ref_com = self._calculate_type_com(msd_df.xs(0, level=0), ["type"])
        com = self._calculate_type_com(msd_df, ["Time (s)", "type"])
        drift = com.sub(ref_com.reindex(com.index.get_level_values("type")).to_numpy())
        rows = pd.MultiIndex.from_arrays([msd_df.index.get_level_values("Time (s)"), msd_df.index.get_level_values("type")])
        msd_df.loc[:, cols] = msd_df[cols].to_numpy() - drift.reindex(rows).to_numpy()
        return msd_df

    def detect_linear_region():
        pass

    # ------------------------------------------------------------------------------------------
    def _entity_frames(self, filename, msd_type, num_mols, num_atoms_per_mol, mass):
        """Parse this process's frames and reduce them to entity coordinates [F_local, 3, E] in LAMMPS length units.
        Returns (times, r, meta, sharded): under torch.distributed every rank parses ITS share of the files and
        `sharded` is True — r then holds only those frames (get_msd_from_dump reduces them where they are)."""
        from .. import io as mio

        if msd_type not in ("allatom", "com"):
            raise ValueError("msd_type must be 'allatom' or 'com'.")
        from .. import dist as D

        times, planes = [], []
        ids = atom_mass = None
        pattern = f"{self.outputs_dir}/{filename}"
        # under torch.distributed every rank parses its own share of the files (parsing is the bottleneck);
        # the reduced frames are all-gathered below and the rest runs replicated
        files = D.my_files(pattern) if mio.USE_NATIVE_READER else None
        if STREAM and mio.USE_NATIVE_READER:
            got = self._entity_frames_streamed(pattern, files, msd_type, num_mols, num_atoms_per_mol, mass)
            if got is not None:
                return got
        err = None
        try:
            for step, names, cols in self._frame_columns(pattern, msd_type, mass, mio.USE_NATIVE_READER, files=files):
                if ids is None:
                    ids = cols["id"]
                planes.append(np.ascontiguousarray(np.stack([cols["xu"], cols["yu"], cols["zu"]])))
                if msd_type == "com":
                    m = cols["mass"] if not mass else np.asarray(mass, dtype=np.float64)[cols["type"].astype(np.int64) - 1]
                    if atom_mass is None:
                        atom_mass = m
                    elif not np.array_equal(atom_mass, m):
                        raise ValueError("atom masses change between frames")
                times.append(step * self.timestep * constants.TIME_CONVERSION[self.units])
        except Exception as e:  # noqa: BLE001
            if files is None:
                raise
            err = e  # (a rank's own parse error: the ranks agree on it before anyone enters a collective)
        if files is not None:
            D.raise_together(err, "parsing its dump files")
        times = np.asarray(times, dtype=np.float64)
        sharded = files is not None and D.is_distributed()
        if files is not None:
            D.require_all_nonempty(len(planes), "dump file")  # every rank raises, or none
        if msd_type == "com":
            seg = molecule_layout(num_mols, num_atoms_per_mol)
            if planes and seg[0][-1] != planes[0].shape[1]:
                raise ValueError(f"Length of values ({int(seg[0][-1])}) does not match length "
                                 f"of index ({planes[0].shape[1]})")
            com, seg_mass, _ = backend.segment_com(np.stack(planes), atom_mass, seg[0])
            return times, com, dict(type=seg[1], mol_id=seg[2], mass=seg_mass), sharded
        r = np.stack(planes)
        # (id: an integer column, as the reference's parser reads it)
        return times, r, dict(id=np.asarray(ids).astype(np.int64)), sharded

    def _entity_frames_streamed(self, pattern, files, msd_type, num_mols, num_atoms_per_mol, mass):
        """The same (times, r [F,3,E], meta) with r a DEVICE tensor: frames go text -> page-locked batch -> GPU while
        the next batch is parsed; 'com' reduces every batch to molecule centres on the way (`mdhip_segment_com` with a
        device destination), so only [F,3,M] stays. None when the dumps need the general route (wrapped coordinates
        to unwrap, compressed text, a column missing — whose error the general route raises in the reference's
        words)."""
        try:
            import torch
        except ImportError:  # libmdhip.so runs on the system ROCm runtime without torch (_lib.py): general route
            return None

        from .. import dist as D
        from .. import io as mio
        from .. import stream as S
        from .._lib import default_context

        mine = files if files is not None else mio._sorted_matches(pattern)
        if not mine or any(str(f).endswith(".gz") for f in mine):
            return None
        nd = mio.NativeDumpFile(mine[0])
        try:
            names = nd.header(0)[4] if nd.n_frames else []
        finally:
            nd.close()
        # the staging batch carries id, ONE per-atom attribute and the three coordinate planes
        second = "id" if msd_type == "allatom" else ("type" if mass else "mass")
        if not {"id", "xu", "yu", "zu", second} <= set(names):
            return None
        ctx = default_context()
        dev = torch.device("cuda", ctx.device)
        seg = molecule_layout(num_mols, num_atoms_per_mol) if msd_type == "com" else None
        times, blocks = [], []
        ids = atom_mass = seg_mass = None
        err = None
        try:
            stream = S.FrameStream(pattern, files=mine, columns=("id", second, "xu", "yu", "zu"),
                                   batch_bytes=STREAM_BATCH_BYTES)
            for batch in stream:
                B, _, n = batch.xyz.shape
                if ids is None:
                    ids = batch.ids[0].copy()
                if msd_type == "com":
                    if seg[0][-1] != n:
                        raise ValueError(f"Length of values ({int(seg[0][-1])}) does not match length of index ({n})")
                    m = batch.types if not mass else np.asarray(mass, dtype=np.float64)[batch.types.astype(np.int64) - 1]
                    if atom_mass is None:
                        atom_mass = m[0].copy()
                    if not (m == atom_mass).all():
                        raise ValueError("atom masses change between frames")
                    out = torch.empty((B, 3, len(seg[0]) - 1), dtype=torch.float64, device=dev)
                    _, seg_mass, _ = backend.segment_com(batch.xyz, atom_mass, seg[0], out=out, ctx=ctx)
                else:
                    out = torch.empty((B, 3, n), dtype=torch.float64, device=dev)
                    out.copy_(torch.from_numpy(batch.xyz))  # synchronous DMA from the page-locked batch
                blocks.append(out)
                times.extend(batch.timesteps.tolist())
        except Exception as e:  # noqa: BLE001
            if files is None:
                raise
            err = e  # (this rank's own parse / reduce error: agreed on below, before the first collective)
        if files is not None:
            D.raise_together(err, "streaming its dump files")
        times = np.asarray(times, dtype=np.float64) * self.timestep * constants.TIME_CONVERSION[self.units]
        if files is not None:
            D.require_all_nonempty(len(times), "dump file")  # every rank raises, or none
        elif not blocks:
            return None
        r = torch.cat(blocks) if len(blocks) > 1 else blocks[0]
        del blocks
        sharded = files is not None and D.is_distributed()
        if msd_type == "com":
            return times, r, dict(type=seg[1], mol_id=seg[2], mass=seg_mass), sharded
        return times, r, dict(id=ids.astype(np.int64)), sharded  # an integer column, as the reference's parser reads it

    def _frame_columns(self, pattern, msd_type, mass, native, files=None):
        """Yields (timestep, column names, {name: id-sorted float64 column}) with xu, yu, zu present
        (made from x + ix*L when they were not dumped, diffusion.py:62-81)."""
        from .. import io as mio

        def wanted(names):
            assert "id" in names, "Missing atom id's in dump file."
            sel = ["id"]
            if "zu" in names:
                sel += ["xu", "yu", "zu"]
            else:
                assert "z" in names, "Missing wrapped and unwrapped coordinates (x y z xu yu zu)"
                assert "iz" in names, (
                    "Missing unwrapped coordinates (xu yu zu) and box location (ix iy iz) for converting "
                    "wrapped coordinates (x y z) into unwrapped coordinates. ")
                sel += ["x", "y", "z", "ix", "iy", "iz"]
            if msd_type == "com":
                if not mass:
                    assert "mass" in names, "Missing atom masses in dump file."
                    sel.append("mass")
                else:
                    sel.append("type")
            return sel

        if native:
            for ts, bounds, _lengths, names, planes in mio.iter_native_frames(pattern, wanted, sort_by="id",
                                                                               files=files):
                cols = dict(zip(wanted(names), planes))
                if "zu" not in cols:
                    for k, axis in enumerate("xyz"):
                        cols[axis + "u"] = cols[axis] + cols["i" + axis] * (bounds[k][1] - bounds[k][0])
                yield ts, names, cols
            return
        for dump in parse_lammps_dumps(pattern):
            names = list(dump.data.columns)
            sel = wanted(names)
            dump.data = dump.data.sort_values(by=["id"])
            dump.data.reset_index(inplace=True)
            dump = self._prepare_unwrapped_coords(dump)
            want = set(sel) | {"xu", "yu", "zu"}
            yield dump.timestep, names, {c: dump.data[c].to_numpy(dtype=np.float64) for c in want}

    def get_msd_from_dump(self, filename, msd_type="com", num_mols=None, num_atoms_per_mol=None, mass=None,
                          com_drift=False, avg_interval=False, tao_coeff=4):
        """
        MSD from LAMMPS dump files; the frame at time 0 is the reference of `msd` / `msd_all`.

        msd_type 'allatom' averages over atoms, 'com' over the molecules of each type (needs num_mols,
        num_atoms_per_mol and masses from `mass` or the dump). com_drift removes the per-type
        centre-of-mass drift first. avg_interval additionally returns `msd_int`: per atom / molecule,
        the displacement over windows of `tao_coeff` frames averaged over the trajectory.

        Returns (msd, msd_all) or (msd, msd_all, msd_int) as DataFrames laid out like the reference's.
        """
        times, r, meta, sharded = self._entity_frames(filename, msd_type, num_mols, num_atoms_per_mol, mass)
        if sharded:
            got = self._msd_frame_sharded(times, r, meta, msd_type, com_drift, avg_interval, tao_coeff)
            if got is not None:
                return got
            # frames out of time order ACROSS ranks: the blocks cannot be reduced where they are; gather them (device
            # to device on RCCL) and go on as one process would
            from .. import dist as D

            counts = D.allgather_counts(len(times))
            times, r = D.allgather_var(times, counts), D.allgather_var(r, counts)
        order = np.argsort(times, kind="stable")  # the reference sorts its (time, id) index
        if not np.array_equal(order, np.arange(len(order))):
            times, r = times[order], _take_frames(r, order)
        F, _, E = r.shape
        id_cols, id_vals, group_off, group_labels = self._groups(meta, msd_type, E)
        dist = constants.DISTANCE_CONVERSION[self.units]
        scale = dist
        if msd_type == "com" and com_drift:
            if _is_device(r):  # small ([F,3,M]) and host arithmetic: same doubles as the load-everything route
                r = r.cpu().numpy()
            r = self._remove_drift(r * dist, meta["mass"] * constants.MASS_CONVERSION[self.units], group_off)
            scale = 1.0
        origin = np.flatnonzero(times == 0)
        if len(origin) == 0:
            raise KeyError(0)  # the reference selects the time-0 rows with .xs(0, 0)
        origin = int(origin[0])
        pairs = np.column_stack([np.full(F, origin), np.arange(F)]).astype(np.int32)
        # the per-entity values come back as the four COLUMNS of msd_all (one contiguous block each); the frame wraps
        # them, the time column and the tiled id columns without consolidating them into a second copy
        col_block = np.empty((4, F * E))
        sums = backend.msd_pairs_cols(r, pairs, group_off, col_block, scale=scale)
        msd_all = self._msd_all_frame(times, E, id_cols, id_vals, col_block)
        msd = self._msd_frame(times, sums, group_off, group_labels, msd_type)
        if not avg_interval:
            return msd, msd_all

        n_kept = len(np.arange(F)[::tao_coeff])  # diffusion.py:226-228: every tao-th time is kept
        win = backend.msd_windows(r, tao_coeff, scale=scale)  # (the kernel strides over the kept frames in place)
        return msd, msd_all, self._msd_int_frame(win, n_kept, E, id_cols, id_vals)

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _groups(meta, msd_type, E):
        if msd_type == "allatom":
            return ["id"], [meta["id"]], np.array([0, E], dtype=np.int64), None
        counts = np.bincount(meta["type"])[1:]
        group_off = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
        return ["type", "mol_id"], [meta["type"], meta["mol_id"]], group_off, np.arange(1, len(counts) + 1)

    @staticmethod
    def _msd_all_frame(times, E, id_cols, id_vals, col_block):
        all_cols = {"Time (s)": np.repeat(times, E)}
        for name, v in zip(id_cols, id_vals):
            all_cols[name] = np.tile(v, len(times))
        for k, name in enumerate(_DISPS + ["msd"]):
            all_cols[name] = col_block[k]
        return pd.DataFrame(all_cols, copy=False)

    @staticmethod
    def _msd_frame(times, sums, group_off, group_labels, msd_type):
        cols_1d = _DISPS + ["msd"]
        means = sums / np.diff(group_off)[None, :, None]
        if msd_type == "allatom":
            return pd.DataFrame({"Time (s)": times, **{c: means[:, 0, k] for k, c in enumerate(cols_1d)}})
        data = {"Time (s)": times}
        for g, lab in enumerate(group_labels):  # diffusion.py:220-222: dx21 dy21 dz21 msd1 dx22 ...
            for k, c in enumerate(cols_1d):
                data[f"{c}{lab}"] = means[:, g, k]
        return pd.DataFrame(data)

    @staticmethod
    def _msd_int_frame(win, n_kept, E, id_cols, id_vals):
        int_cols = {name: v for name, v in zip(id_cols, id_vals)}
        with np.errstate(invalid="ignore", divide="ignore"):
            for k, c in enumerate(_DISPS):
                int_cols[c] = win[:, k] / (n_kept - 1) if n_kept > 1 else np.full(E, np.nan)
        # the first kept frame has no predecessor: its NaN row sums to 0 and still counts in the mean
        int_cols["msd"] = win[:, 3] / n_kept
        return pd.DataFrame(int_cols)

    def _msd_frame_sharded(self, times_l, r_l, meta, msd_type, com_drift, avg_interval, tao_coeff):
        """
        get_msd_from_dump with the FRAMES where their ranks parsed them (torch.distributed): the frame
        at time 0 is broadcast by its owner (24 E bytes), every rank reduces its own frames against it
        (mdhip_msd_origin) and the [F_local, G, 4] sums are all-gathered; the fixed-lag windows need one frame from
        the rank below (dist.msd_windows_sharded) and an all-reduce of [E, 4]. The trajectory itself never moves.
        `msd_all` is F x E rows whatever is done: its four value columns are gathered (device to device on RCCL) so
        that every rank returns the frame one process would — MSD_ALL_ON_EVERY_RANK = False makes every rank return
        the rows of ITS OWN frames instead and skips that gather.
        Returns None when the ranks' blocks are not in time order one after the other (the caller then gathers the
        frames and reduces them as one process).
        """
        from .. import dist as D

        rank, world = D.rank_world()
        counts = D.allgather_counts(len(times_l))
        times = D.allgather_var(np.asarray(times_l, dtype=np.float64), counts)
        if np.any(np.diff(times) < 0):
            return None
        F, F_l, E = int(sum(counts)), int(r_l.shape[0]), int(r_l.shape[2])
        lo = int(sum(counts[:rank]))
        id_cols, id_vals, group_off, group_labels = self._groups(meta, msd_type, E)
        dist = constants.DISTANCE_CONVERSION[self.units]
        scale = dist
        if msd_type == "com" and com_drift:
            # per-type drift: this rank's frames against the per-type centre of the FIRST frame (rank 0's first)
            if _is_device(r_l):
                r_l = r_l.cpu().numpy()
            ent_mass = meta["mass"] * constants.MASS_CONVERSION[self.units]
            r_si = r_l * dist
            com_l = self._type_com(r_si, ent_mass, group_off)  # [F_l, 3, G]
            com0 = D.broadcast_array(com_l[0] if rank == 0 else None, 0, (3, len(group_off) - 1))
            r_l = self._subtract_drift(r_si, com_l - com0[None], group_off)
            scale = 1.0
        origin = np.flatnonzero(times == 0)
        if len(origin) == 0:
            raise KeyError(0)  # the reference selects the time-0 rows with .xs(0, 0)
        origin = int(origin[0])
        on_dev = _is_device(r_l)
        if on_dev:
            import torch

            cols_l = torch.empty((4, F_l * E), dtype=torch.float64, device=r_l.device)
        else:
            cols_l = np.empty((4, F_l * E))
        sums = D.msd_single_origin_sharded(r_l, F, group_off, scale=scale, origin_frame=origin, counts=counts,
                                           cols=cols_l)
        if MSD_ALL_ON_EVERY_RANK:
            col_block = np.empty((4, F * E))
            for k in range(4):  # column k of every rank's frames, in rank (= time) order
                g = D.allgather_var(cols_l[k].reshape(F_l, E), counts)
                col_block[k] = (g.cpu().numpy() if on_dev else g).reshape(-1)
            msd_all = self._msd_all_frame(times, E, id_cols, id_vals, col_block)
        else:
            col_block = cols_l.cpu().numpy() if on_dev else cols_l
            msd_all = self._msd_all_frame(times[lo:lo + F_l], E, id_cols, id_vals, col_block)
        del cols_l
        msd = self._msd_frame(times, sums, group_off, group_labels, msd_type)
        if not avg_interval:
            return msd, msd_all
        n_kept = len(np.arange(F)[::tao_coeff])
        win = D.msd_windows_sharded(r_l, F, tao_coeff, scale=scale, counts=counts)
        return msd, msd_all, self._msd_int_frame(win, n_kept, E, id_cols, id_vals)

    @staticmethod
    def _type_com(r_si, ent_mass, group_off):
        """Per-type mass-weighted centre [F, 3, G] of r_si [F, 3, E] (diffusion.py:83-89)."""
        out = np.empty(r_si.shape[:2] + (len(group_off) - 1,))
        for g in range(len(group_off) - 1):
            lo, hi = group_off[g], group_off[g + 1]
            m = ent_mass[lo:hi]
            out[:, :, g] = (r_si[:, :, lo:hi] @ m) / m.sum()
        return out

    @staticmethod
    def _subtract_drift(r_si, drift, group_off):
        out = r_si.copy()
        for g in range(len(group_off) - 1):
            out[:, :, group_off[g]:group_off[g + 1]] -= drift[:, :, g][:, :, None]
        return out

    @staticmethod
    def _remove_drift(r_si, ent_mass, group_off):
        """Per-type mass-weighted COM minus the same at the first frame, subtracted from every
        molecule of that type (diffusion.py:83-96). r_si [F,3,E]."""
        out = r_si.copy()
        for g in range(len(group_off) - 1):
            lo, hi = group_off[g], group_off[g + 1]
            m = ent_mass[lo:hi]
            com = (r_si[:, :, lo:hi] @ m) / m.sum()  # [F,3]
            out[:, :, lo:hi] -= (com - com[0])[:, :, None]
        return out

    # ------------------------------------------------------------------------------------------
    def get_msd_from_log(self, log_pattern):
        """msd columns of LAMMPS log file(s) in m^2 plus 'Time (s)' (diffusion.py:241-265)."""
        full_log = concat_log(log_pattern, step=None, working_dir=self.outputs_dir)
        msd = full_log.filter(regex="msd").copy()
        for col in msd:
            msd[col] = msd[col] * constants.DISTANCE_CONVERSION[self.units] ** 2
        msd["Time (s)"] = full_log["Step"] * self.timestep * constants.TIME_CONVERSION[self.units]
        return msd

    def calc_diff(self, msd, initial_time=None, final_time=None, dimension=3, diff_names=None, save=False,
                  plot=False):
        """
        D = slope / (2 * dimension) of a through-origin least-squares fit of every column whose name
        contains 'msd' against 'Time (s)', with its standard error and R^2 (diffusion.py:267-408).
        initial_time / final_time: dicts {column position: seconds} restricting the fit range.
        Writes diffusion.csv (and diff_<name>.txt when save, msd.png / msd_log.png when plot).
        """
        initial_time = initial_time or {}
        final_time = final_time or {}
        t_all = msd["Time (s)"]
        t_min, t_max = min(t_all), max(t_all)
        names = [c for c in msd.columns if "msd" in c.lower()]
        table = np.zeros((len(names), 3))
        fits = []
        for k, col in enumerate(names):
            sel = (t_all >= initial_time.get(k, t_min)) & (t_all <= final_time.get(k, t_max))
            fit = _OlsThroughOrigin(msd.loc[sel, col], t_all[sel])
            fits.append((fit, sel))
            table[k] = [fit.slope / (2 * dimension), fit.bse / (2 * dimension), fit.rsquared]
            if save:
                tag = diff_names[k] if diff_names else k + 1
                with open(f"{self.diff_dir}/diff_{tag}.txt" if _writer() else os.devnull, "w") as fh:
                    fh.write(str(fit.summary()))
        index = diff_names or [k + 1 for k in range(len(names))]
        diffusion = pd.DataFrame(table, columns=["diffusion (m2/s)", "std", "R2"], index=index)
        if plot:
            self._plot_msd(msd, names, fits, index)
        if _writer():
            diffusion.to_csv(f"{self.diff_dir}/diffusion.csv")
            print("Diffusion results written to a .csv file.")
        return diffusion

    def _plot_msd(self, msd, names, fits, labels):
        import matplotlib

        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt

        from ..utilities.plots import set_axis

        t_ns = msd["Time (s)"].to_numpy() * 10 ** 9
        nrows = int(np.ceil(len(names) / 2))
        for fname, logscale in (("msd.png", False), ("msd_log.png", True)):
            fig, axes = plt.subplots(nrows, 2, figsize=(12, 8), squeeze=False)
            cmap = plt.get_cmap("Paired")
            for k, (ax, col) in enumerate(zip(axes.flatten(), names)):
                fit, sel = fits[k]
                ax.plot(t_ns, msd[col], color=cmap(k / 10), linewidth=2, label=labels[k])
                if logscale:
                    ref = 10 ** (np.log10(msd[col].max()) - np.log10(t_ns.max()))
                    ax.plot(t_ns, t_ns * ref, color="k", ls="--", linewidth=2)
                    ax.set(xscale="log", yscale="log")
                else:
                    ax.plot(t_ns[np.asarray(sel)], fit.predict(), color="k", ls="--", linewidth=2)
                set_axis(ax, axis="both")
                ax.legend(fontsize=16, frameon=False)
                ax.set_xlabel(r"$\mathrm{Time, 10^9 (s)}$", fontsize=18)
                ax.set_ylabel(r"$\mathrm{MSD\ (m^2)}$", fontsize=18)
            for ax in axes.flatten()[len(names):]:
                fig.delaxes(ax)
            fig.tight_layout()
            fig.savefig(f"{self.diff_dir}/{fname}", bbox_inches="tight", pad_inches=0.1)
            plt.close(fig)

    def get_diff_dist(self, msd_int, dump_freq, dimension=3, tao_coeff=4, plot=False, diff_names=None):
        """
        Per-entity diffusion coefficients from `msd_int`: msd / (2 * dimension * tao_coeff * delta),
        delta = dump_freq * timestep in seconds (diffusion.py:410-516). Adds the column 'diff' and
        returns the DataFrame; with plot=True also saves a histogram to diff_dist.png.
        """
        delta = dump_freq * self.timestep * constants.TIME_CONVERSION[self.units]
        msd_int["diff"] = msd_int["msd"] / (2 * dimension * tao_coeff * delta)
        if plot:
            import matplotlib

            matplotlib.use("Agg", force=False)
            import matplotlib.pyplot as plt

            from ..utilities.plots import set_axis

            if "type" in msd_int.columns:
                groups = list(msd_int.groupby("type"))
                labels = diff_names or [k + 1 for k in range(len(groups))]
                fig, axes = plt.subplots(int(np.ceil(len(groups) / 2)), 2, figsize=(12, 8), squeeze=False)
                for ax, (key, grp) in zip(axes.flatten(), groups):
                    ax.hist(grp["diff"] * 10 ** 9, bins="sqrt", density=True, edgecolor="k", label=labels[key - 1])
                    ax.legend(fontsize=16, frameon=False)
                for ax in axes.flatten()[len(groups):]:
                    fig.delaxes(ax)
                used = axes.flatten()[: len(groups)]
            else:
                fig, ax = plt.subplots(figsize=(8, 6))
                ax.hist(msd_int["diff"] * 10 ** 9, bins="sqrt", density=True, edgecolor="k")
                used = [ax]
            for ax in used:
                set_axis(ax, axis="both")
                ax.set_xlabel(r"$\mathrm{Diffusivity, 10^{-9}\ (m^2/s)}$", fontsize=18)
                ax.set_ylabel("Frequency", fontsize=18)
            fig.tight_layout()
            fig.savefig(f"{self.diff_dir}/diff_dist.png", bbox_inches="tight", pad_inches=0.1)
            plt.close(fig)
        return msd_int
