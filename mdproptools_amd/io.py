"""
LAMMPS text readers for the hot path: dump frames and thermo logs.

The reference takes these from a third-party package that is not vendored
(`pymatgen.io.lammps.outputs.parse_lammps_dumps` / `parse_lammps_log`; call
sites /root/reference/mdproptools/structural/rdf_cn.py:176,260,
dynamical/diffusion.py:75-77,172, dynamical/conductivity.py:87,
dynamical/viscosity.py:211, utilities/log.py:21). No arithmetic of the hot path
lives there, only parsing, so this module restates the observable behaviour the
call sites rely on:

* files are matched with glob and, when the pattern has a ``*``, ordered by the
  integer the ``*`` stands for;
* a frame starts at ``ITEM: TIMESTEP``; it carries ``timestep`` (int),
  ``natoms`` (int), ``box`` (bounds + optional tilt) and ``data`` (a
  ``pandas.DataFrame`` whose columns are the names after ``ITEM: ATOMS``,
  parsed by pandas' whitespace reader so every float is the same correctly
  rounded double the reference sees);
* ``box.bounds[k] = [lo, hi]`` and ``box.to_lattice().lengths`` gives the edge
  lengths (``hi - lo`` for an orthogonal box);
* a log yields one DataFrame per ``run`` (the thermo block between the memory
  usage line and ``Loop time of``).

``read_dump_arrays`` is the fast path used by the drop-in layer when it only
needs SoA float64 planes (it skips the DataFrame entirely).
"""

import glob
import io as _io
import os
import re

import numpy as np
import pandas as pd


USE_NATIVE_READER = True  # drop-in functions read dumps through libmdhip.so's reader (same doubles as pandas)


class _Lattice:
    """Minimal stand-in for the lattice object `box.to_lattice()` returns."""

    def __init__(self, matrix):
        self.matrix = np.asarray(matrix, dtype=np.float64)

    @property
    def lengths(self):
        # |row| of the cell matrix; for a diagonal matrix sqrt(a*a) == |a| exactly.
        return tuple(np.sqrt(np.sum(self.matrix ** 2, axis=1)).tolist())

    @property
    def volume(self):
        return float(abs(np.linalg.det(self.matrix)))


class LammpsBox:
    def __init__(self, bounds, tilt=None):
        self.bounds = [list(map(float, b)) for b in bounds]
        self.tilt = None if tilt is None else [float(t) for t in tilt]

    def to_lattice(self):
        (xlo, xhi), (ylo, yhi), (zlo, zhi) = self.bounds
        xy, xz, yz = self.tilt if self.tilt is not None else (0.0, 0.0, 0.0)
        return _Lattice(
            [[xhi - xlo, 0.0, 0.0], [xy, yhi - ylo, 0.0], [xz, yz, zhi - zlo]]
        )

    @property
    def volume(self):
        m = self.to_lattice().matrix
        return float(m[0, 0] * m[1, 1] * m[2, 2])


class LammpsDump:
    def __init__(self, timestep, natoms, box, data):
        self.timestep = timestep
        self.natoms = natoms
        self.box = box
        self.data = data

    @classmethod
    def from_lines(cls, lines):
        timestep = int(lines[1])
        natoms = int(lines[3])
        header = lines[4].split()
        rows = [lines[5].split(), lines[6].split(), lines[7].split()]
        triclinic = "xy" in header
        bounds = np.array([[float(v) for v in r[:2]] for r in rows])
        tilt = None
        if triclinic:
            tilt = [float(r[2]) for r in rows]
            xy, xz, yz = tilt
            # LAMMPS writes the bounding box of a tilted cell; undo that.
            bounds[0, 0] -= min(0.0, xy, xz, xy + xz)
            bounds[0, 1] -= max(0.0, xy, xz, xy + xz)
            bounds[1, 0] -= min(0.0, yz)
            bounds[1, 1] -= max(0.0, yz)
        box = LammpsBox(bounds.tolist(), tilt)
        columns = lines[8].replace("ITEM: ATOMS", "").split()
        body = "\n".join(lines[9:])
        data = pd.read_csv(_io.StringIO(body), names=columns, sep=r"\s+")
        return cls(timestep, natoms, box, data)


def _sorted_matches(file_pattern):
    files = glob.glob(file_pattern)
    if len(files) > 1 and "*" in file_pattern:
        rx = re.compile(
            ".*" + re.escape(file_pattern).replace(r"\*", "([0-9]+)")
        )

        def key(f):
            m = rx.match(f)
            return int(m.group(1)) if m else 0

        files = sorted(files, key=key)
    return files


def _open_text(fname):
    """Plain or gzip-compressed text (pymatgen's zopen accepts both; the mmap reader takes plain text only)."""
    if str(fname).endswith(".gz"):
        import gzip

        return gzip.open(fname, "rt")
    return open(fname, "rt")


def _iter_frames(fname):
    frame = []
    with _open_text(fname) as fh:
        for line in fh:
            if line.startswith("ITEM: TIMESTEP"):
                if frame:
                    yield frame
                frame = [line.rstrip("\n")]
            elif frame:
                frame.append(line.rstrip("\n"))
    if frame:
        yield frame


def parse_lammps_dumps(file_pattern):
    """Generator of `LammpsDump`, one per frame, files in numeric order."""
    for fname in _sorted_matches(file_pattern):
        for frame in _iter_frames(fname):
            yield LammpsDump.from_lines(frame)


_LOG_BEGIN = ("Memory usage per processor =", "Per MPI rank memory allocation")
_LOG_END = "Loop time of"


def _parse_lammps_log_text(filename):
    """The reference's route: lines -> blocks -> pandas.read_csv (whitespace separated)."""
    with open(filename, "rt") as fh:
        lines = fh.readlines()
    runs = []
    start = None
    for i, line in enumerate(lines):
        if line.startswith(_LOG_BEGIN):
            start = i + 1
        elif line.startswith(_LOG_END) and start is not None:
            runs.append((start, i))
            start = None
    frames = []
    for lo, hi in runs:
        block = [ln for ln in lines[lo:hi] if not ln.startswith("WARNING")]
        if not block:
            continue
        df = pd.read_csv(_io.StringIO("".join(block)), sep=r"\s+")
        frames.append(df)
    return frames


def _parse_lammps_log_native(filename):
    """Thermo tables through the native reader of libmdhip.so (mmap, threaded number parsing, host only);
    None when a table is not plain numbers (the text route then reproduces pandas' behaviour exactly)."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    h = C.c_void_p()
    if lib.mdhip_log_open(str(filename).encode(), C.byref(h)) != 0:
        raise OSError("cannot read log %s: %s" % (filename, (lib.mdhip_log_error(None) or b"").decode()))
    try:
        frames = []
        for run in range(int(lib.mdhip_log_n_runs(h))):
            n_rows, n_cols, regular = C.c_int64(), C.c_int(), C.c_int()
            buf = C.create_string_buffer(1 << 16)
            if lib.mdhip_log_run_info(h, run, C.byref(n_rows), C.byref(n_cols), C.byref(regular), buf, 1 << 16) != 0:
                return None
            names = buf.value.decode().split()
            if not regular.value or len(set(names)) != len(names) or n_rows.value == 0:
                return None
            planes = np.empty((n_cols.value, n_rows.value), dtype=np.float64)
            is_int = np.zeros(n_cols.value, dtype=np.int32)
            rc = lib.mdhip_log_read(h, run, planes.ctypes.data_as(C.POINTER(C.c_double)),
                                    is_int.ctypes.data_as(C.POINTER(C.c_int32)), min(16, os.cpu_count() or 1))
            if rc != 0:
                return None
            cols = {}
            for c, name in enumerate(names):
                col = planes[c]
                # pandas infers int64 for a column of plain integers (e.g. Step)
                cols[name] = col.astype(np.int64) if is_int[c] and np.all(np.abs(col) < 2 ** 62) else col
            frames.append(pd.DataFrame(cols))
        return frames
    finally:
        lib.mdhip_log_close(h)


def parse_lammps_log(filename="log.lammps"):
    """List of thermo DataFrames, one per `run` found in a LAMMPS log."""
    if USE_NATIVE_READER:
        frames = _parse_lammps_log_native(filename)
        if frames is not None:
            return frames
    return _parse_lammps_log_text(filename)


class NativeDumpFile:
    """One dump file opened by the native reader of libmdhip.so (mmap + frame index, host only)."""

    def __init__(self, path):
        import ctypes as C

        from . import _lib

        self._C, self._lib = C, _lib.load()
        h = C.c_void_p()
        rc = self._lib.mdhip_dump_open(str(path).encode(), C.byref(h))
        if rc != 0:
            raise OSError("cannot read dump %s: %s" % (path, (self._lib.mdhip_dump_error(None) or b"").decode()))
        self._h = h
        self.n_frames = int(self._lib.mdhip_dump_n_frames(h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mdhip_dump_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def header(self, f):
        """(timestep, natoms, bounds [3,2] with the tilt correction applied, tilt or None, columns)."""
        C = self._C
        ts, na, tri, nc = C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        b6, t3 = (C.c_double * 6)(), (C.c_double * 3)()
        buf = C.create_string_buffer(4096)
        rc = self._lib.mdhip_dump_frame_info(self._h, f, C.byref(ts), C.byref(na), b6, t3, C.byref(tri),
                                             C.byref(nc), buf, 4096)
        if rc != 0:
            raise IndexError("frame %d" % f)
        bounds = np.array(list(b6)).reshape(3, 2)
        tilt = None
        if tri.value:
            xy, xz, yz = tilt = list(t3)
            bounds[0, 0] -= min(0.0, xy, xz, xy + xz)
            bounds[0, 1] -= max(0.0, xy, xz, xy + xz)
            bounds[1, 0] -= min(0.0, yz)
            bounds[1, 1] -= max(0.0, yz)
        return int(ts.value), int(na.value), bounds, tilt, buf.value.decode().split()

    def read(self, f, columns, sort_by=None, n_threads=0):
        """Columns (names) of frame f as planes [len(columns), natoms], rows ordered by `sort_by`."""
        C = self._C
        ts, na, bounds, tilt, names = self.header(f)
        idx = np.array([names.index(c) for c in columns], dtype=np.int32)
        out = np.empty((len(columns), na), dtype=np.float64)
        sort_col = names.index(sort_by) if sort_by is not None else -1
        rc = self._lib.mdhip_dump_read(self._h, f, len(idx), idx.ctypes.data_as(C.POINTER(C.c_int32)), sort_col,
                                       out.ctypes.data_as(C.POINTER(C.c_double)),
                                       int(n_threads) or min(16, os.cpu_count() or 1))
        if rc != 0:
            raise ValueError((self._lib.mdhip_dump_error(self._h) or b"").decode())
        return out


def _pandas_file_frames(fname, columns, sort_by):
    """The frames of one file through the pandas route, in the tuple form of the native route (compressed files).
    Rows the native reader refuses — short, blank, or a non-numeric token in a REQUESTED column — are an error there,
    not a reason to come here; text in columns nobody asked for (`element`) is skipped by the native reader."""
    out = []
    for frame in _iter_frames(fname):
        d = LammpsDump.from_lines(frame)
        names = list(d.data.columns)
        want = columns(names) if callable(columns) else columns
        df = d.data
        if sort_by is not None:
            df = df.sort_values(sort_by)  # KeyError when the column is absent, as in the reference (rdf_cn.py:192)
        planes = np.ascontiguousarray(df[list(want)].to_numpy(dtype=np.float64).T)
        out.append((d.timestep, np.asarray(d.box.bounds, dtype=np.float64), d.box.to_lattice().lengths, names, planes))
    return out


def native_read_into(nd, f, names, columns, dests, sort_by="id", n_threads=1):
    """Columns `columns` of frame f of the open NativeDumpFile `nd`, parsed straight into the float64 arrays `dests`
    (one [natoms] array per column, each contiguous; they may be slices of a page-locked staging buffer)."""
    C = nd._C
    if sort_by is not None and sort_by not in names:
        raise KeyError(sort_by)
    idx = np.array([names.index(c) for c in columns], dtype=np.int32)
    ptrs = (C.POINTER(C.c_double) * len(dests))(*[d.ctypes.data_as(C.POINTER(C.c_double)) for d in dests])
    rc = nd._lib.mdhip_dump_read_cols(nd._h, f, len(idx), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                      names.index(sort_by) if sort_by is not None else -1, ptrs, int(n_threads))
    if rc != 0:
        raise ValueError((nd._lib.mdhip_dump_error(nd._h) or b"").decode())


def _native_file_frames(fname, columns, sort_by, n_threads):
    if str(fname).endswith(".gz"):
        return _pandas_file_frames(fname, columns, sort_by)
    nd = NativeDumpFile(fname)
    try:
        out = []
        for f in range(nd.n_frames):
            ts, na, bounds, tilt, names = nd.header(f)
            want = columns(names) if callable(columns) else columns
            if sort_by is not None and sort_by not in names:
                raise KeyError(sort_by)  # pandas' sort_values on a missing column, as in the reference (rdf_cn.py:192)
            planes = nd.read(f, want, sort_by=sort_by, n_threads=n_threads)
            lengths = LammpsBox(bounds.tolist(), tilt).to_lattice().lengths
            out.append((ts, bounds, lengths, names, planes))
        return out
    finally:
        nd.close()


def iter_native_frames(file_pattern, columns, sort_by="id", n_threads=0, workers=None, files=None):
    """
    Frames of every file matching `file_pattern` (same ordering rule as parse_lammps_dumps) through the
    native reader: yields (timestep, bounds [3,2], box lengths (lx,ly,lz), columns present, planes
    [len(columns), natoms]). `columns` is a list of names or a callable (names present -> list of names).
    A ValueError is raised for a requested column the frame does not have.
    Several files are parsed concurrently by `workers` host threads (the C calls release the GIL); the
    frames are still yielded in file order.
    """
    if files is None:  # an explicit list (a rank's share of the files) takes the place of the pattern
        files = _sorted_matches(file_pattern)
    if workers is None:
        workers = min(len(files), max(1, (os.cpu_count() or 1) // 2), 32)
    if workers <= 1 or len(files) <= 1:
        for fname in files:
            yield from _native_file_frames(fname, columns, sort_by, n_threads)
        return
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=workers) as pool:
        for frames in pool.map(lambda fn: _native_file_frames(fn, columns, sort_by, 1), files):
            yield from frames


def read_dump_arrays(file_pattern, columns, sort_by_id=True):
    """
    Read every frame matching `file_pattern` into SoA float64 planes.

    Returns (timesteps int64[F], bounds float64[F,3,2], planes float64[F,C,N])
    with atoms ordered by ascending id in every frame (what the reference gets
    from `sort_values("id")`, rdf_cn.py:192, diffusion.py:176). Values go
    through pandas' reader so they are bit-identical to `parse_lammps_dumps`.
    """
    steps, bounds, planes = [], [], []
    for dump in parse_lammps_dumps(file_pattern):
        df = dump.data
        if sort_by_id:
            df = df.sort_values("id")
        planes.append(
            np.ascontiguousarray(df[list(columns)].to_numpy(dtype=np.float64).T)
        )
        steps.append(dump.timestep)
        bounds.append(dump.box.bounds)
    return (
        np.asarray(steps, dtype=np.int64),
        np.asarray(bounds, dtype=np.float64),
        np.stack(planes) if planes else np.zeros((0, len(columns), 0)),
    )


def write_dump(path, timestep, bounds, columns, table, fmt=None):
    """
    Write one LAMMPS-style text frame (used by tests and the synthetic generator).

    With fmt=None every float is written as its shortest round-trip decimal
    (`repr`), so a table that was parsed from a LAMMPS dump with <= 15
    significant digits per field is written back as the same digits and parses
    to the same doubles again (pandas' default float reader is only correctly
    rounded for short mantissas; longer ones can come back 1 ulp off).
    """
    table = np.asarray(table)
    with open(path, "wt") as fh:
        fh.write("ITEM: TIMESTEP\n%d\n" % int(timestep))
        fh.write("ITEM: NUMBER OF ATOMS\n%d\n" % table.shape[0])
        fh.write("ITEM: BOX BOUNDS pp pp pp\n")
        for lo, hi in bounds:
            fh.write("%.16e %.16e\n" % (lo, hi))
        fh.write("ITEM: ATOMS " + " ".join(columns) + " \n")
        int_cols = {"id", "mol", "type", "ix", "iy", "iz"}
        for row in table:
            fh.write(
                " ".join(
                    ("%d" % int(v)) if c in int_cols else (repr(float(v)) if fmt is None else fmt % v)
                    for c, v in zip(columns, row)
                )
                + " \n"
            )


def write_log(path, table, columns, float_fmt=None):
    """Write a minimal LAMMPS log with one thermo block (used by tests and golden generation)."""
    table = np.asarray(table)
    with open(path, "wt") as fh:
        fh.write("LAMMPS (synthetic)\nunits real\nrun %d\n" % max(0, len(table) - 1))
        fh.write("Per MPI rank memory allocation (min/avg/max) = 1.0 | 1.0 | 1.0 Mbytes\n")
        fh.write(" ".join(columns) + " \n")
        for row in table:
            fh.write(" ".join(("%d" % int(v)) if c == "Step" else
                              (repr(float(v)) if float_fmt is None else float_fmt % v)
                              for c, v in zip(columns, row)) + " \n")
        fh.write("Loop time of 1.0 on 1 procs for %d steps with 1 atoms\n" % max(0, len(table) - 1))
