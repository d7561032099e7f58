"""Stand-in for pymatgen.io.lammps.outputs (oracle/shims/README.md): parsing only, written from scratch.

INDEPENDENT of the product: this file does not import mdproptools_amd. The reference, when oracle/make_golden.py
runs it, therefore reads its dump and log files through plain `pandas.read_csv` here, and the script asserts that the
product's own readers (mdproptools_amd.io, native and pandas routes) return the same arrays for the same files — so
the golden inputs are not "parsed by the code under test".

What the reference's call sites use (structural/rdf_cn.py:176,190,191,260; dynamical/diffusion.py:66-77,172;
dynamical/residence_time.py:54,78-80; utilities/log.py:21): `parse_lammps_dumps(pattern)` yields objects with
`.timestep`, `.natoms`, `.box.bounds`, `.box.to_lattice().lengths`, `.data` (DataFrame, columns as after ITEM: ATOMS);
files matching a `*` pattern come in the numeric order of what the `*` stands for. `parse_lammps_log(file)` returns one
DataFrame per run: the thermo table between the memory-usage line and "Loop time of".
"""
import glob
import io
import re

import numpy as np
import pandas as pd


class _Lattice:
    def __init__(self, lengths):
        self.lengths = tuple(float(v) for v in lengths)


class LammpsBox:
    def __init__(self, bounds, tilt=None):
        self.bounds = [[float(lo), float(hi)] for lo, hi in bounds]
        self.tilt = tilt

    def to_lattice(self):
        # orthogonal cells only (all the reference's data): |row| of diag(hi - lo) is hi - lo itself
        return _Lattice([np.sqrt((hi - lo) ** 2) for lo, hi in self.bounds])


class LammpsDump:
    def __init__(self, timestep, natoms, box, data):
        self.timestep, self.natoms, self.box, self.data = timestep, natoms, box, data


def _frames_of(path):
    with open(path) as fh:
        lines = fh.read().split("\n")
    starts = [k for k, ln in enumerate(lines) if ln.startswith("ITEM: TIMESTEP")]
    for a, b in zip(starts, starts[1:] + [len(lines)]):
        blk = lines[a:b]
        timestep = int(blk[1])
        natoms = int(blk[3])
        bounds = [[float(v) for v in blk[5 + k].split()[:2]] for k in range(3)]
        cols = blk[8].split()[2:]
        text = "\n".join(blk[9:9 + natoms])
        data = pd.read_csv(io.StringIO(text), sep=r"\s+", header=None, names=cols)
        yield LammpsDump(timestep, natoms, LammpsBox(bounds), data)


def parse_lammps_dumps(file_pattern):
    files = glob.glob(file_pattern)
    if len(files) > 1:
        pat = re.escape(file_pattern).replace(r"\*", "([0-9]+)")
        files.sort(key=lambda f: int(re.match(pat, f).group(1)))
    for f in files:
        yield from _frames_of(f)


def parse_lammps_log(filename="log.lammps"):
    with open(filename) as fh:
        lines = fh.read().split("\n")
    begin = [k for k, ln in enumerate(lines)
             if ln.startswith("Memory usage per processor =") or ln.startswith("Per MPI rank memory allocation")]
    end = [k for k, ln in enumerate(lines) if ln.startswith("Loop time of")]
    runs = []
    for a, b in zip(begin, end):
        body = [ln for ln in lines[a + 1:b] if ln.strip() and not ln.startswith("WARNING")]
        runs.append(pd.read_csv(io.StringIO("\n".join(body)), sep=r"\s+"))
    return runs
