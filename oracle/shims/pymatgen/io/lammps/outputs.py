"""Stand-in for pymatgen.io.lammps.outputs: our own reader (parsing only; oracle/shims/README.md)."""
from mdproptools_amd.io import (  # noqa: F401
    LammpsBox,
    LammpsDump,
    parse_lammps_dumps,
    parse_lammps_log,
)
