"""
One-frame charge flux — /root/reference/mdproptools/dynamical/_conductivity.py:7-36.

The reference maps this function over frames with a process pool; `Conductivity.get_charge_flux`
here sends all frames to the GPU in one call instead. The single-frame form is kept for callers
that use it directly.
"""

import numpy as np

from .. import backend
from ..common import constants
from ..common.com_mols import atom_masses, molecule_layout


def conductivity_loop(dump, num_mols, num_atoms_per_mol, mass, ind, units):
    """-> (time in s for timestep 1, ind, flux [3, n_types]) for one frame."""
    dump.data = dump.data.sort_values(by=["id"])
    dump.data.reset_index(inplace=True)
    seg_off, mol_type, _ = molecule_layout(num_mols, num_atoms_per_mol)
    vel = np.ascontiguousarray(dump.data[["vx", "vy", "vz"]].to_numpy(dtype=np.float64).T)[None]
    flux = backend.charge_flux(vel, atom_masses(dump.data, mass), dump.data["q"].to_numpy(dtype=np.float64),
                               seg_off, (mol_type - 1).astype(np.int32), len(num_mols),
                               constants.VELOCITY_CONVERSION[units], constants.CHARGE_CONVERSION[units])
    return dump.timestep * constants.TIME_CONVERSION[units], ind, flux[:, :, 0]
