"""
Per-molecule centre of mass — drop-in for /root/reference/mdproptools/common/com_mols.py:5-62.

The segmented mass-weighted sums run on the GPU (`mdhip_segment_com`); molecule membership is the
one the reference derives from the sorted-id order (type-major, then molecule, then atom).
"""

import numpy as np
import pandas as pd

from .. import backend


def molecule_layout(num_mols, num_atoms_per_mol):
    """(seg_off int64 [M+1], mol_type int64 [M] 1-based, mol_id int64 [M] 1-based within its type)."""
    nm = np.asarray(num_mols, dtype=np.int64)
    na = np.asarray(num_atoms_per_mol, dtype=np.int64)
    seg_off = np.concatenate(([0], np.cumsum(np.repeat(na, nm)))).astype(np.int64)
    mol_type = np.repeat(np.arange(1, len(nm) + 1), nm)
    mol_id = np.concatenate([np.arange(1, n + 1) for n in nm]) if len(nm) else np.zeros(0, dtype=np.int64)
    return seg_off, mol_type, mol_id


def atom_masses(data, mass):
    """Per-atom masses: from the `mass` list indexed by type, or from the dump's own column."""
    if not mass:
        assert "mass" in data.columns, "Missing atom masses in dump file."
        return data["mass"].to_numpy(dtype=np.float64)
    return np.asarray(mass, dtype=np.float64)[data["type"].to_numpy().astype(np.int64) - 1]


def calc_com(dump, num_mols, num_atoms_per_mol, mass=None, atom_attributes=["xu", "yu", "zu"],
             calc_charge=False):
    """
    Mass-weighted mean of `atom_attributes` per molecule, molecule mass and (optionally) charge.

    Args as in the reference: dump (frame with `.data` sorted by id), num_mols, num_atoms_per_mol,
    mass (list per atom type, or None to use the dump's mass column), atom_attributes, calc_charge.
    Returns a DataFrame indexed by (type, mol_id).
    """
    data = dump.data
    seg_off, mol_type, mol_id = molecule_layout(num_mols, num_atoms_per_mol)
    if seg_off[-1] != len(data):
        raise ValueError(
            f"Length of values ({int(seg_off[-1])}) does not match length of index ({len(data)})")
    m = atom_masses(data, mass)
    attr = np.ascontiguousarray(data[list(atom_attributes)].to_numpy(dtype=np.float64).T)[None]
    q = data["q"].to_numpy(dtype=np.float64) if calc_charge else None
    com, seg_mass, seg_q = backend.segment_com(attr, m, seg_off, atom_q=q)
    cols = {}
    if not mass:  # the dump's mass column sits in front of the attributes (com_mols.py:49)
        cols["mass"] = seg_mass
    for k, name in enumerate(atom_attributes):
        cols[name] = com[0, k]
    if calc_charge:
        cols["q"] = seg_q
    if mass:
        cols["mass"] = seg_mass
    index = pd.MultiIndex.from_arrays([mol_type, mol_id], names=["type", "mol_id"])
    return pd.DataFrame(cols, index=index)
