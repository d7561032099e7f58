"""Placeholder so out-of-scope reference modules import (oracle/shims/README.md)."""


class Molecule:  # pragma: no cover
    def __init__(self, *a, **k):
        raise NotImplementedError("pymatgen is not installed; shim only")


class IMolecule(Molecule):  # pragma: no cover
    pass
