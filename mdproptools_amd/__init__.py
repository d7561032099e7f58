"""
mdproptools_amd — MI355X-native backend for the RDF/CN and MSD/Green-Kubo hot
path of molmd/mdproptools, behind the reference's own function/class surface.

    from mdproptools_amd.structural.rdf_cn import calc_atomic_rdf
    from mdproptools_amd.dynamical.diffusion import Diffusion

The numerics live in `libmdhip.so` (hand-written HIP for gfx950 behind the
C-ABI declared in include/mdhip.h); importing the package does not load it,
calling any hot-path function does and fails loudly when it is missing.
"""

__version__ = "0.1.0"
