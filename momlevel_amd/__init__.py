""" momlevel_amd - momlevel's steric sea-level hot path on AMD Instinct MI355X (gfx950)

A drop-in for one path of jkrasting/momlevel: the Wright (1997) in-situ density and
the volume-weighted reductions behind ``steric`` / ``halosteric`` / ``thermosteric``
and ``derived.calc_rho`` / ``calc_masso`` / ``calc_volo``, computed by hand-written
HIP kernels behind a C ABI (include/momlevel_hip.h).  Everything else in momlevel
(trends, tide gauges, vorticity, spiciness, ...) is out of scope -- use momlevel.

There is no CPU fallback: without libmomlevel_hip.so and a HIP device the compute
entry points raise ``MomlevelHipError``.
"""

__version__ = "0.1.0"

from . import derived
from . import dynamic
from . import eos
from . import reference
from . import test_data
from . import util
from ._lib import MomlevelHipError
from .dynamic import inverse_barometer
from .labeled import DataArray, Dataset
from .steric import halosteric, steric, steric_variants, thermosteric

__all__ = [
    "DataArray",
    "Dataset",
    "MomlevelHipError",
    "derived",
    "dynamic",
    "inverse_barometer",
    "eos",
    "halosteric",
    "reference",
    "steric",
    "steric_variants",
    "test_data",
    "thermosteric",
    "util",
]
