""" reference.py - module to establish sea level reference states (MI355X)

Mirror of src/momlevel/reference.py:15-85.  The three (z,y,x) slabs of the chosen
time index are moved to the device once; rho0 (K0), volo (mlx_nansum) and masso0
(K1 on a one-step view, so that it is bit-identical to masso(t=time_index) of the
time loop) are computed there.
"""

import numpy as np

from . import engine
from .adapters import accepts_xarray
from .labeled import DataArray, Dataset
from .util import default_coords, eos_func_from_str

__all__ = ["setup_reference_state"]


def canonical_dims(da, tcoord, zcoord):
    """(z, <horizontal dims in their given order>) of a 3-D/4-D field."""
    hdims = tuple(d for d in da.dims if d not in (tcoord, zcoord))
    return (zcoord,) + hdims


def pressure_field(dset, zcoord, patm):
    """pres = z*1e4 + patm (steric.py:93-96); patm may be a float or a DataArray."""
    return (dset[zcoord] * 1.0e4) + patm


def pressure_operand(pres, tcoord, cdims):
    """Raw pressure for the kernels: scalar, z-profile, or a (z,y,x)-broadcastable array
    in canonical dim order (patm given as a DataArray, steric.py:58-60)."""
    if tcoord in pres.dims:
        raise ValueError("a time-dependent `patm` is not supported by the fused steric path")
    if pres.ndim <= 1:
        return pres.data
    sub = tuple(d for d in cdims if d in pres.dims)
    vals = pres.transpose(*sub).values
    return vals.reshape([pres.sizes.get(d, 1) for d in cdims])


def _f32_mode():
    import os

    return os.environ.get("MOMLEVEL_AMD_F32_MODE", "faithful")


def _setup(dset, patm, eos, coord_names, time_index, defer_masso):
    """setup_reference_state, optionally leaving masso / rhoga to the caller (steric() with
    domain="global" reads them off its own K1 launch: engine.reference_state)."""
    coords = default_coords(coord_names)
    tcoord = coords[0]
    zcoord = coords[1]

    eos_func_from_str(eos)  # unknown EOS -> ValueError, as calc_rho raises it (util.py:247)

    pres = pressure_field(dset, zcoord, patm)

    reference = Dataset()
    for name in ("thetao", "so", "volcello"):
        reference[name] = (
            dset[name].isel({tcoord: time_index}).squeeze().reset_coords(drop=True)
        )

    cdims = canonical_dims(reference["thetao"], tcoord, zcoord)
    T0 = reference["thetao"].transpose(*cdims)
    S0 = reference["so"].transpose(*cdims)
    V0 = reference["volcello"].transpose(*cdims)
    p = pressure_operand(pres, tcoord, cdims)

    on_device = T0.is_device or S0.is_device or V0.is_device
    rho0, volo, masso0 = engine.reference_state(
        T0.data, S0.data, V0.data, p, eos=eos.lower(), f32_mode=_f32_mode(),
        with_masso=not defer_masso,
    )

    rho = DataArray(rho0 if on_device else rho0.cpu().numpy(), cdims, T0.coords)
    rho.attrs = {
        "standard_name": "sea_water_density",
        "long_name": "In situ sea water density",
        "comment": f"calculated with the {eos} equation of state",
        "units": "kg m-3",
    }
    reference["rho"] = rho.transpose(*reference["thetao"].dims)

    volo_h = float(volo.item())
    masso_h = float("nan") if masso0 is None else float(masso0.item())
    reference["volo"] = DataArray(
        np.array(volo_h), (), None,
        {"standard_name": "sea_water_volume", "long_name": "Sea Water Volume", "units": "m3"},
    )
    reference["masso"] = DataArray(
        np.array(masso_h), (), None,
        {"standard_name": "sea_water_mass", "long_name": "Sea Water Mass", "units": "kg"},
    )
    reference["rhoga"] = DataArray(
        np.array(np.float64(masso_h) / np.float64(volo_h)), (), None,
        {"long_name": "Global Average Sea Water Density", "units": "kg m-3"},
    )
    reference["areacello"] = dset["areacello"]
    return reference


def set_reference_masso(reference, masso0):
    """Fill the deferred masso / rhoga of a reference built with ``defer_masso``."""
    masso0 = np.float64(masso0)
    reference["masso"].data[...] = masso0
    reference["rhoga"].data[...] = masso0 / np.float64(reference["volo"].values)


@accepts_xarray
def setup_reference_state(
    dset, patm=101325.0, eos="Wright", coord_names=None, time_index=0
):
    """Function to generate reference dataset

    Values are taken from time level ``time_index`` of an input dataset holding
    thetao, so, volcello and areacello (src/momlevel/reference.py:15-85).

    Returns
    -------
    Dataset of reference values: thetao, so, volcello, rho (z,y,x); volo, masso,
    rhoga (scalars); areacello.
    """
    return _setup(dset, patm, eos, coord_names, time_index, defer_masso=False)
