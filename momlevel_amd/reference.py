""" reference.py - module to establish sea level reference states (MI355X)

Mirror of src/momlevel/reference.py:15-85.  The three (z,y,x) slabs of the chosen
time index are moved to the device once; rho0 (K0), volo (mlx_nansum) and masso0
(K1 on a one-step view, so that it is bit-identical to masso(t=time_index) of the
time loop) are computed there.
"""

import numpy as np

from . import engine, hostio
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
    """Raw pressure for the kernels in canonical dim order: a scalar, the z profile, a
    (z,y,x)-broadcastable array (``patm`` given as a (yh,xh) DataArray, steric.py:58-60), or --
    when ``patm`` carries the time dimension -- a 4-D (time,z,y,x)-broadcastable array, which the
    engine cuts into the same time chunks as theta/S (MLX_P_FULL4D)."""
    if pres.ndim <= 1 and tcoord not in pres.dims:
        return pres.data
    order = ((tcoord,) if tcoord in pres.dims else ()) + tuple(cdims)
    sub = tuple(d for d in order if d in pres.dims)
    vals = pres.transpose(*sub).values
    return vals.reshape([pres.sizes.get(d, 1) for d in order])


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

    rho_attrs = {
        "standard_name": "sea_water_density",
        "long_name": "In situ sea water density",
        "comment": f"calculated with the {eos} equation of state",
        "units": "kg m-3",
    }
    if tcoord in pres.dims:
        # A `patm` with a time dimension makes the reference's rho0 = calc_rho(theta0, S0, pres)
        # time dependent too (reference.py:71 broadcasts by name), and with it masso and rhoga
        # (derived.py:435-438 sums every dim except time).  Reproduced as is: steric() then refuses
        # such a self-made reference in validate_dataset, exactly like momlevel.
        rho0, volo, masso_t = engine.reference_state_time_dependent(
            T0.data, S0.data, V0.data, p, eos=eos.lower(), f32_mode=_f32_mode())
        tdim = (tcoord,)
        rho = DataArray(rho0 if on_device else hostio.to_host(rho0), tdim + cdims,
                        dict(T0.coords, **({tcoord: dset[tcoord]} if tcoord in dset.variables
                                           else {})), rho_attrs)
        reference["rho"] = rho
        volo_h = engine.sum_dtype(V0.data)(volo.item())  # numpy: float32 volcello -> float32 volo
        masso_h = masso_t.cpu().numpy()
        masso_dims, masso_val = tdim, masso_h
        rhoga_val = masso_h / np.float64(volo_h)
    else:
        rho0, volo, masso0 = engine.reference_state(
            T0.data, S0.data, V0.data, p, eos=eos.lower(), f32_mode=_f32_mode(),
            with_masso=not defer_masso,
        )
        rho = DataArray(rho0 if on_device else hostio.to_host(rho0), cdims, T0.coords, rho_attrs)
        reference["rho"] = rho.transpose(*reference["thetao"].dims)
        # derived.py:789: volcello.sum() has volcello's dtype -- float32 for the float32 volumes MOM6
        # writes: the float64 device sum of the same values, rounded once; masso and
        # rhoga = masso / volo are float64 in numpy too (rho is float64)
        volo_h = engine.sum_dtype(V0.data)(volo.item())
        masso_h = float("nan") if masso0 is None else float(masso0.item())
        masso_dims, masso_val = (), np.array(masso_h)
        rhoga_val = np.array(np.float64(masso_h) / np.float64(volo_h))

    reference["volo"] = DataArray(
        np.array(volo_h), (), None,
        {"standard_name": "sea_water_volume", "long_name": "Sea Water Volume", "units": "m3"},
    )
    reference["masso"] = DataArray(
        masso_val, masso_dims, None,
        {"standard_name": "sea_water_mass", "long_name": "Sea Water Mass", "units": "kg"},
    )
    reference["rhoga"] = DataArray(
        rhoga_val, masso_dims, None,
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
