""" steric.py - local and global steric sea level on the MI355X

Drop-in for src/momlevel/steric.py:17-196 -- same names, arguments, defaults,
error behaviour and ``(result, reference)`` return value.  The arithmetic is the
fused HIP path:

* ``domain="global"``: K1 (Wright EOS + rho*volcello_ref + sum over z,y,x per time
  step) gives masso(t); the remaining ``log`` on nt scalars is host arithmetic as in
  the reference (steric.py:136-142).
* ``domain="local"``: K2 (Wright EOS + delta_rho + dz-weighted column sum, dz from
  calc_dz's default path) gives ``delta_rho`` and the sea level field in one pass.

Inputs may be labelled datasets backed by numpy (streamed to HBM in time chunks)
or by device tensors (processed in place), or xarray Datasets when xarray is
installed.
"""

import numpy as np

from . import engine
from .adapters import accepts_xarray
from .labeled import DataArray, Dataset
from .reference import (
    _f32_mode,
    canonical_dims,
    pressure_field,
    pressure_operand,
    setup_reference_state,
)
from .util import annual_average, default_coords, eos_func_from_str, validate_dataset

__all__ = ["halosteric", "steric", "thermosteric"]


def _canonical(da, dims):
    return da.transpose(*dims).data


@accepts_xarray
def steric(
    dset,
    reference=None,
    coord_names=None,
    varname_map=None,
    rhozero=1035.0,
    patm=101325.0,
    equation_of_state="Wright",
    variant="steric",
    domain="local",
    dtype="float32",
    strict=True,
    annual=False,
    verbose=False,
):
    """Function to calculate steric sea level change

    Calculates the steric, thermosteric, or halosteric sea level change and
    associated quantities relative to a reference state, locally at each grid
    point or globally (offline Boussinesq approximation).  Parameters and return
    value as in the reference (src/momlevel/steric.py:33-82); ``dtype`` is output
    encoding metadata only, the arithmetic is float64.

    Returns
    -------
    (result, reference) : tuple of Datasets
    """
    # remap variable names, if passed
    dset = dset.rename(varname_map)

    # default coordinate names
    tcoord, zcoord, zbounds = default_coords(coord_names)

    # conduct some sanity checks on the input dataset
    additional_vars = None if domain == "global" else [zbounds, "deptho"]
    validate_dataset(dset, strict=strict, additional_vars=additional_vars)

    # approximate pressure from depth coordinate (1 m ~ 1 dbar = 1e4 Pa) + patm
    pres = pressure_field(dset, zcoord, patm)

    if reference is not None:
        assert isinstance(reference, Dataset), "`reference` must be an xarray Dataset"
        if verbose:
            print("Using supplied reference state")
    else:
        reference = setup_reference_state(
            dset, patm=patm, eos=equation_of_state, coord_names=coord_names
        )
        if verbose:
            print("Generating reference state from first timestep")

    # conduct some sanity checks on the reference state
    validate_dataset(reference, reference=True, strict=strict)

    # determine which fields, if any, to hold fixed
    if variant == "thermosteric":
        thetao = dset["thetao"]
        so = reference["so"]
    elif variant == "halosteric":
        thetao = reference["thetao"]
        so = dset["so"]
    elif variant == "steric":
        thetao = dset["thetao"]
        so = dset["so"]
    else:
        raise ValueError(f"Unknown variant '{variant}' passed to `steric`")

    # canonical (time, z, y, x) layout -- outputs are always time-first (steric.py:154,165)
    streamed = thetao if tcoord in thetao.dims else so
    cdims3 = canonical_dims(streamed, tcoord, zcoord)
    cdims4 = (tcoord,) + cdims3
    hdims = cdims3[1:]

    def field(da):
        return _canonical(da, cdims4 if tcoord in da.dims else cdims3)

    T, S = field(thetao), field(so)
    vol0 = _canonical(reference["volcello"], cdims3)
    p = pressure_operand(pres, tcoord, cdims3)
    eos_func_from_str(equation_of_state)  # unknown EOS -> ValueError (util.py:247)
    eos = equation_of_state.lower()

    def coords_for(dims):
        return {d: dset[d] for d in dims if d in dset.variables}

    result = Dataset()

    if domain == "global":
        masso = engine.global_masso(T, S, vol0, p, eos=eos, f32_mode=_f32_mode())
        masso = masso.cpu().numpy()
        volo = np.float64(reference["volo"].values)
        rhoga = np.float64(reference["rhoga"].values)
        area_sum = np.float64(reference["areacello"].sum().values)
        reference_height, sealevel, _expansion = engine.global_finalize(
            masso, volo, rhoga, area_sum
        )
        rh = DataArray(np.float64(reference_height), (), None,
                       {"long_name": "Reference column height", "units": "m"})
        result["reference_height"] = rh
        result["reference_height"].encoding["dtype"] = dtype
        result[variant] = DataArray(sealevel, (tcoord,), coords_for((tcoord,)))
    else:
        # calc_dz's input checks (derived.py:284-292) stay on the host
        deptho = dset["deptho"].transpose(*hdims)
        assert bool(np.all(np.nan_to_num(deptho.values, nan=0.0) >= 0)), (
            "Depth values must all be positive-definite"
        )
        assert bool(np.all(dset[zcoord].values >= 0)), (
            "Vertical coordinate levels must all be positive-definite"
        )
        assert bool(np.all(dset[zbounds].values >= 0)), (
            "Vertical coordinate interfaces must all be positive-definite"
        )
        rho0 = _canonical(reference["rho"], cdims3)
        delta_rho, sealevel = engine.local_steric(
            T, S, rho0, vol0, p, rhozero, z_i=dset[zbounds].data, deptho=deptho.data,
            eos=eos, f32_mode=_f32_mode(), want_delta_rho=True,
        )
        dr = DataArray(delta_rho, cdims4, coords_for(cdims4))
        dr.attrs = {
            "long_name": "change in in situ density from reference state",
            "units": "kg m-3",
        }
        result["delta_rho"] = dr
        result["delta_rho"].encoding["dtype"] = dtype
        result[variant] = DataArray(sealevel, (tcoord,) + hdims, coords_for((tcoord,) + hdims))

    # fix up variable metadata
    result[variant].attrs.update(
        {"long_name": f"{variant.capitalize()} height adjustment", "units": "m"}
    )
    result[variant].encoding["dtype"] = dtype

    # copy coordinate and dimension attributes
    for var in set(result.coords).union(result.dims):
        if var in dset.variables and var in result.variables:
            result[var].attrs.update(dset[var].attrs)

    if annual:
        result = annual_average(result)

    return (result, reference)


def halosteric(*args, **kwargs):
    """Wrapper for halosteric calculation"""
    result, reference = steric(*args, **kwargs, variant="halosteric")
    return (result, reference)


def thermosteric(*args, **kwargs):
    """Wrapper for thermosteric calculation"""
    result, reference = steric(*args, **kwargs, variant="thermosteric")
    return (result, reference)
