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


def refuse_float32_pressure(pres, thetao, so, zcoord):
    """steric.py:96 / reference.py:53-54 form ``pres = dset[zcoord] * 1e4 + patm`` in the
    COORDINATE's dtype.  A float32 ``z_l`` gives a float32 pressure; against float64 (or mixed)
    theta / S numpy widens it exactly where it meets them -- which is what the kernels do with it --
    but with float32 theta AND S numpy evaluates the WHOLE equation of state in float32, result
    included, and no steric kernel restates that (``derived.calc_rho`` does: mlx_eos_map_promote).
    Rather than answer in other bits (the silent float64 upcast of rounds 3-5), such a call is
    refused, like float16 fields (labeled.check_field_dtype)."""
    if all(str(x.dtype) == "float32" for x in (pres, thetao, so)):
        raise TypeError(
            f"float32 thetao / so with a float32 pressure (a float32 {zcoord!r} coordinate): numpy "
            "would evaluate the whole equation of state in float32, which the steric kernels do "
            f"not restate; convert the coordinate to float64 (dset[{zcoord!r}].astype('float64'))")


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


def _setup(dset, patm, eos, coord_names, time_index, defer_masso, twins=None):
    """setup_reference_state, optionally leaving masso / rhoga to the caller (steric() with
    domain="global" reads them off its own K1 launch: engine.reference_state).
    ``twins``: a dict that receives the DEVICE tensors of the state's (z,y,x) slabs in canonical
    dim order -- thetao, so, volcello (float64) and rho -- for a caller that goes straight on to
    the kernels (steric()): the host arrays of the returned Dataset are for the user, and moving
    them to the device a second time is 1.1 GB of the host link on the reference's recorded call."""
    coords = default_coords(coord_names)
    tcoord = coords[0]
    zcoord = coords[1]

    eos_func_from_str(eos)  # unknown EOS -> ValueError, as calc_rho raises it (util.py:247)

    pres = pressure_field(dset, zcoord, patm)
    refuse_float32_pressure(pres, dset["thetao"], dset["so"], zcoord)

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
        T0d, S0d, V0d = T0.data, S0.data, V0.data
        if twins is not None:  # (the uploads engine.reference_state would make, kept)
            import torch

            dev = engine.device_of(T0d, S0d, V0d)
            T0d = engine.to_device(T0d, dev, engine._stream_dtype(T0d))
            S0d = engine.to_device(S0d, dev, engine._stream_dtype(S0d))
            V0d = engine.to_device(V0d, dev, torch.float64)
        rho0, volo, masso0 = engine.reference_state(
            T0d, S0d, V0d, p, eos=eos.lower(), f32_mode=_f32_mode(),
            with_masso=not defer_masso,
        )
        if twins is not None:
            twins.update(thetao=T0d, so=S0d, volcello=V0d, rho=rho0)
        if on_device:
            rho_data = rho0
        elif twins is not None and rho0.numel() * 8 >= hostio.SMALL_BYTES:
            # the caller goes straight on to the time loop: rho0 leaves for the host on a stream and a
            # worker of its own WHILE the first chunk is uploaded (the other direction of the link)
            # instead of holding the loop back for its 0.43 GB; the caller completes it before it
            # returns (steric._steric_many: twins["pending"].finish())
            rho_data = hostio.result_array(tuple(rho0.shape), np.float64)
            twins["pending"] = hostio.Downloader(rho0.device)
            twins["pending"].submit([(rho_data, rho0)])
        else:
            rho_data = hostio.to_host(rho0)
        rho = DataArray(rho_data, cdims, T0.coords, rho_attrs)
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
