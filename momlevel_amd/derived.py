"""derived.py -- the reference's ``momlevel.derived`` hot subset on the MI355X.

Same signatures as src/momlevel/derived.py: ``calc_rho`` (:597-639), ``calc_masso``
(:414-444), ``calc_volo`` (:769-795), ``calc_rhoga`` (:642-666), ``calc_dz``
(:249-325), plus ``calc_alpha`` / ``calc_beta`` (:74-159) which reuse the EOS
kernel.  Inputs and outputs are labelled arrays (momlevel_amd.labeled, or xarray
objects when xarray is installed -- see adapters.py); every number comes from a
HIP kernel.
"""

import numpy as np
import torch

from . import core, engine, hostio, util
from .adapters import accepts_xarray
from .labeled import DataArray

__all__ = [
    "calc_alpha",
    "calc_beta",
    "calc_dz",
    "calc_masso",
    "calc_pdens",
    "calc_rho",
    "calc_rhoga",
    "calc_volo",
]


def _broadcast_dims(*arrays):
    """Union of dims in first-appearance order (xarray's apply_ufunc rule)."""
    dims = []
    for a in arrays:
        for d in a.dims:
            if d not in dims:
                dims.append(d)
    return tuple(dims)


def _expand_to(da, dims, sizes):
    """Raw data of ``da`` reshaped/transposed so it broadcasts against ``dims``."""
    data = da.data
    order = [d for d in dims if d in da.dims]
    perm = [da.dims.index(d) for d in order]
    if perm != list(range(len(perm))):
        data = data.permute(*perm) if isinstance(data, torch.Tensor) else data.transpose(perm)
    shape = [sizes[d] if d in da.dims else 1 for d in dims]
    return data.reshape(shape)


def _apply_eos(func_name, thetao, so, pres, eos, eos_func=None):
    """xr.apply_ufunc(eos_func, thetao, so, pres) restated: broadcast by dim name."""
    if eos_func is None:
        eos_func = util.eos_func_from_str(eos, func_name=func_name)
    # python floats / ints stay python scalars (xr.apply_ufunc hands them through): numpy lets them
    # take the dtype of the arrays they meet -- calc_pdens' pressure on float32 fields
    weak = [isinstance(a, (bool, int, float)) and not isinstance(a, np.generic)
            for a in (thetao, so, pres)]
    args = [a if (isinstance(a, DataArray) or w) else DataArray(np.asarray(a), ())
            for a, w in zip((thetao, so, pres), weak)]
    arrays = [a for a, w in zip(args, weak) if not w]
    dims = _broadcast_dims(*arrays)
    sizes = {}
    for a in arrays:
        sizes.update(a.sizes)
    raw = [a if w else _expand_to(a, dims, sizes) for a, w in zip(args, weak)]
    out = eos_func(*raw)
    if not arrays:  # python scalars only: a 0-d result
        return DataArray(np.asarray(out), ())
    coords = {}
    for a in arrays:
        coords.update(a.coords)
    return DataArray(out, dims, {k: v for k, v in coords.items() if set(v.dims) <= set(dims)})


@accepts_xarray
def calc_rho(thetao, so, pres, eos="Wright"):
    """In situ density from potential temperature, salinity and pressure (derived.py:597-639)."""
    rho = _apply_eos("density", thetao, so, pres, eos)
    rho.attrs = {
        "standard_name": "sea_water_density",
        "long_name": "In situ sea water density",
        "comment": f"calculated with the {eos} equation of state",
        "units": "kg m-3",
    }
    return rho


@accepts_xarray
def calc_pdens(thetao, so, level=0.0, patm=101325, eos="Wright"):
    """Potential density referenced to depth ``level`` (p = level*1e4 + patm) (derived.py:447-486)."""
    assert 0.0 <= level <= 7500.0, "specified level must be between 0 and 7500 m"
    rhopot = _apply_eos("density", thetao, so, (level * 1.0e4) + patm, eos)
    rhopot.attrs = {
        "standard_name": "sea_water_potential_density",
        "long_name": f"Sea water potential density referenced to {level} m",
        "comment": f"calculated with the {eos} equation of state",
        "units": "kg m-3",
    }
    return rhopot


@accepts_xarray
def calc_alpha(thetao, so, pres, eos="Wright"):
    """Thermal expansion coefficient (derived.py:74-115)."""
    alpha = _apply_eos("alpha", thetao, so, pres, eos)
    alpha.attrs = {
        "long_name": "Thermal expansion coefficient",
        "comment": f"calculated with the {eos} equation of state",
        "units": "degC-1",
    }
    return alpha


@accepts_xarray
def calc_beta(thetao, so, pres, eos="Wright"):
    """Haline contraction coefficient (derived.py:118-159)."""
    beta = _apply_eos("beta", thetao, so, pres, eos)
    beta.attrs = {
        "long_name": "Haline contraction coefficient",
        "comment": f"calculated with the {eos} equation of state",
        "units": "PSU-1",
    }
    return beta


@accepts_xarray
def calc_masso(rho, volcello, tcoord="time"):
    """Total ocean mass: sum(rho*volcello) over every non-time dim, skipna (derived.py:414-444)."""
    dims = _broadcast_dims(rho, volcello)
    sizes = dict(volcello.sizes)
    sizes.update(rho.sizes)
    has_t = tcoord in dims
    inner_dims = tuple(d for d in dims if d != tcoord)
    inner = tuple(sizes[d] for d in inner_dims)
    nt = sizes[tcoord] if has_t else 1
    dev = engine.device_of(rho.data, volcello.data)

    def canon(da, with_time):
        order = ((tcoord,) if with_time else ()) + inner_dims
        raw = _expand_to(da, order, sizes)
        raw = engine.to_device(raw, dev, torch.float64)
        full = ((nt,) if with_time else ()) + inner
        return raw.expand(full).contiguous().reshape((nt, -1) if with_time else (-1,))

    r = canon(rho, has_t)
    if not has_t:
        r = r.reshape(1, -1)
    v = canon(volcello, tcoord in volcello.dims)
    out = core.masso(r, v)
    data = out if has_t else out[0]
    if not (rho.is_device or volcello.is_device):
        data = hostio.to_host(data)
    masso = DataArray(data, (tcoord,) if has_t else (),
                      {tcoord: rho.coords[tcoord]} if has_t and tcoord in rho.coords else None)
    masso.attrs = {
        "standard_name": "sea_water_mass",
        "long_name": "Sea Water Mass",
        "units": "kg",
    }
    return masso


@accepts_xarray
def calc_volo(volcello):
    """Total ocean volume: skipna sum of a 3-D volcello (derived.py:769-795)."""
    assert len(volcello.dims) == 3, "Expecting only 3 dimensions for volcello"
    dev = engine.device_of(volcello.data)
    total = core.nansum(engine.to_device(volcello.data, dev, torch.float64))
    # numpy's sum of a float32 field is a float32: the float64 sum of the same values (mlx_nansum),
    # rounded ONCE -- closer to the exact sum than numpy's own float32 accumulation
    if engine.sum_dtype(volcello.data) == np.float32:
        total = total.to(torch.float32)
    volo = DataArray(total if volcello.is_device else hostio.to_host(total), ())
    volo.attrs = {
        "standard_name": "sea_water_volume",
        "long_name": "Sea Water Volume",
        "units": "m3",
    }
    return volo


@accepts_xarray
def calc_rhoga(masso, volo):
    """Global average ocean density masso/volo (derived.py:642-666); host scalars."""
    rhoga = masso / volo
    rhoga.attrs = {
        "long_name": "Global Average Sea Water Density",
        "units": "kg m-3",
    }
    return rhoga


@accepts_xarray
def calc_dz(levels, interfaces, depth, top=0.0, bottom=None, fraction=False):
    """dz with partial bottom cells (derived.py:249-325); dims (z, y, x)."""
    # sign checks stay on the host (derived.py:284-292)
    assert bool(np.all(np.nan_to_num(depth.values, nan=0.0) >= 0)), (
        "Depth values must all be positive-definite"
    )
    assert bool(np.all(levels.values >= 0)), (
        "Vertical coordinate levels must all be positive-definite"
    )
    assert bool(np.all(interfaces.values >= 0)), (
        "Vertical coordinate interfaces must all be positive-definite"
    )
    dev = engine.device_of(depth.data)
    out = core.calc_dz(
        engine.to_device(interfaces.data, dev, torch.float64),
        engine.to_device(depth.data, dev, torch.float64).contiguous(),
        top=top, bottom=bottom, fraction=fraction,
    )
    zdim = levels.dims[0]
    coords = dict(depth.coords)
    coords[zdim] = levels.coords.get(zdim, levels)
    return DataArray(out if depth.is_device else hostio.to_host(out), (zdim,) + depth.dims, coords)
