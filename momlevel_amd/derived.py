"""derived.py -- the reference's ``momlevel.derived`` hot subset on the MI355X.

Same signatures as src/momlevel/derived.py: ``calc_rho`` (:597-639), ``calc_masso``
(:414-444), ``calc_volo`` (:769-795), ``calc_rhoga`` (:642-666), ``calc_dz``
(:249-325), plus ``calc_alpha`` / ``calc_beta`` (:74-159) which reuse the EOS
kernel, and their consumers: ``calc_n2`` (:328-411),
``calc_stability_angle`` (:714-766), ``adjust_negative_n2`` (:30-71) and
``calc_wave_speed`` (:798-831), each ONE pass of a HIP kernel of its own
(csrc/momlevel_strat.hip) over what the reference evaluates as dozens of numpy passes.  Inputs and outputs are labelled arrays (momlevel_amd.labeled, or xarray
objects when xarray is installed -- see adapters.py); every number comes from a
HIP kernel.
"""

import numpy as np
import torch

from . import core, engine, hostio, util
from .adapters import accepts_xarray
from .labeled import DataArray, is_lazy

__all__ = [
    "adjust_negative_n2",
    "calc_alpha",
    "calc_beta",
    "calc_dz",
    "calc_masso",
    "calc_n2",
    "calc_pdens",
    "calc_rho",
    "calc_rhoga",
    "calc_stability_angle",
    "calc_volo",
    "calc_wave_speed",
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
    if da.is_lazy:  # a dask / netCDF4-like field
        if tuple(da.dims) == tuple(dims):
            return data  # already laid out like the result: read piece by piece (eos/_dispatch.py)
        data = da.values  # (reads all of it; masked elements of a netCDF4 read become NaN)
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


# ---------------------------------------------------------------------------------------
# stratification: the consumers of alpha / beta (csrc/momlevel_strat.hip)
# ---------------------------------------------------------------------------------------
def _z_layout(da, zcoord):
    """(index of zcoord, nt, nz, plane): the field seen as (nt, nz, plane), z the middle axis."""
    if zcoord not in da.dims:
        raise ValueError(f"{zcoord!r} is not a dimension of the field (dims {da.dims})")
    zi = da.dims.index(zcoord)
    shape = tuple(int(v) for v in da.shape)
    return zi, int(np.prod(shape[:zi], dtype=np.int64)), shape[zi], int(np.prod(shape[zi + 1:], dtype=np.int64))


def _level_values(da, zcoord):
    if zcoord not in da.coords:
        raise ValueError(f"the field has no coordinate values for {zcoord!r}")
    return np.asarray(da.coords[zcoord].values)


def _strat_pressure(pres, like, zcoord, dev, f32_fields):
    """The pressure operand of alpha / beta as core.stratification takes it.  float32 fields need
    a float64 pressure ARRAY: with a python float or a float32 array numpy evaluates alpha in
    float32 (eos/_dispatch.py), which this kernel does not restate."""
    if isinstance(pres, DataArray):
        if f32_fields and str(pres.dtype) != "float64":
            raise TypeError("float32 thetao/so need a float64 pressure (numpy evaluates alpha and "
                            f"beta in float32 against a {pres.dtype} one): convert it")
        if not set(pres.dims) <= set(like.dims):
            raise ValueError(f"pressure dims {pres.dims} are not dims of the field {like.dims}")
        zi, nt, nz, plane = _z_layout(like, zcoord)
        if pres.dims == (zcoord,):
            return engine.to_device(pres.data, dev, torch.float64).reshape(nz)
        if pres.dims == ():
            return engine.to_device(pres.data, dev, torch.float64).reshape(1)
        sizes = dict(like.sizes)
        raw = engine.to_device(_expand_to(pres, like.dims, sizes), dev, torch.float64)
        if zi > 0 and all(d not in pres.dims for d in like.dims[:zi]):  # no leading (time) dims
            return raw.reshape(raw.shape[zi:]).expand(tuple(like.shape[zi:])).contiguous().reshape(nz, plane)
        return raw.expand(tuple(like.shape)).contiguous().reshape(nt, nz, plane)
    if f32_fields:
        raise TypeError("float32 thetao/so need a float64 pressure array (numpy evaluates alpha and "
                        "beta in float32 against a python float): pass a DataArray")
    return float(pres)


def _stratification(func, thetao, so, pres, eos, zcoord, gravity=-9.8):
    if tuple(so.dims) != tuple(thetao.dims) or tuple(so.shape) != tuple(thetao.shape):
        raise ValueError(f"thetao {thetao.dims}{tuple(thetao.shape)} and so {so.dims}{tuple(so.shape)} "
                         "must share dims and shape")
    if str(thetao.dtype) != str(so.dtype):
        raise TypeError("thetao and so of different dtypes: convert one of them")
    if str(thetao.dtype) not in ("float32", "float64"):
        raise TypeError(f"thetao/so must be float32 or float64, not {thetao.dtype}")
    f32 = str(thetao.dtype) == "float32"
    if f32 and eos.lower() == "linear":
        raise TypeError("the linear EOS on float32 fields is float32 throughout in numpy: "
                        "convert the fields to float64")
    util.eos_func_from_str(eos, func_name="alpha")  # unknown EOS: the reference's ValueError
    zi, nt, nz, plane = _z_layout(thetao, zcoord)
    z = _level_values(thetao, zcoord)
    if f32 and str(z.dtype) == "float32":
        # numpy.gradient takes its coefficients from the coordinate IN THE COORDINATE'S dtype: with a
        # float32 z and float32 fields the edge rows (every row of an uneven grid) are float32
        # arithmetic, and calc_n2's own pressure `thetao[zcoord] * 1e4 + patm` is float32 too, so
        # that numpy evaluates alpha, beta and N^2 in float32 throughout.  The kernel forms the
        # derivative in float64 and rounds once: not numpy's bits -- refused, like a
        # float32 pressure on float32 fields.
        raise TypeError(f"float32 thetao/so with a float32 {zcoord!r} coordinate: numpy would "
                        "differentiate (and calc_n2 build its pressure) in float32, which no kernel "
                        "here restates; convert the coordinate to float64")
    dev = engine.device_of(thetao.data, so.data)
    dt = torch.float32 if f32 else torch.float64
    p = None if eos.lower() == "linear" else _strat_pressure(pres, thetao, zcoord, dev, f32)
    kw = dict(func=func, eos=eos.lower(), gravity=gravity)
    on_device = thetao.is_device or so.is_device
    if not on_device and nt > 1 and nt * nz * plane > _HOST_PIPELINE_ELEMS:
        out = _stratification_host_rows(thetao.data, so.data, p, z, nt, nz, plane, dev,
                                        lead=nt if zi == 1 else None, **kw)
        return DataArray(out.reshape(tuple(thetao.shape)), thetao.dims, dict(thetao.coords))
    T = engine.to_device(thetao.data, dev, dt).reshape(nt, nz, plane)
    S = engine.to_device(so.data, dev, dt).reshape(nt, nz, plane)
    out = core.stratification(T, S, p, z, **kw).reshape(tuple(thetao.shape))
    return DataArray(out if on_device else hostio.to_host(out), thetao.dims, dict(thetao.coords))


# host fields above this size are evaluated in groups of rows (the dimensions before z: time
# steps), the groups' uploads, kernels and result downloads overlapping -- as the pointwise EOS
# functions do (eos/_dispatch.py) -- and the device never holds more than a few groups
_HOST_PIPELINE_ELEMS = 1 << 26
_HOST_GROUP_ELEMS = 1 << 25


def _stratification_host_rows(T, S, p, z, nt, nz, plane, dev, lead=None, **kw):
    """core.stratification on host fields seen as (nt, nz, plane), group of rows by group: rows
    are independent (the derivative runs along z), so group k+1 uploads (hostio.Uploader) while
    group k's kernel runs and group k-1's result leaves (hostio.Downloader).  ``lead``: the
    fields are (lead, nz, ...) with ONE dimension before z (the usual (time, z, y, x)): they are
    then sliced along it as they are -- a lazy field (dask / netCDF4 / h5py-like) is read group by
    group in the upload worker and never materialised whole.  (A pressure that varies from row to
    row -- ``p`` of shape (nt, nz, plane) -- IS resident whole: _strat_pressure expanded it before
    the groups start; only theta/S and the result are streamed.)"""
    if lead is not None:
        Tn, Sn = T, S  # sliced along their own leading axis
    else:
        Tn = hostio.as_plain(T[...] if is_lazy(T) else T).reshape(nt, nz, plane)
        Sn = hostio.as_plain(S[...] if is_lazy(S) else S).reshape(nt, nz, plane)
    rows = max(1, _HOST_GROUP_ELEMS // (nz * plane))
    bounds = [(i0, min(i0 + rows, nt)) for i0 in range(0, nt, rows)]
    p_rows = isinstance(p, torch.Tensor) and p.dim() == 3  # a pressure that varies from row to row
    out = np.empty((nt, nz, plane), dtype=np.float64)
    main = torch.cuda.current_stream(dev)
    up = hostio.Uploader(dev)
    try:
        with hostio.Downloader(dev) as results:
            nxt = up.submit([hostio.leading_slice(Tn, *bounds[0]), hostio.leading_slice(Sn, *bounds[0])])
            for n, (i0, i1) in enumerate(bounds):
                (Td, Sd), ready = nxt.result()  # (re-raises what the worker raised)
                if n + 1 < len(bounds):
                    j0, j1 = bounds[n + 1]
                    nxt = up.submit([hostio.leading_slice(Tn, j0, j1), hostio.leading_slice(Sn, j0, j1)])
                main.wait_event(ready)
                res = core.stratification(Td.reshape(i1 - i0, nz, plane), Sd.reshape(i1 - i0, nz, plane),
                                          p[i0:i1] if p_rows else p, z, **kw)
                results.submit([(out[i0:i1], res)])
    finally:
        up.close()
    return out


@accepts_xarray
def calc_n2(thetao, so, eos="Wright", gravity=-9.8, patm=101325.0, zcoord="z_l", interfaces=None,
            adjust_negative=False):
    """Buoyancy frequency N^2 at the cell centres (derived.py:328-411):
    ``gravity * ((alpha * dT/dz) - (beta * dS/dz))`` with alpha, beta at
    ``p = thetao[zcoord] * 1e4 + patm`` and d/dz = ``differentiate(zcoord, edge_order=2)``.
    ``interfaces`` (the cell-edge variant) interpolates with xgcm's ``Grid.transform`` in the
    reference: not built.  Deviations, deliberate: ``adjust_negative=True`` forwards ``zcoord``
    (the reference calls ``adjust_negative_n2(n2)`` with its default "z_l", derived.py:409, and so
    fails on any other name); float32 fields need a float64 coordinate (TypeError otherwise, see
    _stratification)."""
    if interfaces is not None:
        raise NotImplementedError("calc_n2(interfaces=...) needs xgcm's linear vertical transform "
                                  "(derived.py:389-394): not built; pass interfaces=None")
    z = DataArray(_level_values(thetao, zcoord).astype(np.float64), (zcoord,))
    pres = (z * 1.0e4) + patm  # derived.py:396
    n2 = _stratification("n2", thetao, so, pres, eos, zcoord, gravity=gravity)
    n2.attrs = {
        "standard_name": "square_of_brunt_vaisala_frequency_in_sea_water",
        "long_name": "Square of seawater buoyancy frequency",
        "units": "s-2",
    }
    return adjust_negative_n2(n2, zcoord=zcoord) if adjust_negative else n2


@accepts_xarray
def calc_stability_angle(thetao, so, pres, eos="Wright", zcoord="z_l"):
    """Stability (Turner) angle in degrees (derived.py:714-766)."""
    result = _stratification("turner", thetao, so, pres, eos, zcoord)
    result.name = "tu_angle"
    result.attrs = {
        "long_name": "Stability angle",
        "units": "degrees",
    }
    return result


def _adjust(n2, zcoord, dz=None, want_adjusted=True):
    zi, nt, nz, plane = _z_layout(n2, zcoord)
    # `adjusted[0]` (derived.py:62) indexes the LEADING dimension: the rows of the (nt, nz, plane)
    # view whose index along it is 0 -- all nt // size0 of them -- unless z leads
    lead0_rows = 0 if zi == 0 else nt // int(n2.shape[0])
    dev = engine.device_of(n2.data)
    x = engine.to_device(n2.data, dev, torch.float64).reshape(nt, nz, plane)
    dzt = None
    if dz is not None:
        if tuple(dz.dims) != tuple(n2.dims[zi:]) or tuple(dz.shape) != tuple(n2.shape[zi:]):
            raise ValueError(f"dz {dz.dims}{tuple(dz.shape)} must cover the field's "
                             f"{n2.dims[zi:]}{tuple(n2.shape[zi:])}")
        dzt = engine.to_device(dz.data, dev, torch.float64).reshape(nz, plane)
    adjusted, speed = core.adjust_negative_n2(x, lead0_rows, dz=dzt, want_adjusted=want_adjusted)
    return adjusted, speed, (zi, nt, nz, plane), x


@accepts_xarray
def adjust_negative_n2(n2, zcoord="z_l"):
    """Remove negative N^2 after Chelton et al. 1998 (derived.py:30-71): non-positive values
    become NaN, NaN at index 0 OF THE LEADING DIMENSION becomes 1e-8 (the reference writes
    ``adjusted[0]``: the first time step of a (time, z, y, x) field, the surface of a (z, y, x)
    one), the rest is forward-filled down the column, the original NaN mask is put back.
    Deviation: the result is float64 whatever ``n2``'s dtype (the reference keeps a float32 n2
    float32 and writes float32(1e-8)); ``zcoord`` is an addition (the reference ffills "z_l")."""
    adjusted, _, _, _ = _adjust(n2, zcoord)
    adjusted = adjusted.reshape(tuple(n2.shape))
    out = DataArray(adjusted if n2.is_device else hostio.to_host(adjusted), n2.dims, dict(n2.coords))
    out.attrs = {**n2.attrs, "comment": "adjustment applied for negative values"}
    return out


@accepts_xarray
def calc_wave_speed(n2, dz, zcoord="z_l"):
    """Gravity wave speed of the first baroclinic mode (derived.py:798-831):
    ``(sqrt(adjust_negative_n2(n2)) * dz).sum(zcoord) / pi``, NaN where ``n2[0]`` is null.  As in
    the reference ``n2[0]`` indexes the leading dimension: for a (z, y, x) field it is the surface
    and the result has dims (y, x); for a (time, z, y, x) field it is the first TIME STEP, dims
    (z, y, x), which xarray broadcasts against the (time, y, x) speeds into (z, y, x, time) -- the
    array whose sum the reference's own test holds (tests/test_derived.py:147-151).  float64 out
    whatever the dtype of ``n2`` (see adjust_negative_n2)."""
    _, speed, (zi, nt, nz, plane), x = _adjust(n2, zcoord, dz=dz, want_adjusted=False)
    trail_dims, trail_shape = tuple(n2.dims[zi + 1:]), tuple(n2.shape[zi + 1:])
    host = not n2.is_device
    if zi == 0:
        data = speed.reshape(trail_shape)
        result = DataArray(hostio.to_host(data) if host else data, trail_dims,
                           {k: v for k, v in n2.coords.items() if set(v.dims) <= set(trail_dims)})
    elif zi == 1:
        data = core.wave_speed_where_time0(x[0], speed).reshape((nz,) + trail_shape + (nt,))
        dims = (zcoord,) + trail_dims + (n2.dims[0],)
        result = DataArray(hostio.to_host(data) if host else data, dims, dict(n2.coords))
    else:
        raise NotImplementedError("calc_wave_speed: more than one dimension before "
                                  f"{zcoord!r} (dims {n2.dims})")
    result.attrs = {
        "long name": "Ocean gravity wave speed of the first baroclinic mode",
        "units": "m s-1",
    }
    return result
