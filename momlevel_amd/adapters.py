"""xarray <-> momlevel_amd.labeled conversion at the public boundary.

The reference's functions take and return xarray objects.  xarray is optional
here: ``accepts_xarray`` makes a function written against the labelled classes
accept ``xarray.DataArray`` / ``xarray.Dataset`` arguments and, when it got any,
return xarray objects again.  The conversion is pure relabelling (dims, coords,
attrs, encoding); array data is shared, never copied.
"""

import functools

import numpy as np

from .labeled import DataArray, Dataset, is_lazy

try:  # optional
    import xarray as xr
except Exception:  # pragma: no cover - xarray is absent from the build image
    xr = None


def have_xarray():
    return xr is not None


def is_xarray(obj):
    return xr is not None and isinstance(obj, (xr.DataArray, xr.Dataset))


def _raw(var):
    """The array behind an xarray variable WITHOUT loading it: numpy stays numpy, a dask (or any
    other lazily evaluated) array is handed on as it is -- the engine reads it one time chunk at a
    time -- and only exotic wrappers fall back to ``.values``."""
    data = var.data
    if isinstance(data, np.ndarray) or is_lazy(data):
        return data
    return var.values


def from_xarray(obj):
    """xarray.DataArray/Dataset -> labelled DataArray/Dataset (data shared)."""
    if xr is None or not isinstance(obj, (xr.DataArray, xr.Dataset)):
        return obj
    if isinstance(obj, xr.DataArray):
        coords = {
            str(k): DataArray(v.values, tuple(map(str, v.dims)), None, dict(v.attrs), str(k))
            for k, v in obj.coords.items()
        }
        out = DataArray(_raw(obj), tuple(map(str, obj.dims)), coords, dict(obj.attrs), obj.name)
        out.encoding = dict(obj.encoding)
        return out
    out = Dataset(attrs=dict(obj.attrs))
    for k, v in obj.coords.items():
        out._set(str(k), DataArray(v.values, tuple(map(str, v.dims)), None, dict(v.attrs)),
                 is_coord=True)
    for k, v in obj.data_vars.items():
        da = DataArray(_raw(v), tuple(map(str, v.dims)), None, dict(v.attrs))
        da.encoding = dict(v.encoding)
        out[str(k)] = da
    return out


def to_xarray(obj):
    """labelled DataArray/Dataset -> xarray (device data is copied to the host)."""
    if xr is None:
        raise RuntimeError("xarray is not installed")
    if isinstance(obj, DataArray):
        coords = {k: (v.dims, v.values, v.attrs) for k, v in obj.coords.items()}
        out = xr.DataArray(obj.values, dims=obj.dims, coords=coords, attrs=obj.attrs,
                           name=obj.name)
        out.encoding.update(obj.encoding)
        return out
    if isinstance(obj, Dataset):
        out = xr.Dataset(attrs=obj.attrs)
        for k, v in obj.coords.items():
            out.coords[k] = (v.dims, v.values, v.attrs)
        for k, v in obj.data_vars.items():
            out[k] = (v.dims, v.values, v.attrs)
            out[k].encoding.update(v.encoding)
        return out
    return obj


def _map(obj, fn):
    if isinstance(obj, tuple):
        return tuple(_map(o, fn) for o in obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    return fn(obj)


def accepts_xarray(func):
    """Let a labelled-array function take xarray arguments and answer in kind."""

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        used = any(is_xarray(a) for a in args) or any(is_xarray(v) for v in kwargs.values())
        if not used:
            return func(*args, **kwargs)
        args = [from_xarray(a) for a in args]
        kwargs = {k: from_xarray(v) for k, v in kwargs.items()}
        return _map(func(*args, **kwargs), to_xarray)

    return wrapper
