"""A tiny stand-in for the slice of the xarray API momlevel_amd.adapters touches.  TEST ONLY:
xarray is not installed in this image, so the adapter would otherwise never execute.  It mimics
documented xarray behaviour (constructor signatures, .dims/.coords/.attrs/.encoding/.data_vars,
tuple assignment ``ds[name] = (dims, data, attrs)``); it is not a substitute for a run against
the real library (tests/test_gpu_steric.py::test_xarray_round_trip_if_available does that where
xarray exists)."""

import numpy as np


class DataArray:
    def __init__(self, data=None, coords=None, dims=None, name=None, attrs=None):
        self.data = np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(self.data.ndim))
        self.attrs = dict(attrs or {})
        self.encoding = {}
        self.name = name
        self.coords = {}
        for k, v in (coords or {}).items():
            if isinstance(v, tuple):
                v = DataArray(v[1], dims=v[0], attrs=v[2] if len(v) > 2 else None, name=k)
            self.coords[k] = v

    @property
    def values(self):
        return np.asarray(self.data)


class _Coords(dict):
    def __init__(self, owner):
        super().__init__()
        self._owner = owner

    def __setitem__(self, key, value):
        if isinstance(value, tuple):
            value = DataArray(value[1], dims=value[0], attrs=value[2] if len(value) > 2 else None,
                              name=key)
        super().__setitem__(key, value)


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.attrs = dict(attrs or {})
        self.coords = _Coords(self)
        self.data_vars = {}
        for k, v in (coords or {}).items():
            self.coords[k] = v
        for k, v in (data_vars or {}).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, tuple):
            value = DataArray(value[1], dims=value[0], attrs=value[2] if len(value) > 2 else None,
                              name=key)
        value.coords = {c: v for c, v in self.coords.items() if set(v.dims) <= set(value.dims)}
        self.data_vars[key] = value

    def __getitem__(self, key):
        return self.data_vars[key] if key in self.data_vars else self.coords[key]
