"""Minimal labelled-array containers with the slice of the xarray API the steric path uses.

momlevel's boundary is xarray-in / xarray-out (src/momlevel/steric.py:17-31,184).
xarray is an optional dependency here (it is absent from the build image): when
it is importable, ``momlevel_amd`` accepts and returns real ``xarray`` objects
(see ``adapters.py``); in every case the work is done on these two classes, which
carry names, dims, coords, attrs and encoding but no arithmetic of their own
beyond what input validation and the tests need.

``DataArray.data`` is a numpy array (host), a torch tensor (device) or a LAZY array
-- anything array-shaped that is not one of the two: a dask array, a netCDF4 / h5py /
zarr variable -- which is kept as it is and only ever read slice by slice (the engine
cuts time chunks out of it; momlevel's real inputs are dask-chunked float32 files,
examples/example.ipynb cell 4).  ``.values`` always gives numpy -- for a lazy array
that reads ALL of it, so nothing on the steric path calls it on a 4-D field.
Nothing here is on the hot path: the 4-D fields are only ever *relabelled*; their
numbers go through the HIP kernels.
"""

import numpy as np

try:  # torch is the device-array container; labelled arrays work without it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_tensor(x):
    return torch is not None and isinstance(x, torch.Tensor)


def is_lazy(x):
    """An array-shaped object that is neither numpy nor torch: left unread until sliced."""
    if _is_tensor(x) or isinstance(x, (np.ndarray, np.generic, list, tuple, int, float, bool)):
        return False
    return all(hasattr(x, a) for a in ("shape", "dtype", "__getitem__"))


def dtype_name(dt):
    """"float32", "float64", "int16" ... of a numpy dtype (whatever its BYTE ORDER: ``>f4`` -- what a
    NetCDF3 file holds and scipy.io.netcdf_file hands out -- is float32 to numpy's promotion and to
    the reference, which sees it through xarray in native order) or of a torch dtype."""
    if isinstance(dt, np.dtype) or not str(dt).startswith("torch."):
        return np.dtype(dt).name
    return str(dt).replace("torch.", "")


def np_dtype(dt):
    """The numpy dtype of ``dt`` in the host's byte order: a numpy dtype keeps everything else it is
    (``<U5``, ``S3``, structured and object dtypes have no ``.name`` that numpy reads back: 'str160'
    is not a dtype), a torch dtype is mapped by its name."""
    if isinstance(dt, np.dtype) or not str(dt).startswith("torch."):
        dt = np.dtype(dt)
        return dt.newbyteorder("=") if dt.kind in "fiuc" else dt
    return np.dtype(str(dt).replace("torch.", ""))


_UNSUPPORTED = ("float16", "bfloat16", "half", "float128", "longdouble", "complex64", "complex128",
                "complex256")


def check_field_dtype(dt, what="theta / salinity"):
    """numpy evaluates the equation of state on float16 (or long double) arrays IN that precision --
    the python-float constants of eos/wright.py take the arrays' dtype -- which no kernel here
    restates; rather than answer in other bits (a silent float64 upcast), such fields are refused.
    float32 / float64 pass; integers and booleans compute as float64, as in numpy.

    One known deviation, parity unpinned (no reference fixture holds such a field): a
    LAZILY read integer field of 16 bits or fewer whose slices come back MASKED (an int16
    netCDF4.Variable with a _FillValue and no scale_factor).  xarray decodes such a variable to
    float32 (``maybe_promote``), so the reference would evaluate its part of the polynomial in
    float32; the declared dtype of a lazy source says "int16", which streams as float64 here (the
    masked slices are promoted to float32 by as_plain and then widened exactly).  An IN-MEMORY masked
    int16 array is float32 from the moment it is wrapped, as in xarray.  Convert such a variable to
    float32 before the call to get the reference's bits."""
    name = dtype_name(dt)
    if name in _UNSUPPORTED:
        raise TypeError(f"{name} {what} are not supported: numpy evaluates their part of the "
                        "equation of state in that precision; convert to float32 or float64")
    return name


def _native(a):
    """``a`` in the host's byte order (a copy only when it is not)"""
    return a if a.dtype.isnative else a.astype(a.dtype.newbyteorder("="))


def as_plain(a):
    """What ``xr.DataArray(a)`` holds for a host array ``a`` -- the container every input of the
    reference arrives in (steric.py:84-96; examples/example.ipynb cell 4).  A ``numpy.ma``
    masked array -- what ``netCDF4.Variable.__getitem__`` returns wherever a ``_FillValue`` /
    ``missing_value`` is declared, for the whole variable and for every slice of it -- MEANS NaN at
    its masked elements: xarray's ``as_compatible_data`` replaces them (``where(~mask, NaN)``) after
    promoting the dtype so that it can hold one (floating dtypes keep theirs; integers of up to 16
    bits become float32, wider ones float64 -- ``dtypes.maybe_promote``; a masked array with nothing
    masked keeps its dtype).  ``np.asarray`` would instead DROP the mask and leave the fill values
    (1e20 ...) to be computed on.  Anything else goes through ``np.asarray`` untouched: no copy of a
    plain array, and an unmasked masked array gives a view of its data.  (A dask array is computed
    first: its chunks may be masked arrays too, and its own ``__array__`` would drop their masks.)
    The result is always in the host's BYTE ORDER: big-endian data (NetCDF3 on disk) are float32 /
    float64 like any other to numpy and to the reference; every dtype test downstream compares with
    the native types."""
    if not isinstance(a, np.ndarray) and callable(getattr(a, "compute", None)):
        a = a.compute()
    if not isinstance(a, np.ma.MaskedArray):
        return _native(np.asarray(a))
    mask = np.ma.getmask(a)
    data = _native(np.ma.getdata(a))
    if mask is np.ma.nomask or not mask.any():
        return np.asarray(data)
    kind = data.dtype.kind
    if kind == "f":
        dtype = data.dtype
    elif kind in "iu":
        dtype = np.dtype(np.float32 if data.dtype.itemsize <= 2 else np.float64)
    else:
        raise TypeError(f"a masked array of dtype {data.dtype} has no NaN to stand for its masked "
                        "elements (xarray would make it an object array); convert it to a "
                        "floating dtype first")
    full = np.broadcast_to(mask, data.shape)
    if data.dtype == dtype and data.nbytes >= _NATIVE_FILL_BYTES:
        out = _native_fill(data, full)  # (the library's copy team: host memory bandwidth)
        if out is not None:
            return out
    out = np.array(data, dtype=dtype, order="C")  # a copy: the caller's array is never written
    np.putmask(out, full, np.nan)
    return out


# Above this size the NaN fill of a floating masked array goes to mlx_host_copy_masked (one foreign
# call, a team of native threads): numpy's own single-threaded pass runs at 0.7-1 GB/s, which would
# put a lazily read netCDF4 field two orders of magnitude below the host link it is uploaded over.
_NATIVE_FILL_BYTES = 4 << 20


def _native_fill(data, mask):
    """dst = where(mask, NaN, data) for C-contiguous float32 / float64 ``data`` and a boolean
    ``mask`` of the same shape, by the library's host copy team.  None when the layout does not fit
    or the library cannot be loaded (labelled arrays work without it): the caller falls back to
    numpy."""
    if not (data.flags["C_CONTIGUOUS"] and data.flags["ALIGNED"] and data.dtype.itemsize in (4, 8)
            and data.dtype.isnative):
        return None  # (an element-misaligned view -- frombuffer at an odd offset -- is numpy's)
    mask = np.ascontiguousarray(mask, dtype=np.bool_)  # (a broadcast scalar mask is expanded here)
    try:
        from . import _lib, hostio

        lib = _lib.load()
    except Exception:  # no library (a CPU-only install of the labelled layer): numpy does it
        return None
    out = np.empty(data.shape, dtype=data.dtype)
    rc = lib.mlx_host_copy_masked(out.ctypes.data, data.ctypes.data, mask.ctypes.data, data.size,
                                  data.dtype.itemsize, hostio.host_threads())
    if rc != 0:
        raise RuntimeError(f"mlx_host_copy_masked failed ({rc}): {_lib.last_error()}")
    return out


class MaskedSource:
    """A large floating ``numpy.ma`` masked array held AS IT IS -- data + mask, nothing copied, nothing
    filled -- behind the interface of a lazily read field (``shape`` / ``dtype`` / slicing): what
    ``nc.variables["thetao"][:]`` of a netCDF4 file returns.  Its masked elements MEAN NaN (as_plain),
    but a 13 GB field is not rewritten into a NaN-filled copy to say so: the engine cuts time chunks
    out of it exactly as out of a netCDF4 variable -- every slice a masked VIEW -- and the NaN goes in
    while each piece is copied into the staging ring (hostio.split_masked -> mlx_host_copy_masked),
    in the pass over the bytes the upload makes anyway.  ``np.asarray`` / ``DataArray.values`` fill on
    demand."""

    def __init__(self, array):
        self.array = array
        self.shape = tuple(array.shape)
        self.dtype = np_dtype(array.dtype)
        self.ndim = array.ndim

    @staticmethod
    def wanted(a):
        """floating, 4 / 8 bytes wide, actually carrying a mask, and large enough for the second
        pass to matter; everything else is filled at once (small arrays feed host-side validation)"""
        return (isinstance(a, np.ma.MaskedArray) and a.dtype.kind == "f" and a.dtype.itemsize in (4, 8)
                and a.ndim >= 1 and a.nbytes >= _NATIVE_FILL_BYTES
                and np.ma.getmask(a) is not np.ma.nomask)

    def __getitem__(self, key):
        return self.array[key]  # (a masked view, or a masked scalar: as_plain takes both)

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        out = as_plain(self.array)
        return out if dtype is None else out.astype(dtype, copy=False)


class LazySource:
    """``shape`` / ``dtype`` / slicing for a sliceable source that lacks a ``dtype`` attribute of its
    own -- ``scipy.io.netcdf_file`` variables keep theirs on ``.data`` -- so that it is read slice by
    slice like any netCDF4 / h5py / zarr variable instead of being np.asarray'ed into nonsense."""

    def __init__(self, source, dtype):
        self.source = source
        self.shape = tuple(source.shape)
        self.dtype = np.dtype(dtype)
        self.ndim = len(self.shape)

    def __getitem__(self, key):
        return self.source[key]

    def __len__(self):
        return self.shape[0]


def _as_lazy_source(x):
    """x itself, or a LazySource around it when it is array-shaped and sliceable but keeps its dtype
    elsewhere (``x.data.dtype``)"""
    if (not _is_tensor(x) and not isinstance(x, (np.ndarray, np.generic, list, tuple, int, float, bool))
            and hasattr(x, "shape") and hasattr(x, "__getitem__") and not hasattr(x, "dtype")
            and hasattr(getattr(x, "data", None), "dtype")):
        return LazySource(x, x.data.dtype)
    return x


def _to_numpy(x):
    if _is_tensor(x):
        from . import hostio

        return hostio.to_host(x)
    if is_lazy(x):
        return as_plain(x[...])  # reads all of it
    return as_plain(x)


class LazyTranspose:
    """A transposed view of a lazy array that stays lazy: indexing the leading axes of the VIEW
    reads just that part of the base and transposes it in memory."""

    def __init__(self, base, perm):
        self.base, self.perm = base, tuple(perm)
        self.shape = tuple(base.shape[i] for i in self.perm)
        self.dtype = base.dtype
        self.ndim = len(self.shape)

    def __getitem__(self, key):
        if key is Ellipsis:
            key = ()
        if not isinstance(key, tuple):
            key = (key,)
        key = key + (slice(None),) * (self.ndim - len(key))
        base_key = [slice(None)] * self.ndim
        for view_axis, k in enumerate(key):
            base_key[self.perm[view_axis]] = k
        block = as_plain(self.base[tuple(base_key)])
        kept = [ax for ax, k in enumerate(key) if not isinstance(k, (int, np.integer))]
        order = sorted(range(len(kept)), key=lambda i: self.perm[kept[i]])  # base order of kept axes
        inv = [order.index(i) for i in range(len(kept))]
        return block.transpose(inv)

    def __array__(self, dtype=None, copy=None):
        out = _to_numpy(self.base).transpose(self.perm)
        return out.astype(dtype) if dtype is not None else out


class DataArray:
    """A named n-d array: data + dims + coords + attrs + encoding."""

    __array_priority__ = 50

    def __init__(self, data, dims=None, coords=None, attrs=None, name=None):
        if isinstance(data, DataArray):
            dims = data.dims if dims is None else dims
            coords = data.coords if coords is None else coords
            attrs = data.attrs if attrs is None else attrs
            data = data.data
        data = _as_lazy_source(data)  # (a scipy.io netcdf variable: no dtype attribute of its own)
        if MaskedSource.wanted(data):
            data = MaskedSource(data)  # (a big masked field: NaN-filled while it is uploaded)
        elif not _is_tensor(data) and not is_lazy(data):
            data = as_plain(data)  # (a small numpy masked array becomes NaN-filled here, once)
        ndim = len(data.shape)
        if dims is None:
            dims = tuple(f"dim_{i}" for i in range(ndim))
        if isinstance(dims, str):
            dims = (dims,)
        dims = tuple(dims)
        if len(dims) != ndim:
            raise ValueError(f"{len(dims)} dims given for a {ndim}-d array")
        self.data = data
        self.dims = dims
        self.coords = dict(coords) if coords else {}
        self.attrs = dict(attrs) if attrs else {}
        self.encoding = {}
        self.name = name

    # ---- basic properties ---------------------------------------------------------
    @property
    def values(self):
        return _to_numpy(self.data)

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def ndim(self):
        return len(self.dims)

    @property
    def dtype(self):
        # (native byte order whatever a lazy source stores: reads come back native, as_plain)
        return np_dtype(self.data.dtype)

    @property
    def is_lazy(self):
        return is_lazy(self.data)

    @property
    def sizes(self):
        return dict(zip(self.dims, self.shape))

    @property
    def is_device(self):
        return _is_tensor(self.data) and self.data.is_cuda

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        v = self.values
        return v.astype(dtype) if dtype is not None else v

    def __float__(self):
        return float(self.values)

    def __bool__(self):
        return bool(self.values)

    def __repr__(self):
        return f"<momlevel_amd.DataArray {self.name!r} {dict(self.sizes)}>"

    def item(self):
        return self.values.item()

    # ---- construction helpers -----------------------------------------------------
    def _like(self, data, dims=None, keep_attrs=False):
        dims = self.dims if dims is None else dims
        coords = {k: v for k, v in self.coords.items() if set(v.dims) <= set(dims)}
        out = DataArray(data, dims, coords, self.attrs if keep_attrs else None, self.name)
        return out

    def copy(self, deep=True):
        data = self.data
        if deep and not is_lazy(data):
            data = data.clone() if _is_tensor(data) else np.array(data)
        out = DataArray(data, self.dims, self.coords, self.attrs, self.name)
        out.encoding = dict(self.encoding)
        return out

    # ---- indexing -------------------------------------------------------------------
    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        if not isinstance(key, tuple):
            key = (key,)
        key = key + (slice(None),) * (self.ndim - len(key))
        dims = tuple(d for d, k in zip(self.dims, key) if not isinstance(k, (int, np.integer)))
        data = self.data[key]
        if is_lazy(data) and len(dims) < self.ndim:
            data = _to_numpy(data)  # a slab picked out of a lazy field (the reference state)
        out = self._like(data, dims, keep_attrs=True)
        for d, k in zip(self.dims, key):
            if d in out.coords and not isinstance(k, (int, np.integer)):
                c = self.coords[d]
                out.coords[d] = DataArray(c.data[k], (d,), None, c.attrs, d)
        return out

    def __setitem__(self, key, value):
        self.data[key] = value.data if isinstance(value, DataArray) else value

    def isel(self, indexers=None, **kw):
        indexers = dict(indexers or {}, **kw)
        key = tuple(indexers.get(d, slice(None)) for d in self.dims)
        return self[key]

    def squeeze(self):
        keep = [i for i, n in enumerate(self.shape) if n != 1]
        if len(keep) == self.ndim:
            return self
        data = self.data if not is_lazy(self.data) else _to_numpy(self.data)
        data = data.reshape([self.shape[i] for i in keep])
        return self._like(data, tuple(self.dims[i] for i in keep), keep_attrs=True)

    def reset_coords(self, drop=True):
        out = self.copy(deep=False)
        out.coords = {k: v for k, v in self.coords.items() if k in self.dims}
        return out

    def transpose(self, *dims):
        dims = list(dims)
        if Ellipsis in dims:
            i = dims.index(Ellipsis)
            rest = [d for d in self.dims if d not in dims]
            dims = dims[:i] + rest + dims[i + 1:]
        if not dims:
            dims = list(reversed(self.dims))
        if tuple(dims) == self.dims:
            return self
        perm = [self.dims.index(d) for d in dims]
        if _is_tensor(self.data):
            data = self.data.permute(*perm)
        elif is_lazy(self.data):
            data = LazyTranspose(self.data, perm)
        else:
            data = self.data.transpose(perm)
        out = self._like(data, tuple(dims), keep_attrs=True)
        out.encoding = dict(self.encoding)
        return out

    def rename(self, name):
        out = self.copy(deep=False)
        out.name = name
        return out

    # ---- reductions / element-wise (host side, validation-sized) -------------------
    def sum(self, dim=None, skipna=True):
        v = self.values
        fn = np.nansum if skipna else np.sum
        if dim is None:
            return DataArray(fn(v), (), None, None, self.name)
        dims = (dim,) if isinstance(dim, str) else tuple(dim)
        axes = tuple(self.dims.index(d) for d in dims)
        keep = tuple(d for d in self.dims if d not in dims)
        return self._like(fn(v, axis=axes), keep)

    def mean(self, dim=None):
        v = self.values
        if dim is None:
            return DataArray(np.nanmean(v), ())
        dims = (dim,) if isinstance(dim, str) else tuple(dim)
        axes = tuple(self.dims.index(d) for d in dims)
        return self._like(np.nanmean(v, axis=axes), tuple(d for d in self.dims if d not in dims))

    def notnull(self):
        return self._like(~np.isnan(self.values))

    def isnull(self):
        return self._like(np.isnan(self.values))

    def fillna(self, value):
        v = self.values
        return self._like(np.where(np.isnan(v), value, v), keep_attrs=True)

    def where(self, cond, other=np.nan):
        a, c = _align(self, cond if isinstance(cond, DataArray) else DataArray(cond, self.dims))
        return DataArray(np.where(c[0], a[0], other), a[1], self.coords, self.attrs, self.name)

    def astype(self, dtype):
        return self._like(self.values.astype(dtype), keep_attrs=True)

    def _binary(self, other, op, reflexive=False):
        if isinstance(other, DataArray):
            (a, dims), (b, _) = _align(self, other)
            coords = dict(other.coords)
            coords.update(self.coords)
        else:
            # (a python number stays one: numpy lets it take the ARRAY's dtype -- float32 z_l * 1e4
            #  is float32, steric.py:96 -- whereas np.asarray(1e4) is a float64 array and promotes)
            weak = isinstance(other, (bool, int, float)) and not isinstance(other, np.generic)
            a, b, dims, coords = self.values, other if weak else _to_numpy(other), self.dims, self.coords
        res = op(b, a) if reflexive else op(a, b)
        return DataArray(res, dims, {k: v for k, v in coords.items() if set(v.dims) <= set(dims)})

    def __add__(self, o):
        return self._binary(o, np.add)

    def __radd__(self, o):
        return self._binary(o, np.add, True)

    def __sub__(self, o):
        return self._binary(o, np.subtract)

    def __rsub__(self, o):
        return self._binary(o, np.subtract, True)

    def __mul__(self, o):
        return self._binary(o, np.multiply)

    def __rmul__(self, o):
        return self._binary(o, np.multiply, True)

    def __truediv__(self, o):
        return self._binary(o, np.divide)

    def __rtruediv__(self, o):
        return self._binary(o, np.divide, True)

    def __neg__(self):
        return self._like(-self.values)

    def __ge__(self, o):
        return self._binary(o, np.greater_equal)

    def __lt__(self, o):
        return self._binary(o, np.less)


def _align(a, b):
    """Broadcast two DataArrays by dim NAME (first-appearance order, as xarray does)."""
    dims = list(a.dims) + [d for d in b.dims if d not in a.dims]

    def expand(x):
        v = x.values
        order = [d for d in dims if d in x.dims]
        v = v.transpose([x.dims.index(d) for d in order])
        shape = [x.sizes[d] if d in x.dims else 1 for d in dims]
        return v.reshape(shape)

    return (expand(a), tuple(dims)), (expand(b), tuple(dims))


class Dataset:
    """An ordered mapping name -> DataArray sharing coordinates."""

    def __init__(self, data_vars=None, coords=None, attrs=None):
        object.__setattr__(self, "_vars", {})
        object.__setattr__(self, "_coord_names", set())
        object.__setattr__(self, "attrs", dict(attrs) if attrs else {})
        for k, v in (coords or {}).items():
            self._set(k, v, is_coord=True)
        for k, v in (data_vars or {}).items():
            self[k] = v

    # ---- mapping protocol -----------------------------------------------------------
    def _set(self, key, value, is_coord=False):
        if isinstance(value, tuple):
            dims, data = value[0], value[1]
            attrs = value[2] if len(value) > 2 else None
            value = DataArray(data, dims, None, attrs)
        elif not isinstance(value, DataArray):
            value = DataArray(value, ())
        da = DataArray(value.data, value.dims, None, value.attrs, key)
        da.encoding = dict(value.encoding)
        # adopt coordinates that ride along on the DataArray
        for cname, c in value.coords.items():
            if cname not in self._vars:
                cc = DataArray(c.data, c.dims, None, c.attrs, cname)
                self._vars[cname] = cc
                self._coord_names.add(cname)
        self._vars[key] = da
        if is_coord or (da.dims == (key,)):
            self._coord_names.add(key)

    def __setitem__(self, key, value):
        self._set(key, value)

    def __getitem__(self, key):
        if isinstance(key, (list, tuple)):
            return Dataset({k: self[k] for k in key if k not in self._coord_names},
                           {k: self._vars[k] for k in self._coord_names}, self.attrs)
        da = self._vars[key]
        out = DataArray(da.data, da.dims, self._coords_for(da.dims, exclude=key), da.attrs, key)
        out.encoding = da.encoding  # shared on purpose: result["x"].encoding["dtype"] = ... sticks
        out.attrs = da.attrs
        return out

    def _coords_for(self, dims, exclude=None):
        return {
            k: self._vars[k]
            for k in self._coord_names
            if k != exclude and set(self._vars[k].dims) <= set(dims)
        }

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __setattr__(self, key, value):
        if key == "attrs":
            object.__setattr__(self, key, value)
        else:
            raise AttributeError("assign variables with dset[name] = ...")

    def __contains__(self, key):
        return key in self._vars

    def __iter__(self):
        return iter(self.data_vars)

    def __len__(self):
        return len(self.data_vars)

    def __repr__(self):
        return f"<momlevel_amd.Dataset vars={list(self.data_vars)} coords={sorted(self._coord_names)}>"

    def keys(self):
        return self.data_vars.keys()

    @property
    def variables(self):
        return dict(self._vars)

    @property
    def data_vars(self):
        return {k: self[k] for k in self._vars if k not in self._coord_names}

    @property
    def coords(self):
        return {k: self[k] for k in self._vars if k in self._coord_names}

    @property
    def dims(self):
        out = {}
        for da in self._vars.values():
            for d, n in zip(da.dims, da.shape):
                out.setdefault(d, n)
        return out

    sizes = dims

    # ---- the few Dataset methods the steric path and its tests use -----------------
    def copy(self, deep=False):
        out = Dataset(attrs=self.attrs)
        for k, v in self._vars.items():
            vv = v.copy(deep=deep)
            vv.encoding = dict(v.encoding)
            out._vars[k] = vv
        out._coord_names.update(self._coord_names)
        return out

    def rename(self, name_dict=None):
        """Dataset.rename(varname_map); None is a no-op (steric.py:84)."""
        if not name_dict:
            return self
        out = Dataset(attrs=self.attrs)
        for k, v in self._vars.items():
            nk = name_dict.get(k, k)
            dims = tuple(name_dict.get(d, d) for d in v.dims)
            da = DataArray(v.data, dims, None, v.attrs, nk)
            da.encoding = dict(v.encoding)
            out._vars[nk] = da
            if k in self._coord_names:
                out._coord_names.add(nk)
        return out

    def drop_vars(self, names):
        names = [names] if isinstance(names, str) else list(names)
        out = self.copy()
        for n in names:
            out._vars.pop(n)
            out._coord_names.discard(n)
        return out

    def isel(self, indexers=None, **kw):
        indexers = dict(indexers or {}, **kw)
        out = Dataset(attrs=self.attrs)
        for k, v in self._vars.items():
            sub = v.isel({d: i for d, i in indexers.items() if d in v.dims})
            sub.coords = {}
            out._vars[k] = sub
            if k in self._coord_names:
                out._coord_names.add(k)
        return out

    def sum(self, dim=None):
        out = Dataset(attrs=self.attrs)
        for k, v in self.data_vars.items():
            if v.values.dtype.kind in "fiub":
                out[k] = v.sum(dim) if (dim is None or dim in v.dims) else v
        return out

    def assign_coords(self, coords):
        out = self.copy()
        for k, v in coords.items():
            out._set(k, v if isinstance(v, DataArray) else DataArray(np.asarray(v), (k,)),
                     is_coord=True)
        return out
