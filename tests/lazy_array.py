"""A duck-typed LAZY array for the tests: array-shaped, not numpy, readable only by slicing --
what a dask array or a netCDF4 / h5py variable looks like to momlevel_amd -- that records every
read, so a test can assert that a 4-D field is never materialised beyond one time chunk."""

import numpy as np


class CountingLazy:
    def __init__(self, array):
        self._a = array
        self.shape, self.dtype, self.ndim = array.shape, array.dtype, array.ndim
        self.reads = []  # bytes of every slice that was materialised

    def __getitem__(self, key):
        block = self._a[key]
        self.reads.append(int(np.asarray(block).nbytes))
        return np.array(block)

    def __len__(self):
        return self.shape[0]

    @property
    def largest_read(self):
        return max(self.reads) if self.reads else 0


FILL = 1e20  # netCDF's customary _FillValue of float fields


def as_masked(array, fill=FILL):
    """The ``numpy.ma.MaskedArray`` a netCDF4 read of ``array`` gives when its NaNs are stored as
    ``_FillValue``: the data hold ``fill`` where the mask is set (never NaN)."""
    array = np.asarray(array)
    mask = np.isnan(array)
    return np.ma.masked_array(np.where(mask, array.dtype.type(fill), array), mask=mask)


class MaskedLazy(CountingLazy):
    """``netCDF4.Variable``-like: every slice comes back as a masked array holding the fill value
    where the file has ``_FillValue`` (``set_auto_mask(True)``, netCDF4's default)."""

    def __getitem__(self, key):
        block = self._a[key]
        self.reads.append(int(np.asarray(block).nbytes))
        return as_masked(block)


class PrebuiltMaskedLazy(CountingLazy):
    """As MaskedLazy, but the masked array is built once: slices are VIEWS (no per-read work of the
    stand-in itself), for timing what the product does with a masked slice."""

    def __init__(self, array):
        super().__init__(array)
        self._m = as_masked(array)

    def __getitem__(self, key):
        block = self._m[key]
        self.reads.append(int(block.nbytes))
        return block


class NetCDFVar:
    """A variable of a ``scipy.io.netcdf_file`` (a REAL NetCDF-3 reader: big-endian data on disk,
    ``maskandscale=True`` -> every read is a numpy masked array honouring ``_FillValue`` /
    ``missing_value``) with the three attributes momlevel_amd asks of a lazy source: ``shape``,
    ``dtype``, slicing.  (netCDF4 / h5py / zarr variables have them; scipy's lacks ``dtype``.)"""

    def __init__(self, var):
        self._v = var
        self.shape = tuple(var.shape)
        self.dtype = var.data.dtype  # ">f4" / ">f8": the file's byte order
        self.ndim = len(self.shape)
        self.reads = []

    def __getitem__(self, key):
        block = self._v[key]
        self.reads.append(int(np.asarray(block).nbytes))
        return block

    def __len__(self):
        return self.shape[0]


def write_netcdf3(path, dset, fill=FILL):
    """``dset`` (a momlevel_amd Dataset of numpy-backed variables) as a NetCDF-3 file: NaN cells
    stored as ``_FillValue`` (what MOM6 writes for land), dims and coordinates as they are."""
    from scipy.io import netcdf_file

    with netcdf_file(path, "w") as f:
        for name in ("time", "z_l", "z_i", "yh", "xh"):
            f.createDimension(name, len(dset[name].values))
        for name in dset.variables:
            da = dset[name]
            a = np.asarray(da.values)
            code = "f" if a.dtype == np.float32 else "d"
            v = f.createVariable(name, code, tuple(da.dims))
            nan = np.isnan(a)
            v[:] = np.where(nan, a.dtype.type(fill), a)
            if nan.any():
                v._FillValue = a.dtype.type(fill)
                v.missing_value = a.dtype.type(fill)
