""" test_data - deterministic MOM6-shaped test datasets (config #1 of BASELINE.md)

Restates the reference's generators (src/momlevel/test_data/__init__.py:16-140,
tripolar/horizontal.py:11-122, tripolar/vertical.py:13-86, time.py:42-120) on the
labelled classes: the same ``numpy.random.default_rng`` draws, so the datasets are
number-for-number the ones the reference's own tests run on.
"""

import numpy as np

from .cftime_lite import monthly_midpoints
from .labeled import DataArray, Dataset

__all__ = ["generate_test_data", "generate_test_data_dz", "generate_test_data_time"]


def _time_stub(start_year, nyears, calendar):
    time = np.array(monthly_midpoints(start_year, nyears, calendar), dtype=object)
    return DataArray(
        time, ("time",), None,
        {"long_name": "time", "cartesian_axis": "T", "calendar_type": calendar,
         "bounds": "time_bnds"},
    )


def xy_fields(dset=None, seed=123):
    """5x5 h-point grid with geolon/geolat and a normalised areacello (horizontal.py)."""
    dset = Dataset() if dset is None else dset
    dset["xh"] = DataArray(
        [1.0, 2.0, 3.0, 4.0, 5.0], ("xh",), None,
        {"long_name": "h point nominal longitude", "units": "degrees_east", "axis": "X",
         "cartesian_axis": "X"},
    )
    dset["yh"] = DataArray(
        [1.0, 2.0, 3.0, 4.0, 5.0], ("yh",), None,
        {"long_name": "h point nominal latitude", "units": "degrees_north", "axis": "Y",
         "cartesian_axis": "Y"},
    )
    lon = np.arange(0.0, 361.0, 72.0)
    lat = np.arange(-90.0, 91.0, 36.0)
    lon = [(lon[x] + lon[x + 1]) / 2.0 for x in range(0, len(lon) - 1)]
    lat = [(lat[x] + lat[x + 1]) / 2.0 for x in range(0, len(lat) - 1)]
    geolon, geolat = np.meshgrid(lon, lat)
    dset["geolon"] = DataArray(geolon, ("yh", "xh"), None,
                               {"long_name": "Longitude of tracer (T) points",
                                "units": "degrees_east", "cell_methods": "time: point"})
    dset["geolat"] = DataArray(geolat, ("yh", "xh"), None,
                               {"long_name": "Latitude of tracer (T) points",
                                "units": "degrees_north", "cell_methods": "time: point"})
    areacello = np.random.default_rng(seed).normal(100.0, 10.0, (5, 5))
    areacello = areacello / areacello.sum()
    dset["areacello"] = DataArray(
        areacello * 3.6111092e14, ("yh", "xh"), None,
        {"long_name": "Ocean Grid-Cell Area", "units": "m2",
         "cell_methods": "area:sum yh:sum xh:sum time: point", "standard_name": "cell_area"},
    )
    return dset


def zlevel_fields(dset=None, include_deptho=True, seed=123):
    """5-level z grid and a matching deptho (vertical.py)."""
    dset = Dataset() if dset is None else dset
    dset["z_i"] = DataArray(
        np.array([0.0, 5.0, 15.0, 185.0, 1815.0, 6185.0]), ("z_i",), None,
        {"long_name": "Depth at interface", "units": "meters", "axis": "Z", "positive": "down"},
    )
    dset["z_l"] = DataArray(
        np.array([2.5, 10.0, 100.0, 1000.0, 4000.0]), ("z_l",), None,
        {"long_name": "Depth at cell center", "units": "meters", "axis": "Z",
         "positive": "down", "edges": "z_i"},
    )
    if include_deptho:
        deptho = np.array(
            [np.random.default_rng(seed).uniform(0.0, hi, 5)
             for hi in (5.0, 15.0, 185.0, 1815.0, 6185.0)]
        )
        if ("yh" not in dset.dims) or ("xh" not in dset.dims):
            dset = xy_fields(dset)
        dset["deptho"] = DataArray(
            deptho, ("yh", "xh"), None,
            {"long_name": "Sea Floor Depth", "units": "m",
             "cell_methods": "area:mean yh:mean xh:mean time: point",
             "cell_measures": "area: areacello",
             "standard_name": "sea_floor_depth_below_geoid"},
        )
    return dset


def generate_test_data(start_year=1981, nyears=0, calendar="noleap", seed=123):
    """ntimes x 5 x 5 x 5 dataset for unit testing (test_data/__init__.py:16-105)."""
    dset = Dataset()
    if nyears >= 1:
        dset["time"] = _time_stub(start_year, nyears, calendar)
    else:
        dset["time"] = DataArray(
            [1.0, 2.0, 3.0, 4.0, 5.0], ("time",), None,
            {"long_name": "time", "cartesian_axis": "T", "calendar_type": calendar,
             "bounds": "time_bnds"},
        )
    ntimes = len(dset["time"])
    dset = xy_fields(dset)
    dset = zlevel_fields(dset)
    dims = ("time", "z_l", "yh", "xh")
    shape = (ntimes, 5, 5, 5)
    tavg = {"time_avg_info": "average_T1,average_T2,average_DT"}
    dset["thetao"] = DataArray(
        np.random.default_rng(seed).normal(15.0, 5.0, shape), dims, None,
        dict(long_name="Sea Water Potential Temperature", units="degC",
             cell_measures="volume: volcello area: areacello",
             standard_name="sea_water_potential_temperature",
             cell_methods="area:mean z_l:mean yh:mean xh:mean time: mean", **tavg),
    )
    dset["so"] = DataArray(
        np.random.default_rng(seed).normal(35.0, 1.5, shape), dims, None,
        dict(long_name="Sea Water Salinity", units="psu",
             cell_measures="volume: volcello area: areacello",
             standard_name="sea_water_salinity",
             cell_methods="area:mean z_l:mean yh:mean xh:mean time: mean", **tavg),
    )
    dset["volcello"] = DataArray(
        np.random.default_rng(seed).normal(1000.0, 100.0, shape), dims, None,
        dict(long_name="Ocean grid-cell volume", units="m3", cell_measures="area: areacello",
             standard_name="ocean_volume",
             cell_methods="area:sum z_l:sum yh:sum xh:sum time: mean", **tavg),
    )
    return dset


def generate_test_data_dz(seed=123):
    """Partial-bottom-cell test dataset (test_data/__init__.py:108-140)."""
    xh = DataArray(np.arange(1, 6), ("xh",))
    yh = DataArray(np.arange(10, 60, 10), ("yh",))
    deptho = np.random.default_rng(seed).uniform(0.0, 100.0, (5, 5))
    deptho[2, 2] = np.nan
    deptho[2, 3] = np.nan
    z_i = np.array([0.0, 5.0, 10.0, 20.0, 50.0, 100.0])
    z_l = np.array((z_i[1::] + z_i[0:-1]) / 2.0)
    return Dataset(
        {"deptho": DataArray(deptho, ("yh", "xh"), {"yh": yh, "xh": xh}),
         "z_l": DataArray(z_l, ("z_l",)), "z_i": DataArray(z_i, ("z_i",))}
    )


def generate_test_data_time(start_year=1981, nyears=5, calendar="noleap", seed=123):
    """Monthly time-series dataset var_a/var_b (test_data/__init__.py:143-191, "MS" only)."""
    dset = Dataset()
    dset["time"] = _time_stub(start_year, nyears, calendar)
    nt = len(dset["time"])
    lon = DataArray([1.0, 2.0, 3.0, 4.0, 5.0], ("lon",))
    lat = DataArray([1.0, 2.0, 3.0, 4.0, 5.0], ("lat",))
    dset["lon"], dset["lat"] = lon, lat
    attrs = {"first_attribute": "foo", "second_attribute": "bar"}
    dset["var_a"] = DataArray(np.random.default_rng(seed).normal(100, 20, (nt, 5, 5)),
                              ("time", "lat", "lon"), None, attrs)
    dset["var_b"] = DataArray(np.random.default_rng(seed * 2).normal(100, 20, (nt, 5, 5)),
                              ("time", "lat", "lon"), None, attrs)
    return dset
