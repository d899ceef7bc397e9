"""CPU: the xarray <-> labelled conversion, exercised against tests/fake_xarray.py (see there)."""

import numpy as np
import pytest

import fake_xarray
from momlevel_amd import adapters
from momlevel_amd.labeled import DataArray, Dataset
from momlevel_amd.test_data import generate_test_data


@pytest.fixture
def fake_xr(monkeypatch):
    monkeypatch.setattr(adapters, "xr", fake_xarray)
    return fake_xarray


def test_dataset_round_trip(fake_xr):
    d = generate_test_data()
    d["thetao"].encoding["dtype"] = "float32"
    x = adapters.to_xarray(d)
    assert isinstance(x, fake_xr.Dataset)
    assert set(x.data_vars) == set(d.data_vars) and set(x.coords) == set(d.coords)
    assert x["thetao"].dims == ("time", "z_l", "yh", "xh")
    assert x["thetao"].attrs["units"] == "degC" and x["thetao"].encoding["dtype"] == "float32"
    assert x.coords["z_l"].attrs["edges"] == "z_i"
    back = adapters.from_xarray(x)
    assert isinstance(back, Dataset)
    for k in d.data_vars:
        assert back[k].dims == d[k].dims
        assert np.array_equal(back[k].values, d[k].values, equal_nan=True)
        assert back[k].attrs == d[k].attrs
    assert back["thetao"].encoding["dtype"] == "float32"
    assert np.shares_memory(back["so"].data, x["so"].data)  # relabelling only, no copy


def test_accepts_xarray_answers_in_kind(fake_xr):
    calls = []

    @adapters.accepts_xarray
    def f(ds, scale=1.0, other=None):
        calls.append(type(ds))
        out = Dataset()
        out["y"] = DataArray(ds["areacello"].values * scale, ds["areacello"].dims)
        return out, ds

    d = generate_test_data()
    r1, r2 = f(d, scale=2.0)  # labelled in -> labelled out, untouched
    assert isinstance(r1, Dataset) and r2 is d
    x = adapters.to_xarray(d)
    r1, r2 = f(x, scale=2.0, other=x["areacello"])
    assert calls[-1] is Dataset  # the wrapped function always sees the labelled classes
    assert isinstance(r1, fake_xr.Dataset) and isinstance(r2, fake_xr.Dataset)
    assert np.allclose(r1["y"].values, d["areacello"].values * 2.0)


def test_dataarray_round_trip(fake_xr):
    d = generate_test_data()
    x = adapters.to_xarray(d["areacello"])
    assert isinstance(x, fake_xr.DataArray) and x.dims == ("yh", "xh")
    assert set(x.coords) >= {"yh", "xh"}
    back = adapters.from_xarray(x)
    assert back.dims == ("yh", "xh") and np.array_equal(back.values, d["areacello"].values)
    assert adapters.from_xarray(3.0) == 3.0  # non-xarray objects pass through


def test_without_xarray_to_xarray_raises(monkeypatch):
    monkeypatch.setattr(adapters, "xr", None)
    assert not adapters.have_xarray()
    with pytest.raises(RuntimeError):
        adapters.to_xarray(generate_test_data())
