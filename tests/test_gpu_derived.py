"""GPU: momlevel_amd.derived -- the reference's tests/test_derived.py hot subset
(:26-51, :74-103), then oracle parity."""

import numpy as np
import pytest

from momlevel_amd import derived
from momlevel_amd.labeled import DataArray
from momlevel_amd.test_data import generate_test_data, generate_test_data_dz
from oracle import momlevel_numpy as o
from conftest import assert_bit_equal, assert_rel

pytestmark = pytest.mark.gpu

dset1 = generate_test_data()
dset2 = generate_test_data_dz()


def test_calc_dz_1(goldens):
    dz = derived.calc_dz(dset2.z_l, dset2.z_i, dset2.deptho)
    assert np.allclose(dz.sum(), goldens["calc_dz"]["default"])
    assert dz.dims == ("z_l", "yh", "xh")
    assert_bit_equal(dz.values, o.calc_dz(dset2.z_l.values, dset2.z_i.values, dset2.deptho.values))


def test_calc_dz_2(goldens):
    dz = derived.calc_dz(dset2.z_l, dset2.z_i, dset2.deptho, fraction=True)
    assert np.allclose(dz.sum(), goldens["calc_dz"]["fraction"])
    assert_bit_equal(dz.values, o.calc_dz(dset2.z_l.values, dset2.z_i.values,
                                          dset2.deptho.values, fraction=True))


def test_calc_dz_3():
    deptho = dset2.deptho.copy()
    deptho[4, 4] = -200.0
    with pytest.raises(Exception):
        derived.calc_dz(dset2.z_l, dset2.z_i, deptho)


def test_calc_dz_4(goldens):
    dz = derived.calc_dz(dset2.z_l, dset2.z_i, dset2.deptho, top=12.0, bottom=33.0)
    assert np.allclose(dz.sum(), goldens["calc_dz"]["top12_bottom33"])
    assert_bit_equal(dz.values, o.calc_dz(dset2.z_l.values, dset2.z_i.values,
                                          dset2.deptho.values, top=12.0, bottom=33.0))


def test_calc_rho(goldens):
    rho = derived.calc_rho(dset1.thetao, dset1.so, dset1.z_l * 1.0e4, eos="Wright")
    pytest.rho = rho
    assert np.allclose(rho.sum(), goldens["derived"]["calc_rho_loose"])  # the reference's bar
    assert rho.dims == ("time", "z_l", "yh", "xh")
    assert rho.attrs["comment"] == "calculated with the Wright equation of state"
    assert_bit_equal(rho.values, o.calc_rho(dset1.thetao.values, dset1.so.values,
                                            dset1.z_l.values * 1.0e4))


def test_calc_rho_on_a_lazy_field(monkeypatch):
    """a field that can only be read by slicing (dask / netCDF4 / h5py-like) is read and evaluated
    -- piece by piece along its leading axis when it is large (eos/_dispatch.py), whole when small
    or when it has to be transposed first"""
    from lazy_array import CountingLazy
    from momlevel_amd.eos import _dispatch

    ref = o.calc_rho(dset1.thetao.values, dset1.so.values, dset1.z_l.values * 1.0e4)
    lazy = CountingLazy(dset1.thetao.values)
    rho = derived.calc_rho(DataArray(lazy, dset1.thetao.dims, dict(dset1.thetao.coords)), dset1.so,
                           dset1.z_l * 1.0e4)
    assert_bit_equal(rho.values, ref)
    monkeypatch.setattr(_dispatch, "_HOST_PIPELINE_ELEMS", 100)
    monkeypatch.setattr(_dispatch, "_HOST_CHUNK_ELEMS", 2 * 125)  # two time steps a piece
    lazy = CountingLazy(dset1.thetao.values)
    rho = derived.calc_rho(DataArray(lazy, dset1.thetao.dims, dict(dset1.thetao.coords)), dset1.so,
                           dset1.z_l * 1.0e4)
    assert_bit_equal(rho.values, ref)
    assert lazy.largest_read == 2 * 125 * 8 and len(lazy.reads) == 3
    # (z, y, x, time) order in, (time, z, y, x) out of the broadcast: read whole, transposed on the host
    tl = np.ascontiguousarray(np.moveaxis(dset1.thetao.values, 0, -1))
    lazy = CountingLazy(tl)
    rho = derived.calc_rho(dset1.so, DataArray(lazy, ("z_l", "yh", "xh", "time")), dset1.z_l * 1.0e4)
    assert rho.dims == ("time", "z_l", "yh", "xh")
    assert_bit_equal(rho.values, o.calc_rho(dset1.so.values, dset1.thetao.values, dset1.z_l.values * 1.0e4))


def test_calc_rho_on_masked_arrays(monkeypatch):
    """numpy masked arrays (a netCDF4 read: 1e20 under the mask) mean NaN -- xarray has filled them
    before the reference's calc_rho (derived.py:597-639) sees anything: in memory, as the slices of
    a lazy field (small: read whole; large: piece by piece in the upload worker), as a field that
    must be transposed, and handed straight to the numpy-level EOS function."""
    from lazy_array import MaskedLazy, as_masked
    from momlevel_amd.eos import _dispatch, wright

    T = dset1.thetao.values.copy()
    S = dset1.so.values.copy()
    T[:, 2, 1, :] = np.nan
    S[:, 2, 1, :] = np.nan
    T[3, :, :, 4] = np.nan
    p = dset1.z_l.values * 1.0e4
    ref = o.calc_rho(T, S, p)
    assert np.isnan(ref).sum() == np.isnan(T).sum()
    dims, coords = dset1.thetao.dims, dict(dset1.thetao.coords)
    pres = dset1.z_l * 1.0e4
    rho = derived.calc_rho(DataArray(as_masked(T), dims, coords), DataArray(as_masked(S), dims, coords), pres)
    assert_bit_equal(rho.values, ref)
    rho = derived.calc_rho(DataArray(MaskedLazy(T), dims, coords), DataArray(as_masked(S), dims, coords), pres)
    assert_bit_equal(rho.values, ref)
    got = wright.density(as_masked(T), as_masked(S), p[:, None, None])
    assert type(got) is np.ndarray
    assert_bit_equal(got, ref)
    f32 = wright.density(as_masked(T.astype(np.float32)), as_masked(S.astype(np.float32)), p[:, None, None])
    assert_bit_equal(f32, o.calc_rho(T.astype(np.float32), S.astype(np.float32), p))
    monkeypatch.setattr(_dispatch, "_HOST_PIPELINE_ELEMS", 100)
    monkeypatch.setattr(_dispatch, "_HOST_CHUNK_ELEMS", 2 * 125)  # two time steps a piece
    lazy = MaskedLazy(T)
    rho = derived.calc_rho(DataArray(lazy, dims, coords), DataArray(as_masked(S), dims, coords), pres)
    assert_bit_equal(rho.values, ref)
    assert lazy.largest_read == 2 * 125 * 8
    tl = np.ascontiguousarray(np.moveaxis(T, 0, -1))
    rho = derived.calc_rho(DataArray(as_masked(S), dims, coords),
                           DataArray(MaskedLazy(tl), ("z_l", "yh", "xh", "time")), pres)
    assert_bit_equal(rho.values, o.calc_rho(S, T, p))
    # round 6: a LARGE in-memory masked array is held as data + mask (labeled.MaskedSource; here the
    # threshold is lowered to the test's size) and read piece by piece like the lazy field above --
    # NaN-filled by the staging copy -- also when it has to be transposed, and for float32
    from momlevel_amd import hostio, labeled

    monkeypatch.setattr(labeled, "_NATIVE_FILL_BYTES", 256)
    monkeypatch.setattr(hostio, "SMALL_BYTES", 256)
    held = DataArray(as_masked(T), dims, coords)
    assert isinstance(held.data, labeled.MaskedSource)
    rho = derived.calc_rho(held, DataArray(as_masked(S), dims, coords), pres)
    assert_bit_equal(rho.values, ref)
    rho = derived.calc_rho(DataArray(as_masked(S), dims, coords),
                           DataArray(as_masked(tl), ("z_l", "yh", "xh", "time")), pres)
    assert_bit_equal(rho.values, o.calc_rho(S, T, p))
    T32, S32 = T.astype(np.float32), S.astype(np.float32)
    held32 = DataArray(as_masked(T32), dims, coords)
    assert isinstance(held32.data, labeled.MaskedSource) and held32.dtype == np.float32
    rho = derived.calc_rho(held32, DataArray(as_masked(S32), dims, coords), pres)
    assert_bit_equal(rho.values, o.calc_rho(T32, S32, p))


def test_big_endian_fields_compute_in_their_own_precision():
    """">f4" arrays (NetCDF-3 on disk; scipy.io.netcdf_file) are float32 to numpy: the polynomial is
    evaluated in float32 (eos/wright.py:44-46 on float32 arrays), not upcast to float64 because a
    dtype test did not recognise the byte order.  Same bits as the native arrays, for the numpy-level
    EOS functions (tuned and promote kernels) and for calc_rho."""
    from momlevel_amd.eos import wright

    T = dset1.thetao.values.astype(np.float32)
    S = dset1.so.values.astype(np.float32)
    p = dset1.z_l.values[:, None, None] * 1.0e4
    for func in (wright.density, wright.alpha, wright.beta, wright.drho_dtemp):
        want = func(T, S, p)
        got = func(T.astype(">f4"), S.astype(">f4"), p.astype(">f8"))
        assert got.dtype == want.dtype
        assert_bit_equal(got, want, func.__name__)
    want = wright.density(T, S, 2.0e5)  # python-float pressure: float32 throughout (the promote kernel)
    got = wright.density(T.astype(">f4"), S.astype(">f4"), 2.0e5)
    assert got.dtype == want.dtype == np.float32
    assert_bit_equal(got, want)
    assert_bit_equal(want, o.wright_density(T, S, 2.0e5))
    dims, coords = dset1.thetao.dims, dict(dset1.thetao.coords)
    rho = derived.calc_rho(DataArray(T.astype(">f4"), dims, coords), DataArray(S.astype(">f4"), dims, coords),
                           dset1.z_l * 1.0e4)
    assert_bit_equal(rho.values, o.calc_rho(T, S, dset1.z_l.values * 1.0e4))


def test_calc_rho_held_field_broadcast_order():
    """halosteric's call: thetao (z,y,x), so (t,z,y,x) -> dims in first-appearance order."""
    rho = derived.calc_rho(dset1.thetao.isel(time=0), dset1.so, dset1.z_l * 1.0e4)
    assert rho.dims == ("z_l", "yh", "xh", "time")
    ref = o.calc_rho(dset1.thetao.values[0], dset1.so.values, dset1.z_l.values * 1.0e4)
    assert_bit_equal(rho.transpose("time", ...).values, ref)


def test_calc_alpha_beta(goldens):
    a = derived.calc_alpha(dset1.thetao, dset1.so, dset1.z_l * 1.0e4, eos="Wright")
    b = derived.calc_beta(dset1.thetao, dset1.so, dset1.z_l * 1.0e4, eos="Wright")
    assert np.allclose(a.sum(), goldens["derived"]["calc_alpha"])
    assert np.allclose(b.sum(), goldens["derived"]["calc_beta"])
    p3 = (dset1.z_l.values * 1.0e4)[:, None, None]
    assert_bit_equal(a.values, o.wright_alpha(dset1.thetao.values, dset1.so.values, p3))
    assert_bit_equal(b.values, o.wright_beta(dset1.thetao.values, dset1.so.values, p3))


def test_calc_masso(goldens):
    rho = derived.calc_rho(dset1.thetao, dset1.so, dset1.z_l * 1.0e4, eos="Wright")
    masso = derived.calc_masso(rho, dset1.volcello)
    pytest.masso = masso
    assert masso.dims == ("time",)
    assert np.allclose(masso.sum(), goldens["derived"]["calc_masso"])
    assert_rel(masso.values, o.calc_masso(rho.values, dset1.volcello.values), 1e-10)
    m3 = derived.calc_masso(rho.isel(time=0), dset1.volcello.isel(time=0))
    assert m3.dims == ()
    assert_rel(m3.values, o.calc_masso(rho.values[0], dset1.volcello.values[0]), 1e-10)
    mb = derived.calc_masso(rho, dset1.volcello.isel(time=0))  # 3-D volcello broadcast over time
    assert_rel(mb.values, o.calc_masso(rho.values, dset1.volcello.values[0]), 1e-10)


def test_calc_volo_1():
    with pytest.raises(Exception):
        _ = derived.calc_volo(dset1.volcello)


def test_calc_volo_2(goldens):
    volo = derived.calc_volo(dset1.volcello.isel(time=0))
    pytest.volo = volo
    assert np.allclose(volo, goldens["derived"]["calc_volo"])
    v = dset1.volcello.isel(time=0).values.copy()
    v[0, 0, :3] = np.nan
    assert_rel(derived.calc_volo(DataArray(v, ("z_l", "yh", "xh"))).values, np.nansum(v), 1e-13)


def test_rhoga(goldens):
    rho = derived.calc_rho(dset1.thetao, dset1.so, dset1.z_l * 1.0e4, eos="Wright")
    masso = derived.calc_masso(rho, dset1.volcello)
    volo = derived.calc_volo(dset1.volcello.isel(time=0))
    rhoga = derived.calc_rhoga(masso, volo)
    assert np.allclose(rhoga.sum(), goldens["derived"]["rhoga_sum"])
    assert rhoga.attrs["units"] == "kg m-3"


def test_calc_pdens(goldens):
    g = goldens["next_consumers"]
    r0 = derived.calc_pdens(dset1.thetao, dset1.so, eos="Wright")
    r2 = derived.calc_pdens(dset1.thetao, dset1.so, level=2000.0, eos="Wright")
    assert np.allclose(r0.sum(), g["calc_pdens_level0_loose"])      # the reference's own bar
    assert np.allclose(r2.sum(), g["calc_pdens_level2000_loose"])
    assert_bit_equal(r0.values, o.calc_pdens(dset1.thetao.values, dset1.so.values))
    assert_bit_equal(r2.values, o.calc_pdens(dset1.thetao.values, dset1.so.values, level=2000.0))
    assert r2.attrs["long_name"] == "Sea water potential density referenced to 2000.0 m"
    with pytest.raises(AssertionError):
        derived.calc_pdens(dset1.thetao, dset1.so, level=8000.0)


def test_inverse_barometer(goldens):
    from momlevel_amd import inverse_barometer

    d = generate_test_data().isel(z_l=0)
    result = inverse_barometer(d.thetao, d.so, 101325.0)
    assert np.allclose(result.sum(), goldens["next_consumers"]["inverse_barometer"])
    assert result.dims == ("time", "yh", "xh") and result.name == "ibh"
    assert_bit_equal(result.values, o.inverse_barometer(d.thetao.values, d.so.values, 101325.0))
    pso = DataArray(101325.0 + np.random.default_rng(5).normal(0, 800.0, (5, 5)), ("yh", "xh"))
    r2 = inverse_barometer(d.thetao, d.so, pso, gravity=9.81)
    assert_bit_equal(r2.values, o.inverse_barometer(d.thetao.values, d.so.values, pso.values[None],
                                                    gravity=9.81))
