"""GPU: the consumers of alpha / beta -- derived.calc_n2, adjust_negative_n2,
calc_stability_angle, calc_wave_speed (csrc/momlevel_strat.hip) -- against the reference's own
goldens (tests/test_derived.py:15-18, :54-61, :140-151) and, value for value, against the numpy
oracle (which calls numpy.gradient as xarray's differentiate does).  N^2, its adjustment and the
wave speed are bit-identical to numpy; the stability angle differs by the arctan's last bits
(numpy's libm vs the device's): 1e-12 of a right angle here, north_star's gate is 1e-10."""

import numpy as np
import pytest
import torch

from momlevel_amd import core, derived
from momlevel_amd.labeled import DataArray
from momlevel_amd.test_data import generate_test_data, generate_test_data_dz
from oracle import momlevel_numpy as o
from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu

dset1 = generate_test_data()


def test_reference_goldens(goldens):
    """The four values the reference's tests hold, through the reference's signatures."""
    g = goldens["stratification"]
    n2 = derived.calc_n2(dset1.thetao, dset1.so)
    assert n2.dims == ("time", "z_l", "yh", "xh")
    assert n2.attrs["standard_name"] == "square_of_brunt_vaisala_frequency_in_sea_water"
    assert np.allclose(n2.sum(), g["calc_n2_sum"])  # the reference's own bar
    ref = o.calc_n2(dset1.thetao.values, dset1.so.values, dset1.z_l.values)
    assert_bit_equal(n2.values, ref, "calc_n2")

    adjusted = derived.adjust_negative_n2(n2)
    assert np.allclose(adjusted.sum(), g["adjust_negative_n2_sum"])
    assert adjusted.attrs["comment"] == "adjustment applied for negative values"
    assert adjusted.attrs["units"] == "s-2"
    assert_bit_equal(adjusted.values, o.adjust_negative_n2(ref), "adjust_negative_n2")
    both = derived.calc_n2(dset1.thetao, dset1.so, adjust_negative=True)
    assert np.allclose(both.sum(), g["adjust_negative_n2_sum"])
    assert_bit_equal(both.values, adjusted.values)

    tu = derived.calc_stability_angle(dset1.thetao, dset1.so, dset1.z_l * 1.0e4, eos="Wright")
    assert tu.name == "tu_angle" and tu.attrs["units"] == "degrees"
    assert np.allclose(tu.sum(), g["calc_stability_angle_sum"])
    tu_ref = o.calc_stability_angle(dset1.thetao.values, dset1.so.values,
                                    dset1.z_l.values * 1.0e4, dset1.z_l.values)
    assert np.array_equal(np.isnan(tu.values), np.isnan(tu_ref))
    assert np.nanmax(np.abs(tu.values - tu_ref)) <= 90.0 * 1e-12

    dz = derived.calc_dz(dset1.z_l, dset1.z_i, dset1.deptho)
    speed = derived.calc_wave_speed(n2, dz)
    assert speed.dims == ("z_l", "yh", "xh", "time")  # xarray's broadcast of n2[0]: see the docstring
    assert np.allclose(speed.sum(), g["calc_wave_speed_sum"])
    dz_ref = o.calc_dz(dset1.z_l.values, dset1.z_i.values, dset1.deptho.values)
    assert_bit_equal(speed.values, o.calc_wave_speed_4d_quirk(ref, dz_ref), "calc_wave_speed")
    # one time level: n2[0] is the surface, the result a (yh, xh) map
    one = derived.calc_wave_speed(n2.isel(time=0), dz)
    assert one.dims == ("yh", "xh")
    assert_bit_equal(one.values, o.calc_wave_speed(ref[0], dz_ref))


BLK_Z = np.cumsum(2.0 * 1.075 ** np.arange(7)) * 40.0  # the levels behind wright_vectors' blk_p


@pytest.mark.parametrize("prec", ["blk", "f32"])
def test_n2_from_the_reference_modules_own_alpha_and_beta(wright_vectors, prec):
    """derived.py:396-401 evaluated with alpha and beta AS THE REFERENCE'S eos/wright.py RETURNED
    THEM (tests/golden/wright_vectors.npz: outputs of the reference module on these fields at
    p = z_l * 1e4 + 101325) and numpy.gradient: calc_n2 must give those bits -- float64 fields and
    float32 fields (numpy's mixed precision: float64 alpha / beta, float32 derivatives)."""
    v = wright_vectors
    T, S = v[f"{prec}_T"], v[f"{prec}_S"]
    assert np.array_equal(v["blk_p"].reshape(-1), BLK_Z * 1.0e4 + 101325.0)
    dtdz = np.gradient(T, BLK_Z, axis=1, edge_order=2)
    dsdz = np.gradient(S, BLK_Z, axis=1, edge_order=2)
    assert dtdz.dtype == T.dtype
    want = -9.8 * ((v[f"{prec}_alpha"] * dtdz) - (v[f"{prec}_beta"] * dsdz))
    dims = ("time", "z_l", "yh", "xh")
    coords = {"z_l": DataArray(BLK_Z, ("z_l",))}
    got = derived.calc_n2(DataArray(T, dims, coords), DataArray(S, dims, coords))
    assert_bit_equal(got.values, want, "calc_n2 vs the reference module's alpha / beta")
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (v[f"{prec}_beta"] * dsdz) / (v[f"{prec}_alpha"] * dtdz)
        tu_want = np.degrees(np.arctan((1 + r) / (1 - r)))
    tu = derived.calc_stability_angle(DataArray(T, dims, coords), DataArray(S, dims, coords),
                                      DataArray(v["blk_p"].reshape(-1), ("z_l",)))
    assert np.nanmax(np.abs(tu.values - tu_want)) <= 90.0 * 1e-12


def _fields(shape, dtype, seed, land=True):
    r = np.random.default_rng(seed)
    T = r.uniform(-2.0, 30.0, shape).astype(dtype)
    S = r.uniform(30.0, 40.0, shape).astype(dtype)
    if land:  # whole columns of NaN and sub-bottom NaN, as a masked ocean field has
        col = r.random(shape[-2:]) < 0.2
        T[..., col] = np.nan
        S[..., col] = np.nan
        deep = r.random(shape[-2:]) < 0.3
        T[..., shape[-3] // 2:, :, :][..., deep] = np.nan
        S[..., shape[-3] // 2:, :, :][..., deep] = np.nan
    return T, S


LEVELS = {
    "mom6_like": lambda nz: np.cumsum(2.0 * 1.075 ** np.arange(nz)) - 1.0,  # uneven
    "uniform": lambda nz: 5.0 + 10.0 * np.arange(nz),                     # numpy's central branch
    "uniform_int": lambda nz: np.arange(nz, dtype=np.int64) * 3,           # integer coordinate
}


@pytest.mark.parametrize("levels", sorted(LEVELS))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(3, 7, 6, 10), (2, 3, 5, 7), (9, 4, 8), (1, 12, 3, 64)])
def test_n2_bit_identical_to_numpy(shape, dtype, levels):
    """Every value of N^2 is numpy's: uneven and even level spacing (numpy.gradient's two
    branches), float64 and float32 fields (a float32 field's derivative is float32, alpha and beta
    float64), land / sub-bottom NaN, odd planes (the one-cell-per-thread kernel) and 16-byte ones,
    3-D and 4-D layouts; the adjustment and the wave speed on top."""
    nz = shape[-3]
    z = LEVELS[levels](nz)
    T, S = _fields(shape, dtype, seed=sum(shape) + nz)
    dims = ("time", "z_l", "yh", "xh")[-len(shape):]
    coords = {"z_l": DataArray(z, ("z_l",))}
    n2 = derived.calc_n2(DataArray(T, dims, coords), DataArray(S, dims, coords))
    ref = o.calc_n2(T, S, z)
    assert n2.values.dtype == np.float64 and ref.dtype == np.float64
    assert_bit_equal(n2.values, ref, "calc_n2")
    assert_bit_equal(derived.adjust_negative_n2(n2).values, o.adjust_negative_n2(ref))
    tu = derived.calc_stability_angle(DataArray(T, dims, coords), DataArray(S, dims, coords),
                                      DataArray(z * 1.0e4 + 101325.0, ("z_l",)))
    tu_ref = o.calc_stability_angle(T, S, z * 1.0e4 + 101325.0, z)
    assert np.array_equal(np.isnan(tu.values), np.isnan(tu_ref))
    assert np.nanmax(np.abs(tu.values - tu_ref)) <= 90.0 * 1e-12
    dz = np.abs(np.random.default_rng(3).normal(10.0, 3.0, shape[-3:]))
    dz[..., np.isnan(T.reshape((-1,) + shape[-3:])[0, 0])] = np.nan
    speed = derived.calc_wave_speed(n2, DataArray(dz, dims[-3:]))
    if len(shape) == 3:
        assert_bit_equal(speed.values, o.calc_wave_speed(ref, dz))
    else:
        assert_bit_equal(speed.values, o.calc_wave_speed_4d_quirk(ref, dz))


def test_pressure_operands_and_device_inputs():
    """calc_stability_angle's pressure by name: a z profile, a scalar DataArray, a (z,y,x) and a
    (time,z,y,x) field; device tensors in -> device tensor out; the linear EOS at float64."""
    shape = (3, 6, 4, 8)
    T, S = _fields(shape, np.float64, 5)
    z = LEVELS["mom6_like"](6)
    dims = ("time", "z_l", "yh", "xh")
    coords = {"z_l": DataArray(z, ("z_l",))}
    Td, Sd = DataArray(T, dims, coords), DataArray(S, dims, coords)
    r = np.random.default_rng(8)
    for pdims, pshape in ((("z_l",), (6,)), ((), ()), (("z_l", "yh", "xh"), shape[1:]), (dims, shape),
                          (("yh", "xh"), shape[2:])):
        p = np.asarray(r.uniform(1e5, 5e7, pshape))
        got = derived.calc_stability_angle(Td, Sd, DataArray(p, pdims))
        full = p if p.ndim in (0, 4) else (p[:, None, None] if pdims == ("z_l",) else p)
        ref = o.calc_stability_angle(T, S, np.broadcast_to(full, shape), z)
        assert np.nanmax(np.abs(got.values - ref)) <= 90.0 * 1e-12, pdims
    dev = DataArray(torch.from_numpy(T).cuda(), dims, coords), DataArray(torch.from_numpy(S).cuda(), dims, coords)
    n2 = derived.calc_n2(*dev)
    assert isinstance(n2.data, torch.Tensor) and n2.data.is_cuda
    assert_bit_equal(n2.data.cpu().numpy(), o.calc_n2(T, S, z))
    lin = derived.calc_n2(Td, Sd, eos="linear")
    assert_bit_equal(lin.values, o.calc_n2(T, S, z, eos="linear"))
    # gravity and patm are the reference's keyword arguments
    alt = derived.calc_n2(Td, Sd, gravity=-9.81, patm=0.0)
    assert_bit_equal(alt.values, o.calc_n2(T, S, z, gravity=-9.81, patm=0.0))


def test_large_host_fields_are_pipelined(monkeypatch):
    """Host fields above the pipeline limit go through groups of time steps whose uploads, kernels
    and result downloads overlap (hostio.Uploader / Downloader): same bits, whatever the group
    size; a pressure that varies with time is sliced with its rows."""
    from momlevel_amd import hostio

    monkeypatch.setattr(derived, "_HOST_PIPELINE_ELEMS", 1000)
    monkeypatch.setattr(derived, "_HOST_GROUP_ELEMS", 2 * 6 * 40 * 50)  # two time steps a group
    used = []
    real = hostio._enqueue_download
    monkeypatch.setattr(hostio, "_enqueue_download",
                        lambda out, dev, *a: used.append(out.shape) or real(out, dev, *a))
    shape = (5, 6, 40, 50)
    z = LEVELS["mom6_like"](6)
    dims = ("time", "z_l", "yh", "xh")
    coords = {"z_l": DataArray(z, ("z_l",))}
    for dtype in (np.float64, np.float32):
        T, S = _fields(shape, dtype, 21)
        used.clear()
        n2 = derived.calc_n2(DataArray(T, dims, coords), DataArray(S, dims, coords))
        assert [u[0] for u in used] == [2, 2, 1] and isinstance(n2.data, np.ndarray)
        assert_bit_equal(n2.values, o.calc_n2(T, S, z))
    # a lazy field (dask / netCDF4-like: readable by slicing only) is read group by group
    from lazy_array import CountingLazy

    T, S = _fields(shape, np.float32, 23)
    lazy = CountingLazy(T)
    n2 = derived.calc_n2(DataArray(lazy, dims, coords), DataArray(S, dims, coords))
    assert lazy.reads and lazy.largest_read == 2 * T[0].nbytes  # never more than one group
    assert_bit_equal(n2.values, o.calc_n2(T, S, z))
    T, S = _fields(shape, np.float64, 22)
    p4 = np.random.default_rng(4).uniform(1e5, 5e7, shape)
    tu = derived.calc_stability_angle(DataArray(T, dims, coords), DataArray(S, dims, coords),
                                      DataArray(p4, dims))
    ref = o.calc_stability_angle(T, S, p4, z)
    assert np.nanmax(np.abs(tu.values - ref)) <= 90.0 * 1e-12


def test_xarray_objects_in_xarray_objects_out(monkeypatch):
    """With xarray objects (tests/fake_xarray.py stands in: xarray is not installable here) the
    functions answer in kind, coordinates and attributes included."""
    import fake_xarray
    from momlevel_amd import adapters

    monkeypatch.setattr(adapters, "xr", fake_xarray)
    x = adapters.to_xarray(dset1)
    n2 = derived.calc_n2(x["thetao"], x["so"])
    assert isinstance(n2, fake_xarray.DataArray) and n2.dims == ("time", "z_l", "yh", "xh")
    assert n2.attrs["units"] == "s-2" and "z_l" in n2.coords
    ref = o.calc_n2(dset1.thetao.values, dset1.so.values, dset1.z_l.values)
    assert_bit_equal(np.asarray(n2.values), ref)
    adj = derived.adjust_negative_n2(n2)
    assert isinstance(adj, fake_xarray.DataArray)
    assert_bit_equal(np.asarray(adj.values), o.adjust_negative_n2(ref))
    pres = fake_xarray.DataArray(np.asarray(x["z_l"].values) * 1.0e4, dims=("z_l",))  # (the stand-in
    # has no arithmetic of its own)
    tu = derived.calc_stability_angle(x["thetao"], x["so"], pres)
    assert isinstance(tu, fake_xarray.DataArray) and tu.attrs["units"] == "degrees"


def test_float32_upcast_mode_matches_float64_arithmetic():
    T, S = _fields((2, 5, 4, 8), np.float32, 9)
    z = LEVELS["mom6_like"](5)
    p = z * 1.0e4 + 101325.0
    Td, Sd = torch.from_numpy(T).cuda().reshape(2, 5, 32), torch.from_numpy(S).cuda().reshape(2, 5, 32)
    got = core.stratification(Td, Sd, torch.from_numpy(p).cuda(), z, f32_mode="upcast")
    ref = o.calc_n2(T.astype(np.float64), S.astype(np.float64), z)
    assert_bit_equal(got.cpu().numpy().reshape(T.shape), ref)


def test_errors():
    z = LEVELS["mom6_like"](5)
    T, S = _fields((2, 5, 4, 4), np.float64, 1, land=False)
    dims = ("time", "z_l", "yh", "xh")
    coords = {"z_l": DataArray(z, ("z_l",))}
    Td, Sd = DataArray(T, dims, coords), DataArray(S, dims, coords)
    with pytest.raises(NotImplementedError):
        derived.calc_n2(Td, Sd, interfaces=DataArray(np.arange(6.0), ("z_i",)))
    with pytest.raises(ValueError):  # the reference: unknown EOS
        derived.calc_n2(Td, Sd, eos="teos10")
    with pytest.raises(ValueError):  # numpy.gradient: at least edge_order + 1 levels
        short = {"z_l": DataArray(z[:2], ("z_l",))}
        derived.calc_n2(DataArray(T[:, :2], dims, short), DataArray(S[:, :2], dims, short))
    with pytest.raises(ValueError):
        derived.calc_n2(Td, DataArray(S[0], dims[1:], coords))
    with pytest.raises(ValueError):
        derived.calc_n2(DataArray(T, dims), DataArray(S, dims))  # no level values
    T32, S32 = DataArray(T.astype(np.float32), dims, coords), DataArray(S.astype(np.float32), dims, coords)
    with pytest.raises(TypeError):  # numpy would evaluate alpha in float32 against a python float
        derived.calc_stability_angle(T32, S32, 2.0e5)
    with pytest.raises(TypeError):
        derived.calc_n2(T32, S32, eos="linear")
    with pytest.raises(TypeError):
        derived.calc_n2(T32, Sd)
    # a float32 coordinate with float32 fields: numpy differentiates (and calc_n2 builds its
    # pressure) in float32 -- refused rather than answered in other bits
    z32 = {"z_l": DataArray(z.astype(np.float32), ("z_l",))}
    with pytest.raises(TypeError, match="float32"):
        derived.calc_n2(DataArray(T.astype(np.float32), dims, z32), DataArray(S.astype(np.float32), dims, z32))
    derived.calc_n2(DataArray(T, dims, z32), DataArray(S, dims, z32))  # float64 fields: any coordinate
    n2 = derived.calc_n2(Td, Sd)
    with pytest.raises(ValueError):
        derived.calc_wave_speed(n2, DataArray(np.ones((5, 4)), ("z_l", "yh")))


def test_c_abi_argument_checks():
    from momlevel_amd import _lib

    lib = _lib.load()
    t = torch.zeros(2 * 5 * 8, dtype=torch.float64, device="cuda")
    coef = torch.zeros(15, dtype=torch.float64, device="cuda")
    a = t.data_ptr()
    base = [a, a, _lib.DTYPE_F64, a, 0, 1, 0, 0, _lib.STRAT_N2, coef.data_ptr(), 0, 0.0, -9.8, 2, 5, 8, a, None]

    def call(**kw):
        args = list(base)
        for k, v in kw.items():
            args[int(k[1:])] = v
        return lib.mlx_stratification(*args)

    assert call(_0=None) == -1 and call(_9=None) == -1 and call(_16=None) == -1
    assert call(_3=None) == -1  # Wright needs a pressure
    assert call(_3=None, _7=1) != -1  # the linear EOS does not
    assert call(_14=2) == -2 and "nz must be >= 3" in _lib.last_error()
    assert call(_8=7) == -3 and call(_7=9) == -3 and call(_2=_lib.DTYPE_T32_S64) == -3
    assert call(_2=_lib.DTYPE_F32, _7=1) == -3  # linear EOS on float32 fields: not built
    assert call(_6=2) == -2 and call(_10=1, _11=0.0) == -2
    assert call(_0=a + 4) == -5
    assert lib.mlx_adjust_negative_n2(a, 2, 5, 8, 0, None, a, None, None) == -2  # lead0_rows 0 needs nt 1
    assert lib.mlx_adjust_negative_n2(a, 2, 5, 8, 1, None, None, None, None) == -1
    assert lib.mlx_adjust_negative_n2(a, 2, 5, 8, 1, None, None, a, None) == -1  # speed needs dz
    assert lib.mlx_wave_speed_where_time0(None, a, 2, 5, 8, a, None) == -1
    torch.cuda.synchronize()
