"""CPU: host-side mirror of the reference interface (no kernels are launched)."""

import numpy as np
import pytest

import momlevel_amd as m
from momlevel_amd import synthetic, util
from momlevel_amd.labeled import DataArray, Dataset
from momlevel_amd.test_data import (
    generate_test_data,
    generate_test_data_dz,
    generate_test_data_time,
)
from oracle import momlevel_numpy as o

dset = generate_test_data()


# ---- reference tests/test_util.py:47-121 ------------------------------------------------
def test_default_coords_1():
    assert util.default_coords() == ("time", "z_l", "z_i")


def test_default_coords_2():
    assert util.default_coords(coord_names={"z": "lev", "t": "TIME"}) == ("TIME", "lev", "z_i")


def test_validate_areacello_1():
    assert util.validate_areacello(dset.areacello)


def test_validate_areacello_2():
    assert not util.validate_areacello(dset.areacello * 1.3)


def test_validate_dataset_1():
    util.validate_dataset(dset)


def test_validate_dataset_2():
    with pytest.raises(ValueError):
        util.validate_dataset(dset.copy().drop_vars(["thetao"]))


def test_validate_dataset_3():
    t = dset.copy()
    t["areacello"] = t["areacello"] * 1.3
    with pytest.raises(ValueError):
        util.validate_dataset(t)


def test_validate_dataset_4():
    t = dset.copy()
    t["areacello"] = t["areacello"] * 1.3
    with pytest.warns(UserWarning):
        util.validate_dataset(t, strict=False)


def test_validate_dataset_5():
    with pytest.raises(ValueError):
        util.validate_dataset(dset.copy(), reference=True)


def test_validate_dataset_8():
    with pytest.raises(ValueError):
        util.validate_dataset(dset.copy(), additional_vars=["foo", "bar"])


def test_eos_func_from_str_errors():
    """unknown EOS -> ValueError (util.py:247); a function the EOS module lacks -> ValueError
    naming it, not the bare KeyError of the reference's __dict__ lookup"""
    with pytest.raises(ValueError, match="Unknown equation of state: teos10"):
        util.eos_func_from_str("TEOS10")
    with pytest.raises(ValueError, match="Unknown equation of state"):
        util.eos_func_from_str("_dispatch")
    with pytest.raises(ValueError, match="linear.*speed_of_sound"):
        util.eos_func_from_str("linear", func_name="speed_of_sound")
    assert util.eos_func_from_str("linear", func_name="alpha") is m.eos.linear.alpha
    assert m.eos.linear.drho_dtemp() == -0.2 and m.eos.linear.drho_dsal(1.0, 2.0, 3.0) == 0.8
    with pytest.raises(AssertionError):
        util.eos_func_from_str(3)
    with pytest.raises(AssertionError):
        util.default_coords(["time"])


def test_validate_dataset_reports_every_finding(capsys):
    """all findings are printed, one ValueError raised (util.py:808-814); messages as the reference"""
    t = dset.copy().drop_vars(["so"])
    t["areacello"] = t["areacello"] * 1.3
    t["thetao"] = t["thetao"].isel({"time": 0})
    with pytest.raises(ValueError, match="Errors found in dataset."):
        util.validate_dataset(t, additional_vars="deptho_missing")
    out = capsys.readouterr().out.strip().splitlines()
    assert out == [
        "Reference dataset is missing variables: ['so', 'deptho_missing']",
        "Variable thetao must have exactly 4 dimensions t,z,y,x",
        "Variable `areacello` field is out of range. It may not be masked.",
    ]
    ref = dset.copy()
    for k in ("thetao", "so", "volcello"):
        ref[k] = ref[k].isel({"time": 0})
    ref["rho"] = dset["thetao"]
    for k in ("volo", "masso"):
        ref[k] = ref["areacello"].sum()
    ref["rhoga"] = dset["areacello"]
    with pytest.raises(ValueError):
        util.validate_dataset(ref, reference=True)
    out = capsys.readouterr().out.strip().splitlines()
    assert out == ["Variable areacello must have exactly 3 dimensions (z,y,x)",
                   "Variable rhoga must be a scalar"]


def test_eos_func_from_str():
    assert util.eos_func_from_str("Wright") is m.eos.wright.density
    assert util.eos_func_from_str("wright", func_name="alpha") is m.eos.wright.alpha
    assert util.eos_func_from_str("LINEAR") is m.eos.linear.density
    with pytest.raises(ValueError):
        util.eos_func_from_str("teos10")
    with pytest.raises(AssertionError):
        util.eos_func_from_str(10)


# ---- generators reproduce the reference's datasets (via the oracle's restatement) -------
@pytest.mark.parametrize("seed", [123, 999])
def test_generate_test_data_matches_oracle(seed):
    a, b = generate_test_data(seed=seed), o.generate_test_data(seed=seed)
    for k in ("thetao", "so", "volcello", "areacello", "deptho", "z_i", "z_l"):
        assert np.array_equal(a[k].values, b[k]), k
    assert a["thetao"].dims == ("time", "z_l", "yh", "xh")
    assert np.isclose(a["areacello"].values.sum(), 3.6111092e14)


def test_generate_test_data_monthly_axis():
    a = generate_test_data(start_year=1983, nyears=2, calendar="julian")
    b = o.generate_test_data(start_year=1983, nyears=2, calendar="julian")
    assert len(a["time"]) == 24
    t = a["time"].values
    assert [x.daysinmonth for x in t] == list(b["time_days_in_month"])
    assert t[13].year == 1984 and t[13].month == 2 and t[13].daysinmonth == 29
    assert np.array_equal(a["thetao"].values, b["thetao"])


def test_generate_test_data_dz_matches_oracle():
    a, b = generate_test_data_dz(), o.generate_test_data_dz()
    assert np.array_equal(np.isnan(a["deptho"].values), np.isnan(b["deptho"]))
    assert np.allclose(np.nan_to_num(a["deptho"].values), np.nan_to_num(b["deptho"]))


# ---- annual_average: reference tests/test_util.py:125-148 ------------------------------
def test_annual_average(goldens):
    g = goldens["annual_average"]
    for cal in ("noleap", "julian"):
        d = generate_test_data_time(calendar=cal)
        res = util.annual_average(d)
        assert len(res["time"]) == 5
        s = res.sum()
        assert np.allclose(s["var_a"], g[f"{cal}_var_a"])
        assert np.allclose(s["var_b"], g[f"{cal}_var_b"])
        assert np.allclose(util.annual_average(d["var_a"]).sum(), g[f"{cal}_var_a"])
        assert res["var_a"].attrs == {"first_attribute": "foo", "second_attribute": "bar"}


def test_annual_average_requires_12_steps():
    d = generate_test_data_time(nyears=1)
    short = Dataset({"var_a": d["var_a"][0:11]}, {"time": d["time"][0:11]})
    with pytest.raises(AssertionError):
        util.annual_average(short)


# ---- labelled containers ------------------------------------------------------------------
def test_labelled_basics():
    da = dset["thetao"]
    assert da.dims == ("time", "z_l", "yh", "xh") and da.shape == (5, 5, 5, 5)
    assert float(da[0, 1, 2, 3]) == da.values[0, 1, 2, 3]
    sub = da.isel(time=0)
    assert sub.dims == ("z_l", "yh", "xh") and "time" not in sub.coords
    tr = da.transpose("xh", ...)
    assert tr.dims == ("xh", "time", "z_l", "yh")
    assert np.array_equal(tr.values, np.moveaxis(da.values, 3, 0))
    pres = dset["z_l"] * 1.0e4 + 101325.0
    assert pres.dims == ("z_l",) and np.allclose(pres.values, o.pressure_from_depth(dset["z_l"].values))
    prod = dset["z_l"] * dset["areacello"]  # broadcast by NAME
    assert prod.dims == ("z_l", "yh", "xh")
    ren = dset.rename({"thetao": "temp", "z_l": "lev"})
    assert ren["temp"].dims == ("time", "lev", "yh", "xh") and "thetao" not in ren
    assert dset.rename(None) is dset
    res = Dataset()
    res["x"] = DataArray(np.arange(3.0), ("time",))
    res["x"].encoding["dtype"] = "float32"
    assert res["x"].encoding["dtype"] == "float32"
    assert float(dset.sum()["thetao"]) == pytest.approx(np.sum(dset["thetao"].values))


# ---- synthetic grids -------------------------------------------------------------------------
def test_synthetic_grid_is_a_valid_momlevel_input():
    g = synthetic.make_grid(48, 64, 15)
    assert util.validate_areacello(DataArray(g["areacello"], ("yh", "xh")))
    land = np.isnan(g["deptho"])
    assert 0.1 < land.mean() < 0.5
    assert np.isnan(g["volcello"][:, land]).all()
    assert not np.isnan(g["volcello"][0, ~land]).any()  # every ocean column has a wet top cell
    # volcello = areacello * calc_dz, NaN below the bottom
    dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"])
    wet = ~np.isnan(g["volcello"])
    assert np.allclose(g["volcello"][wet], (g["areacello"][None] * dz)[wet])
    assert (dz[~wet & ~land[None]] == 0).all()


def test_synthetic_field_tiles_reproduce_global_field():
    shape = (3, 4, 8, 12)
    kw = dict(seed=synthetic.SEED, field_id=1, lo=-2.0, scale=34.0)
    full = synthetic.field_numpy(shape, **kw)
    assert full.min() >= -2.0 and full.max() < 32.0
    for rank in range(4):
        y0, y1, x0, x1 = synthetic.tile_bounds(8, 12, rank, 4)
        tile = synthetic.field_numpy((3, 4, y1 - y0, x1 - x0), global_hw=(8, 12),
                                     origin=(y0, x0), **kw)
        assert np.array_equal(tile, full[:, :, y0:y1, x0:x1])
    later = synthetic.field_numpy((1, 4, 8, 12), t0=2, **kw)
    assert np.array_equal(later[0], full[2])


def test_tile_bounds_cover_the_grid():
    for world in (1, 2, 4, 8):
        seen = np.zeros((1080, 1440), dtype=int)
        for r in range(world):
            y0, y1, x0, x1 = synthetic.tile_bounds(1080, 1440, r, world)
            seen[y0:y1, x0:x1] += 1
        assert (seen == 1).all()


# ---- no CPU fallback ---------------------------------------------------------------------------
def test_product_fails_loudly_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(m.MomlevelHipError):
        m.steric(dset)
    with pytest.raises(m.MomlevelHipError):
        m.eos.wright.density(18.0, 35.0, 2.0e5)
    with pytest.raises(m.MomlevelHipError):
        m.derived.calc_volo(dset["volcello"].isel(time=0))
    # the tiled (multi-GPU) front end resolves its imports and fails just as loudly
    from momlevel_amd import parallel

    for call in (lambda: parallel.steric(dset, domain="global"),
                 lambda: parallel.steric_variants(dset, domain="global", heat_content=True),
                 lambda: parallel.setup_reference_state(dset)):
        with pytest.raises(m.MomlevelHipError):
            call()


def test_product_refuses_to_bind_the_host_build(tmp_path):
    """MOMLEVEL_AMD_LIB must not be a way to run the product on the CPU
    restatement -- oracle/libmomlevel_host.so exports every symbol of the ABI.  Refused by location
    (under oracle/) and, for a copy placed elsewhere, by mlx_build_kind()."""
    import os
    import shutil
    import subprocess
    import sys

    from oracle import host_abi

    host = host_abi.build()
    copy = tmp_path / "libmomlevel_hip.so"  # renamed and moved: only the build kind gives it away
    shutil.copy(host, copy)
    root = os.path.dirname(os.path.dirname(os.path.abspath(m.__file__)))
    code = ("from momlevel_amd import _lib\n"
            "try:\n    _lib.load()\nexcept _lib.MomlevelHipError as e:\n    print('REFUSED', e)\n"
            "else:\n    print('BOUND')\n")
    for path, why in ((host, "under oracle/"), (str(copy), "not the HIP build")):
        out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                             env=dict(os.environ, MOMLEVEL_AMD_LIB=path))
        assert out.stdout.startswith("REFUSED"), out.stdout + out.stderr
        assert why in out.stdout, out.stdout


def test_product_never_imports_the_oracle():
    import os
    import re

    root = os.path.dirname(os.path.abspath(m.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_lazy_arrays_stay_lazy_in_the_labelled_layer():
    """dask-like inputs (VERDICT r1 missing #4): relabelling, transposing and picking the
    reference slab never read the whole 4-D field; only slicing does"""
    from lazy_array import CountingLazy
    from momlevel_amd.labeled import DataArray, is_lazy

    a = np.arange(6 * 3 * 4 * 5, dtype=np.float32).reshape(6, 3, 4, 5)
    lazy = CountingLazy(a)
    assert is_lazy(lazy) and not is_lazy(a) and not is_lazy([1.0, 2.0])
    da = DataArray(lazy, ("time", "z_l", "yh", "xh"))
    assert da.is_lazy and da.shape == a.shape and da.dtype == np.float32 and lazy.reads == []
    assert da.transpose("time", "z_l", "yh", "xh").data is lazy
    moved = da.transpose("time", "yh", "z_l", "xh")
    assert moved.is_lazy and moved.shape == (6, 4, 3, 5) and lazy.reads == []
    assert np.array_equal(moved.data[2:4], a.transpose(0, 2, 1, 3)[2:4])
    assert lazy.largest_read == 2 * 3 * 4 * 5 * 4
    slab = da.isel({"time": 0}).squeeze().reset_coords(drop=True)
    assert not slab.is_lazy and np.array_equal(slab.values, a[0])
    assert lazy.largest_read == 2 * 3 * 4 * 5 * 4  # still: nothing larger than the 2-step chunk
    assert np.array_equal(da.values, a)  # an explicit .values reads everything -- by request


def test_reported_time_chunks_match_the_kernel_constants():
    """bench.py reports K1's time steps per block from core.K1_TCHUNK: keep it equal to the
    constants the library is compiled with"""
    import os
    import re

    from momlevel_amd import core

    root = os.path.dirname(os.path.abspath(m.__file__))
    text = open(os.path.join(root, "csrc", "momlevel_hip.hip")).read()
    assert int(re.search(r"constexpr int kTChunk = (\d+);", text).group(1)) == core.K1_TCHUNK["steric"]
    assert int(re.search(r"constexpr int kTChunkHeld = (\d+);", text).group(1)) == core.K1_TCHUNK["held"]


def test_leading_one_pressure_is_not_time_dependent():
    """a (1,nz,1,1) pressure is a z profile (core._pressure squeezes it); slicing it per
    time chunk would hand chunk 2 an empty operand"""
    from momlevel_amd import engine

    assert not engine.time_dependent(np.zeros((1, 5, 1, 1)))
    assert engine.time_dependent(np.zeros((3, 5, 1, 1)))
    assert not engine.time_dependent(np.zeros((5, 4, 3)))
    p = np.arange(5.0).reshape(1, 5, 1, 1)
    assert engine.pressure_chunk(p, 2, 4, None) is p


def test_operand_kinds_follow_numpys_promotion():
    """eos/_dispatch._kind: python scalars are weak, float32 arrays float32, and integer / boolean
    arrays float64 -- numpy promotes ``int_array * python_float`` to float64, so a field of integers
    behaves exactly like its float64 conversion (checked here against numpy itself on the oracle's
    op-for-op Wright expression); float16 arrays, whose part numpy would evaluate in float16, are
    refused"""
    import pytest

    from momlevel_amd.eos import _dispatch
    from oracle import momlevel_numpy as o

    S = np.array([35.0, 34.5], np.float32)
    assert _dispatch._kind(3.5) == "weak" and _dispatch._kind(S) == "f32"
    assert _dispatch._kind(np.float32(1.0)) == "f32" and _dispatch._kind(np.float64(1.0)) == "f64"
    for dt in (np.bool_, np.int16, np.uint8, np.int32, np.int64):
        T = np.array([10, 1]).astype(dt)
        assert (T * 3.5).dtype == np.float64  # the premise
        assert _dispatch._kind(T) == "f64"
        got, want = o.wright_density(T, S, 2.0e7), o.wright_density(T.astype(np.float64), S, 2.0e7)
        assert got.dtype == want.dtype == np.float64 and np.array_equal(got, want)
    with pytest.raises(TypeError, match="float16"):
        _dispatch._kind(np.array([10.0], np.float16))


def test_deferred_slices_read_only_when_converted():
    """hostio.leading_slice: a numpy array is sliced as a view at once; anything that is READ by
    slicing (dask / netCDF4 / h5py-like) is wrapped and read where np.asarray is called on it --
    in the pipelined host paths that is the upload worker's thread"""
    from lazy_array import CountingLazy
    from momlevel_amd import hostio

    a = np.arange(24.0).reshape(4, 3, 2)
    v = hostio.leading_slice(a, 1, 3)
    assert isinstance(v, np.ndarray) and np.shares_memory(v, a)
    lazy = CountingLazy(a.astype(np.float32))
    d = hostio.leading_slice(lazy, 1, 3)
    assert not isinstance(d, np.ndarray) and d.shape == (2, 3, 2) and d.dtype == np.float32
    assert lazy.reads == []  # nothing read yet
    got = np.ascontiguousarray(d)
    assert lazy.reads == [2 * 3 * 2 * 4] and np.array_equal(got, a[1:3].astype(np.float32))
    assert np.asarray(d, dtype=np.float64).dtype == np.float64


def test_masked_arrays_mean_nan_wherever_an_array_enters():
    """The reference only ever sees xarray objects (steric.py:84-96); ``xr.DataArray(masked)`` --
    and ``open_mfdataset`` decoding ``_FillValue`` (examples/example.ipynb cell 4) -- hold NaN where
    the numpy masked array a netCDF4 read returns is masked.  ``np.asarray`` would drop the mask and
    keep the 1e20 fill values (VERDICT r4 missing #2): every entry of host data goes through
    labeled.as_plain instead."""
    from lazy_array import FILL, MaskedLazy, as_masked
    from momlevel_amd import engine, hostio
    from momlevel_amd.labeled import DataArray, LazyTranspose, as_plain

    m0 = np.ma.masked_array([[1.0, FILL, 3.0]], mask=[[0, 1, 0]])
    da = DataArray(m0)
    assert type(da.data) is np.ndarray and da.dtype == np.float64
    assert np.array_equal(da.values, [[1.0, np.nan, 3.0]], equal_nan=True)
    assert m0.data[0, 1] == FILL and m0.mask[0, 1]  # the caller's array is not written
    # dtypes as xarray's as_compatible_data / dtypes.maybe_promote: floats keep theirs, integers of
    # up to 16 bits -> float32, wider -> float64; nothing masked -> untouched, no copy
    f32 = as_plain(np.ma.masked_array(np.array([1, 2], np.float32), mask=[1, 0]))
    assert f32.dtype == np.float32 and np.isnan(f32[0]) and f32[1] == 2
    i16 = as_plain(np.ma.masked_array(np.array([1, 2], np.int16), mask=[0, 1]))
    assert i16.dtype == np.float32 and i16[0] == 1 and np.isnan(i16[1])
    i32 = as_plain(np.ma.masked_array(np.array([1, 2], np.int32), mask=[0, 1]))
    assert i32.dtype == np.float64 and np.isnan(i32[1])
    base = np.array([1, 2], np.int32)
    kept = as_plain(np.ma.masked_array(base))  # nomask
    assert kept.dtype == np.int32 and type(kept) is np.ndarray and np.shares_memory(kept, base)
    kept = as_plain(np.ma.masked_array(base, mask=[0, 0]))
    assert kept.dtype == np.int32 and np.shares_memory(kept, base)
    plain = np.arange(3.0)
    assert as_plain(plain) is plain
    with pytest.raises(TypeError, match="masked"):
        as_plain(np.ma.masked_array(np.array([True, False]), mask=[0, 1]))
    scalar_mask = np.ma.masked_array(np.array([1.0, 2.0]), mask=True)
    assert np.isnan(as_plain(scalar_mask)).all()

    # a lazy source whose slices are masked arrays (netCDF4.Variable): every read path
    a = np.arange(4 * 3 * 2 * 5, dtype=np.float32).reshape(4, 3, 2, 5)
    a[:, 1, 0, 2] = np.nan
    a[2, :, 1, :] = np.nan
    assert as_masked(a).data[2, 0, 1, 0] == np.float32(FILL)
    lazy = MaskedLazy(a)
    lda = DataArray(lazy, ("time", "z_l", "yh", "xh"))
    assert lda.is_lazy
    assert np.array_equal(lda.values, a, equal_nan=True)
    slab = lda.isel({"time": 2}).squeeze()
    assert type(slab.data) is np.ndarray and np.array_equal(slab.values, a[2], equal_nan=True)
    moved = lda.transpose("time", "yh", "xh", "z_l")
    assert isinstance(moved.data, LazyTranspose)
    assert np.array_equal(moved.data[1:3], a.transpose(0, 2, 3, 1)[1:3], equal_nan=True)
    assert np.array_equal(np.asarray(moved.data), a.transpose(0, 2, 3, 1), equal_nan=True)
    d = hostio.leading_slice(lazy, 1, 3)
    assert np.array_equal(np.ascontiguousarray(d), a[1:3], equal_nan=True)
    # ... and an in-memory masked array handed to the chunked host paths is filled slice by slice
    d = hostio.leading_slice(as_masked(a), 1, 3)
    assert not isinstance(d, np.ndarray)
    assert np.array_equal(np.asarray(d), a[1:3], equal_nan=True)
    t = engine._host_tensor(as_masked(a)[1:3], np.float64)
    assert t.dtype.is_floating_point and np.array_equal(t.numpy(), a[1:3].astype(np.float64), equal_nan=True)
    assert np.array_equal(hostio.to_host(as_masked(a)), a, equal_nan=True)


def test_a_dask_like_array_with_masked_chunks_is_computed_before_it_is_filled():
    """dask's own ``__array__`` computes and then np.asarray's the result -- dropping the masks of
    masked chunks; as_plain computes first"""
    from lazy_array import FILL
    from momlevel_amd.labeled import as_plain

    class DaskLike:
        shape, dtype, ndim = (3,), np.dtype(np.float64), 1

        def compute(self):
            return np.ma.masked_array([1.0, FILL, 3.0], mask=[0, 1, 0])

        def __array__(self, dtype=None, copy=None):
            return np.asarray(self.compute())

        def __getitem__(self, key):
            return self

    assert np.array_equal(as_plain(DaskLike()), [1.0, np.nan, 3.0], equal_nan=True)


def test_k2_block_mapping_covers_every_tile_and_time_block_once():
    """k_steric_local's 1-D grid (csrc/momlevel_hip.hip, round 5): block id b -> xcd = b % 8,
    q = b // 8, time block = q % ntb, tile = (q // ntb) * 8 + xcd, launched with
    ceil(tiles / 8) * 8 * ntb blocks; blocks whose tile is past the plane exit.  Restated here and
    checked for what the kernel relies on: every (tile, time block) exactly once for any tile count
    (also not a multiple of 8), and a tile's ntb time blocks on ONE XCD as consecutive workgroups of
    it (ids b, b+8, ...) -- the property that turns their re-reads of rho0m / the held slab into L2
    hits.  The decode in the kernel is these three lines; the GPU suite's ragged shapes run it."""
    import os
    import re

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                            "momlevel_amd", "csrc", "momlevel_hip.hip")).read()
    # the restatement below is the kernel's: keep the two in step
    assert re.search(r"const int64_t q = blockIdx\.x >> 3;\s*tb = \(int\)\(q % ntb_major\);\s*"
                     r"tile = \(q / ntb_major\) \* 8 \+ \(blockIdx\.x & 7\);", src)
    assert "const int64_t blocks = ceil_div(gx, 8) * 8 * ntb;" in src
    # a HIP grid holds fewer than 2^32 threads, so the 1-D form is taken only up to
    # 2^24 - 1 blocks of 256; beyond it the (tiles, time blocks) grid -- gx <= 2^31 - 1 on grid.x,
    # ntb <= 65535 on grid.y -- launches as before
    assert "constexpr int64_t kK2MaxBlocks1D = ((int64_t)1 << 24) - 1;" in src
    assert "ntb > 1 && blocks <= kK2MaxBlocks1D" in src
    assert re.search(r"constexpr int kBlock = 256;", src)
    for gx, ntb, one_d in ((3038, 15, True), (2097144, 8, True), (2097151, 8, False),
                           (1 << 21, 9, False), (8, 2097151, True), (9, 2097151, False)):
        blocks = -(-gx // 8) * 8 * ntb
        assert (blocks <= (1 << 24) - 1) == one_d and (not one_d or blocks * 256 < 1 << 32)
    for tiles in (1, 7, 8, 9, 64, 3038):
        for ntb in (2, 3, 10, 15):
            blocks = -(-tiles // 8) * 8 * ntb
            seen = {}
            for b in range(blocks):
                q = b >> 3
                tb, tile = q % ntb, (q // ntb) * 8 + (b & 7)
                if tile >= tiles:
                    continue
                assert (tile, tb) not in seen
                seen[(tile, tb)] = b
            assert len(seen) == tiles * ntb
            for tile in range(tiles):
                ids = [seen[(tile, tb)] for tb in range(ntb)]
                assert {i % 8 for i in ids} == {tile % 8}           # one XCD
                assert [i // 8 for i in ids] == list(range(ids[0] // 8, ids[0] // 8 + ntb))  # adjacent there


def test_as_plain_property_random_masked_arrays():
    """labeled.as_plain against its definition -- xarray's as_compatible_data for masked arrays:
    where(~mask, data, NaN) after dtypes.maybe_promote -- on random shapes, dtypes, masks (incl.
    nomask, scalar masks, non-contiguous views); the input is never written."""
    hypothesis = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    from hypothesis.extra import numpy as hnp

    from momlevel_amd.labeled import as_plain

    dtypes = st.sampled_from([np.float64, np.float32, np.int8, np.int16, np.uint16, np.int32, np.int64])

    @settings(max_examples=120, deadline=None)
    @given(st.data())
    def check(data):
        dt = np.dtype(data.draw(dtypes))
        shape = data.draw(hnp.array_shapes(min_dims=0, max_dims=4, max_side=5))
        elems = (st.floats(-1e6, 1e6, width=32) if dt.kind == "f" else st.integers(-100, 100)
                 if dt.kind == "i" else st.integers(0, 200))
        arr = data.draw(hnp.arrays(dt, shape, elements=elems))
        kind = data.draw(st.sampled_from(["array", "nomask", "all", "none"]))
        if kind == "array":
            mask = data.draw(hnp.arrays(np.bool_, shape))
        else:
            mask = {"nomask": np.ma.nomask, "all": True, "none": False}[kind]
        m = np.ma.masked_array(arr.copy(), mask=mask)
        if m.ndim >= 2 and data.draw(st.booleans()):
            m = m.T  # a non-contiguous view
        before = (np.ma.getdata(m).copy(), np.ma.getmaskarray(m).copy())
        got = as_plain(m)
        full = np.ma.getmaskarray(m)
        assert type(got) is np.ndarray and got.shape == m.shape
        if not full.any():
            assert got.dtype == dt and np.array_equal(got, np.ma.getdata(m))
        else:
            want_dt = dt if dt.kind == "f" else (np.float32 if dt.itemsize <= 2 else np.float64)
            assert got.dtype == want_dt
            assert np.array_equal(np.isnan(got), full | (np.isnan(before[0]) if dt.kind == "f" else False))
            assert np.array_equal(got[~full], before[0][~full].astype(want_dt))
        assert np.array_equal(np.ma.getdata(m), before[0], equal_nan=True)
        assert np.array_equal(np.ma.getmaskarray(m), before[1])

    check()


def test_big_endian_data_are_float32_and_float64_like_any_other(tmp_path):
    """NetCDF-3 stores big-endian; scipy.io.netcdf_file hands out ">f4" / ">f8" (masked) arrays.
    numpy -- and the reference, which sees such data through xarray in native order -- treats them as
    float32 / float64; a dtype test written ``str(dt) == "float32"`` or ``dt in (np.float32, ...)``
    does not (``np.dtype(">f4") == np.float32`` is False) and would send float32 fields down the
    float64 path.  Every entry of host data returns native byte order; every dtype query goes through
    labeled.dtype_name."""
    from scipy.io import netcdf_file

    from lazy_array import NetCDFVar, write_netcdf3
    from momlevel_amd import engine
    from momlevel_amd.eos import _dispatch
    from momlevel_amd.labeled import as_plain, dtype_name

    be = np.arange(6, dtype=">f4").reshape(2, 3)
    assert dtype_name(be.dtype) == "float32" and dtype_name(np.dtype(">f8")) == "float64"
    got = as_plain(be)
    assert got.dtype == np.float32 and got.dtype.isnative and np.array_equal(got, be)
    m = as_plain(np.ma.masked_array(be, mask=[[0, 1, 0], [0, 0, 0]]))
    assert m.dtype == np.float32 and m.dtype.isnative and np.isnan(m[0, 1]) and m[1, 2] == 5
    assert DataArray(be).dtype == np.float32
    assert _dispatch._kind(be) == "f32" and _dispatch._kind(be.astype(">f8")) == "f64"
    assert str(engine._stream_dtype(be)) == "torch.float32" and engine.sum_dtype(be) is np.float32
    t = engine._host_tensor(be)
    assert str(t.dtype) == "torch.float32" and np.array_equal(t.numpy(), be)
    # ... and from a real file
    d = generate_test_data()
    d32 = d.copy()
    for k in ("thetao", "so"):
        a = d[k].values.astype(np.float32)
        a[:, 0, 1, :] = np.nan
        d32[k] = DataArray(a, d[k].dims)
    path = str(tmp_path / "t.nc")
    write_netcdf3(path, d32)
    f = netcdf_file(path, "r", mmap=False, maskandscale=True)
    try:
        # a scipy variable handed over as it is (it has no dtype attribute): wrapped, stays lazy
        raw = DataArray(f.variables["thetao"], tuple(f.variables["thetao"].dimensions))
        assert raw.is_lazy and raw.dtype == np.float32 and raw.shape == d32["thetao"].shape
        assert np.array_equal(raw.values, d32["thetao"].values, equal_nan=True)
        assert np.array_equal(raw.isel({"time": 1}).values, d32["thetao"].values[1], equal_nan=True)
        del raw
        v = NetCDFVar(f.variables["thetao"])
        assert str(v.dtype) == ">f4" and isinstance(v[0:1], np.ma.MaskedArray)
        da = DataArray(v, tuple(f.variables["thetao"].dimensions))
        assert da.is_lazy and da.dtype == np.float32
        vals = da.values
        assert vals.dtype == np.float32 and vals.dtype.isnative
        assert np.array_equal(vals, d32["thetao"].values, equal_nan=True)
        slab = da.isel({"time": 2}).squeeze().values
        assert np.array_equal(slab, d32["thetao"].values[2], equal_nan=True)
    finally:
        f.close()


def test_field_dtypes_that_numpy_would_compute_in_are_refused_not_upcast():
    from momlevel_amd import engine
    from momlevel_amd.labeled import check_field_dtype

    assert check_field_dtype(np.dtype(">f4")) == "float32" and check_field_dtype(np.int16) == "int16"
    for bad in (np.float16, np.longdouble, np.complex128):
        with pytest.raises(TypeError, match="not supported"):
            check_field_dtype(np.dtype(bad))
        with pytest.raises(TypeError):
            engine._stream_dtype(np.zeros(3, dtype=bad))


def test_a_big_masked_array_is_held_as_data_and_mask_until_it_is_uploaded():
    """``DataArray(nc.variables["thetao"][:])`` -- a 13 GB in-memory masked array
    in the reference's recorded call -- used to be NaN-filled into a copy on construction (0.5-0.7 s
    before the call).  Now a large floating masked array is kept as it is (labeled.MaskedSource) and
    read like a netCDF4 variable: every time chunk a masked VIEW, which hostio.split_masked hands
    the staging copy as data + mask; ``.values`` / ``np.asarray`` / slabs fill on demand."""
    import torch
    from lazy_array import FILL, as_masked
    from momlevel_amd import engine, hostio
    from momlevel_amd.labeled import DataArray, Dataset, LazyTranspose, MaskedSource

    rng = np.random.default_rng(5)
    a = rng.normal(10.0, 3.0, (6, 5, 40, 900)).astype(np.float32)  # 4.3 MB: above the threshold
    a[:, :, 3:9, 100:300] = np.nan
    a[2, 1] = np.nan
    m = as_masked(a)
    assert MaskedSource.wanted(m) and not MaskedSource.wanted(m[:1, :1, :4])  # small: filled at once
    assert not MaskedSource.wanted(np.ma.masked_array(a))  # nomask: plain data
    da = DataArray(m, ("time", "z_l", "yh", "xh"))
    src = da.data
    assert isinstance(src, MaskedSource) and da.is_lazy and src.array is m  # nothing copied
    assert da.dtype == np.float32 and da.shape == a.shape
    # a time chunk is a VIEW of the caller's data and mask ...
    chunk = src[2:5]
    assert isinstance(chunk, np.ma.MaskedArray) and np.shares_memory(np.ma.getdata(chunk), m.data)
    # ... which travels as (data, mask): no NaN-filled intermediate
    data, mask = hostio.split_masked(chunk, np.dtype(np.float32))
    assert np.shares_memory(data, m.data) and mask.dtype == np.bool_ and mask.shape == data.shape
    assert np.array_equal(mask, np.isnan(a[2:5])) and data[0, 1, 0, 0] == np.float32(FILL)
    # the engine sees a streamable float32 field and cuts it like any lazy one
    assert engine._stream_dtype(src) == torch.float32
    d = hostio.leading_slice(src, 1, 3)
    assert isinstance(d, hostio._DeferredSlice) and isinstance(d.read(), np.ma.MaskedArray)
    # fill on demand: .values, np.asarray, slabs, transposes, Dataset round trips
    assert np.array_equal(da.values, a, equal_nan=True) and type(da.values) is np.ndarray
    assert np.array_equal(np.asarray(src), a, equal_nan=True)
    assert np.asarray(src, dtype=np.float64).dtype == np.float64
    slab = da.isel({"time": 2})
    assert np.array_equal(slab.values, a[2], equal_nan=True)
    moved = da.transpose("time", "yh", "xh", "z_l")
    assert isinstance(moved.data, LazyTranspose)
    assert np.array_equal(moved.data[1:3], a.transpose(0, 2, 3, 1)[1:3], equal_nan=True)
    ds = Dataset()
    ds["thetao"] = da
    assert isinstance(ds["thetao"].data, MaskedSource) and ds["thetao"].data.array is m
    assert isinstance(ds.rename({"thetao": "temp"})["temp"].data, MaskedSource)
    assert np.array_equal(hostio.to_host(src), a, equal_nan=True)
    # the caller's array is never written
    assert m.data[2, 1, 0, 0] == np.float32(FILL) and m.mask[2, 1, 0, 0]
    # big-endian data (a NetCDF-3 read): native dtype outside, filled natively on the way
    be = np.ma.masked_array(np.where(np.isnan(a), FILL, a).astype(">f4"), mask=np.isnan(a))
    dbe = DataArray(be, ("time", "z_l", "yh", "xh"))
    assert isinstance(dbe.data, MaskedSource) and dbe.dtype == np.float32 and dbe.dtype.isnative
    assert np.array_equal(dbe.values, a, equal_nan=True)


def test_string_and_bytes_variables_keep_their_dtype_and_are_skipped_by_annual_average():
    """``np.dtype(dt).name`` does not round-trip for str / bytes / void dtypes
    ('<U3' -> 'str96'); DataArray.dtype and util.annual_average (which skips non-numeric variables,
    util.py:79-84) raised TypeError on them."""
    import torch
    from momlevel_amd import util
    from momlevel_amd.labeled import DataArray, np_dtype
    from momlevel_amd.test_data import generate_test_data_time

    s = DataArray(np.array(["abc", "de"]), ("n",))
    assert s.dtype == np.dtype("<U3") and s.dtype.kind == "U"
    assert DataArray(np.array([b"ab", b"c"]), ("n",)).dtype == np.dtype("S2")
    rec = np.zeros(2, dtype=[("a", "<f4"), ("b", "<i2")])
    assert DataArray(rec, ("n",)).dtype == rec.dtype
    assert np_dtype(np.dtype(">f4")) == np.float32 and np_dtype(np.dtype(">f4")).isnative
    assert np_dtype(np.dtype(">i2")) == np.int16 and np_dtype(torch.float32) == np.float32
    assert np_dtype(np.dtype("O")) == np.dtype("O")
    dset = generate_test_data_time()
    nt = dset["time"].shape[0]
    dset["label"] = DataArray(np.array(["m%02d" % (i % 12) for i in range(nt)]), ("time",))
    dset["tag"] = DataArray(np.array([b"x"] * nt), ("time",))
    numeric = [k for k, v in dset.data_vars.items() if v.dtype.kind in "fiu"]
    out = util.annual_average(dset)
    assert "label" not in out and "tag" not in out
    assert sorted(out.data_vars) == sorted(numeric)


def test_element_misaligned_masked_data_fall_back_to_numpy():
    """a float64 view of a buffer at a 4-byte offset is C-contiguous and native but
    not ALIGNED; mlx_host_copy_masked answers MLX_E_ALIGN.  as_plain fills it with numpy instead
    (as its docstring promises) and split_masked does not hand it to the staging copy."""
    from momlevel_amd import hostio
    from momlevel_amd.labeled import _native_fill, as_plain

    n = 1 << 20  # 8 MiB of float64: above the native-fill threshold
    raw = np.zeros(8 * n + 8, dtype=np.uint8)
    off = 4 if raw.ctypes.data % 8 == 0 else (12 - raw.ctypes.data % 8) % 8 or 4
    a = np.frombuffer(raw, dtype=np.float64, count=n, offset=off)
    assert a.flags["C_CONTIGUOUS"] and not a.flags["ALIGNED"]
    mask = np.zeros(n, dtype=bool)
    mask[::3] = True
    m = np.ma.masked_array(a, mask=mask)
    assert _native_fill(a, mask) is None
    got = as_plain(m)
    assert np.array_equal(np.isnan(got), mask) and not got[~mask].any()
    plain, mk = hostio.split_masked(m)
    assert mk is None and np.array_equal(np.isnan(plain), mask)


def test_small_integer_masked_sources_in_memory_follow_xarray_lazy_ones_are_documented():
    """xarray promotes a masked int16 variable to float32 (dtypes.maybe_promote), so
    the reference computes such a field in float32.  In memory that is what happens here too -- the
    DataArray holds float32 from the moment the masked array is wrapped.  A LAZY int16 source only
    declares "int16": it streams as float64 (its masked slices, float32 after as_plain, widen
    exactly) -- the deviation documented at labeled.check_field_dtype, parity unpinned."""
    import torch
    from lazy_array import MaskedLazy
    from momlevel_amd import engine, hostio
    from momlevel_amd.labeled import DataArray, check_field_dtype

    a = np.arange(24, dtype=np.int16).reshape(2, 3, 4)
    m = np.ma.masked_array(a, mask=(a % 5 == 0))
    da = DataArray(m, ("t", "y", "x"))
    assert da.dtype == np.float32 and np.isnan(da.values[0, 0, 0]) and da.values[0, 0, 1] == 1.0
    assert engine._stream_dtype(da.data) == torch.float32
    assert DataArray(a, ("t", "y", "x")).dtype == np.int16  # unmasked: numpy's int * float is float64
    assert engine._stream_dtype(a) == torch.float64
    lazy = MaskedLazy(a.astype(np.float64))  # (the stand-in masks NaNs: build the int16 case by hand)

    class Int16Lazy:
        shape, dtype, ndim = a.shape, a.dtype, a.ndim

        def __getitem__(self, key):
            return m[key]

    src = Int16Lazy()
    assert engine._stream_dtype(src) == torch.float64  # the documented deviation
    plain, mask = hostio.split_masked(src[0:1], np.dtype(np.float64))
    assert mask is None and plain.dtype == np.float64 and np.isnan(plain[0, 0, 0]) and plain[0, 0, 1] == 1.0
    assert "int16" in check_field_dtype.__doc__ and "parity unpinned" in check_field_dtype.__doc__
    del lazy
