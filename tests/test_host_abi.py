"""CPU: the HOST build of the C ABI (oracle/host_abi.c -> libmomlevel_host.so).

A second restatement of the path in plain C (independent of the device code except for
mlx_eos_map_promote, whose host build compiles the product's own eos_promote.hpp) behind the SAME header as the HIP
library.  Pinned here against (1) the vectors the reference's own eos/wright.py produced
(tests/golden/wright_vectors.npz) -- bit for bit, float64 and float32 mixed precision, all five
functions, broadcast held fields; (2) the reference's goldens through the numpy oracle's steric();
(3) the numpy oracle on random land-masked cases.  tests/test_gpu_host_abi.py then holds the HIP
library against this build on the GPU box."""

import ctypes
import os
import re

import numpy as np
import pytest

from momlevel_amd import _lib as abi
from momlevel_amd import synthetic
from oracle import host_abi as h
from oracle import momlevel_numpy as o
from conftest import MIX_FUNCS, MIX_KINDS, mixed_operands, assert_bit_equal, assert_rel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_build_exports_the_whole_abi():
    lib = h.load()
    header = open(os.path.join(ROOT, "include", "momlevel_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t)\s+(mlx_\w+)\s*\(", header, flags=re.M))
    assert declared == set(abi.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mlx_version() == abi.ABI_VERSION


@pytest.mark.parametrize("tag", ["tw", "rnd", "blk"])
@pytest.mark.parametrize("func", ["density", "drho_dtemp", "drho_dsal", "alpha", "beta"])
def test_eos_bit_identical_to_reference_vectors(wright_vectors, tag, func):
    v = wright_vectors
    T, S, p = v[f"{tag}_T"], v[f"{tag}_S"], v[f"{tag}_p"]
    shape = np.broadcast_shapes(T.shape, S.shape, p.shape)
    n = int(np.prod(shape))
    Tb, Sb, pb = (np.ascontiguousarray(np.broadcast_to(x, shape)).reshape(1, 1, n) for x in (T, S, p))
    got = h.eos_map(Tb, Sb, pb.reshape(1, 1, 1, n), func=func).reshape(shape)
    assert_bit_equal(got, v[f"{tag}_{func}"], f"host {tag}/{func}")


def test_linear_density_with_a_reference_density_through_the_host_build(wright_vectors):
    """MLX_FUNC_DENSITY_REF (ABI v8): eos.linear.density's rho_ref form -- the constant term rides in
    the p operand with its kind -- against the reference module's outputs, dtype included; refused
    for the Wright EOS (which has no such argument)."""
    v = wright_vectors
    py, py2, n64, n32 = (float(x) for x in v["linref_values"])
    for tag, (T, S) in {"tw": (v["tw_T"], v["tw_S"]), "f32": (v["f32_T"], v["f32_S"])}.items():
        for rk, rv in {"py": py, "py2": py2, "np64": np.float64(n64), "np32": np.float32(n32)}.items():
            base = 1000.0 - rv  # eos/linear.py:55, as python / numpy form it
            if isinstance(base, np.generic):
                base = np.asarray(base).reshape(1)
            got = h.eos_map_promote(T.reshape(-1), S.reshape(-1), base, eos="linear", func="density_ref")
            want = v[f"linref_{tag}_{rk}"]
            assert got.dtype == want.dtype, (tag, rk)
            assert_bit_equal(got.reshape(want.shape), want, f"host density_ref {tag}/{rk}")
    with pytest.raises(RuntimeError, match="rho_ref"):
        h.eos_map_promote(v["tw_T"].reshape(-1), v["tw_S"].reshape(-1), 1.0, eos="wright", func="density_ref")
    with pytest.raises(RuntimeError):  # the constant is required
        h.eos_map_promote(v["tw_T"].reshape(-1), v["tw_S"].reshape(-1), None, eos="linear", func="density_ref")


@pytest.mark.parametrize("func", ["density", "drho_dtemp", "drho_dsal", "alpha", "beta"])
def test_float32_mixed_precision_and_held_fields(wright_vectors, func):
    v = wright_vectors
    got = h.eos_map(v["f32_T"], v["f32_S"], v["blk_p"], func=func)
    assert_bit_equal(got, v[f"f32_{func}"], f"host f32 {func}")
    if func == "density":
        assert_bit_equal(h.eos_map(v["f32_T"], v["f32_S"][0], v["blk_p"]), v["f32_density_heldS"])
        assert_bit_equal(h.eos_map(v["f32_T"][0], v["f32_S"], v["blk_p"]), v["f32_density_heldT"])
        assert_bit_equal(h.eos_map(v["blk_T"], v["blk_S"][0], v["blk_p"]), v["f64_density_heldS"])
        up = h.eos_map(v["f32_T"], v["f32_S"], v["blk_p"], f32_mode="upcast")
        assert_bit_equal(up, o.wright_density(v["f32_T"].astype(float), v["f32_S"].astype(float),
                                              v["blk_p"]))


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_pathological_operands_through_the_host_build(wright_vectors, prec):
    """huge / tiny / infinite operands and pressures of 1e300 / 1e-300: the reference module's own
    outputs (inf, 0, NaN included), bit for bit"""
    v = wright_vectors
    got = h.eos_map(v[f"patho_{prec}_T"], v[f"patho_{prec}_S"], v["patho_p"])
    assert_bit_equal(got, v[f"patho_{prec}_density"], f"host, pathological operands, {prec}")


@pytest.mark.parametrize("func", ["density", "alpha", "beta", "drho_dtemp", "drho_dsal"])
def test_linear_eos_bit_identical_to_reference_vectors(wright_vectors, func):
    v = wright_vectors
    for tag, (T, S) in {"blk": (v["blk_T"], v["blk_S"]), "f32": (v["f32_T"], v["f32_S"])}.items():
        got = h.eos_map(T, S, 0.0, eos="linear", func=func)
        if func in ("drho_dtemp", "drho_dsal"):
            assert np.all(got == (-0.2 if func == "drho_dtemp" else 0.8))
        else:
            assert_bit_equal(got, v[f"lin_{tag}_{func}"].astype(np.float64), f"host linear {tag}/{func}")


def _case(shape, dtype, seed=3):
    nt, nz, ny, nx = shape
    g = synthetic.make_grid(ny, nx, nz)
    r = np.random.default_rng(seed)
    mask = np.isnan(g["volcello"])
    T = np.where(mask[None], np.nan, r.uniform(-2, 32, shape)).astype(dtype)
    S = np.where(mask[None], np.nan, r.uniform(30, 40, shape)).astype(dtype)
    return g, T, S


@pytest.mark.parametrize("dtype", [np.float64, np.float32, (np.float32, np.float64),
                                   (np.float64, np.float32)])
@pytest.mark.parametrize("shape", [(7, 5, 12, 20), (3, 4, 7, 9)])
def test_fused_passes_match_the_numpy_oracle(shape, dtype):
    """(a pair of dtypes: theta and salinity stored with different precisions -- numpy's promotion
    per sub-expression, MLX_DTYPE_T32_S64 / _T64_S32)"""
    if isinstance(dtype, tuple):
        g, T, S = _case(shape, np.float64)
        T, S = T.astype(dtype[0]), S.astype(dtype[1])
    else:
        g, T, S = _case(shape, dtype)
    pres = o.pressure_from_depth(g["z_l"])
    pb = pres[:, None, None]
    vol = g["volcello"]
    rows = h.steric_global_decomp(T, S, T[0], S[0], vol, pres)
    combos = [(T, S), (T, S[0]), (T[0], S)]
    for row, (a, b) in zip(rows, combos):
        ref = np.nansum(np.broadcast_to(o.wright_density(a, b, pb), T.shape) * vol, axis=(1, 2, 3))
        assert_rel(row, ref, 1e-12, "host masso")
        assert np.array_equal(row, h.steric_global(a, b, vol, pres))  # stride-0 held field
    assert_rel(rows[3], np.nansum(T.astype(np.float64) * vol, axis=(1, 2, 3)), 1e-12, "host heat")
    rho0 = o.wright_density(T[0], S[0], pb)
    rho0m = h.fold_mask(rho0, vol)
    dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"])
    assert_bit_equal(h.calc_dz(g["z_i"], g["deptho"]), dz, "host calc_dz")
    d3, e3 = h.steric_local_decomp(T, S, T[0], S[0], rho0m, vol[0], pres, -1.0 / 1035.0,
                                   z_i=g["z_i"], deptho=g["deptho"])
    for i, (a, b) in enumerate(combos):
        dref = np.where(~np.isnan(vol), np.broadcast_to(o.wright_density(a, b, pb), T.shape) - rho0, np.nan)
        eref = np.where(~np.isnan(vol[0]), (-1.0 / 1035.0) * np.nansum(dz * dref, axis=1), np.nan)
        assert_bit_equal(d3[i], dref, f"host delta_rho {i}")
        assert_bit_equal(e3[i], eref, f"host eta {i}")
        d1, e1 = h.steric_local(a, b, rho0m, vol[0], pres, -1.0 / 1035.0, dz=dz)
        assert_bit_equal(d1, dref)
        assert_bit_equal(e1, eref)
    assert h.nansum(vol) == pytest.approx(np.nansum(vol), rel=1e-13)


def test_reference_goldens_through_the_host_build(goldens):
    """momlevel's own tight goldens (tests/test_steric.py:32-77 local sums, test_derived.py calc_dz)
    reproduced by the C build, as they are by the numpy oracle"""
    d = o.generate_test_data()
    T, S, vol = d["thetao"], d["so"], d["volcello"][0]
    pres = o.pressure_from_depth(d["z_l"])
    rho0 = h.eos_map(T[0], S[0], pres)[0]
    rho0m = h.fold_mask(rho0, vol)
    d3, e3 = h.steric_local_decomp(T, S, T[0], S[0], rho0m, vol[0], pres, -1.0 / 1035.0,
                                   z_i=d["z_i"], deptho=d["deptho"])
    for i, variant in enumerate(("steric", "thermosteric", "halosteric")):
        ores, _ = o.steric(d["thetao"], d["so"], d["volcello"], d["areacello"], d["z_l"], d["z_i"],
                           d["deptho"], variant=variant)
        assert_bit_equal(d3[i], ores["delta_rho"], variant)
        assert_bit_equal(e3[i], ores[variant], variant)


def test_annual_weighted_mean_and_error_codes():
    r = np.random.default_rng(1)
    x = r.normal(size=(24, 6, 5))
    x[r.uniform(size=x.shape) < 0.2] = np.nan
    x[:, 0, 0] = np.nan
    w = np.tile([31.0, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31], 2)
    got = h.group_weighted_mean(x, w, 12)
    for gi in range(2):
        xs, ws = x[gi * 12:(gi + 1) * 12], w[gi * 12:(gi + 1) * 12].reshape(12, 1, 1)
        num = np.sum(np.where(np.isnan(xs), 0.0, xs) * ws, axis=0)
        den = np.sum(np.where(np.isnan(xs), 0.0, 1.0) * ws, axis=0)
        with np.errstate(invalid="ignore"):
            ref = num / np.where(den != 0, den, np.nan)
        assert np.allclose(got[gi], ref, rtol=1e-14, atol=0, equal_nan=True)
    lib = h.load()
    assert lib.mlx_steric_global(None, None, 0, None, None, 1, 0, 1, 1, 1, 0, 0, 0, None, None, 0,
                                 None) == -1  # MLX_E_NULL
    one = np.ones(1)
    rc = lib.mlx_steric_global(one.ctypes.data, one.ctypes.data, 9, one.ctypes.data, one.ctypes.data,
                               1, 0, 1, 1, 1, 1, 1, 0, one.ctypes.data, None, 0, None)
    assert rc == -3 and "dtype" in h.last_error()
    rc = lib.mlx_steric_global(one.ctypes.data, one.ctypes.data, 0, one.ctypes.data, one.ctypes.data,
                               1, 0, 1, 1, 1, 1, 1, abi.FLAG_FMA, one.ctypes.data, None, 0, None)
    assert rc == -3 and "exact arithmetic" in h.last_error()
    buf = ctypes.create_string_buffer(8)
    assert lib.mlx_last_error(buf, 8) > 8 and len(buf.value) == 7


def test_synthetic_generator_matches_the_numpy_replay():
    lib = h.load()
    nt, nz, ny, nx = 2, 3, 4, 5
    g = synthetic.make_grid(ny, nx, nz)
    out = np.empty((nt, nz, ny, nx))
    mask = np.ascontiguousarray(g["volcello"])
    rc = lib.mlx_synth_field(out.ctypes.data, abi.DTYPE_F64, nt, nz, ny, nx, 3, ny, nx, 0, 0,
                             synthetic.SEED, 1, -2.0, 34.0, mask.ctypes.data, None)
    assert rc == 0
    ref = synthetic.field_numpy((nt, nz, ny, nx), seed=synthetic.SEED, field_id=1, lo=-2.0,
                                scale=34.0, mask3d=g["volcello"], t0=3)
    assert_bit_equal(out, ref, "host synth vs numpy replay")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("levels", ["uneven", "even"])
def test_stratification_through_the_host_build(goldens, dtype, levels):
    """mlx_stratification / mlx_adjust_negative_n2 / mlx_wave_speed_where_time0 of the host build
    (plain C loops) against the numpy oracle -- numpy.gradient + the EOS module, as the reference
    evaluates derived.calc_n2 -- bit for bit, and against the reference's own goldens."""
    from momlevel_amd import core

    r = np.random.default_rng(2)
    nz = 7
    z = np.cumsum(2.0 * 1.3 ** np.arange(nz)) if levels == "uneven" else 3.0 * np.arange(nz) + 1.0
    T = r.uniform(-2, 30, (2, nz, 4, 6)).astype(dtype)
    S = r.uniform(30, 40, (2, nz, 4, 6)).astype(dtype)
    T[:, :, 1, 2] = np.nan
    S[:, :, 1, 2] = np.nan
    coef, uniform, two_dx = core.gradient_coefficients(z)
    assert uniform == (levels == "even")
    flat = lambda a: a.reshape(a.shape[0], nz, -1)  # noqa: E731
    p = z * 1.0e4 + 101325.0
    n2 = h.stratification(flat(T), flat(S), p, coef, uniform, two_dx)
    ref = o.calc_n2(T, S, z)
    assert_bit_equal(n2.reshape(T.shape), ref, "n2")
    tu = h.stratification(flat(T), flat(S), p, coef, uniform, two_dx, func="turner").reshape(T.shape)
    tu_ref = o.calc_stability_angle(T, S, p, z)
    assert np.nanmax(np.abs(tu - tu_ref)) <= 90.0 * 1e-12
    dz = np.abs(r.normal(10.0, 3.0, (nz, 4, 6)))
    adj, speed = h.adjust_negative_n2(n2, 1, dz=dz.reshape(nz, -1))
    assert_bit_equal(adj.reshape(T.shape), o.adjust_negative_n2(ref), "adjusted")
    quirk = h.wave_speed_where_time0(n2[0], speed).reshape(nz, 4, 6, 2)
    assert_bit_equal(quirk, o.calc_wave_speed_4d_quirk(ref, dz), "wave speed, (z,y,x,time)")
    _, sp1 = h.adjust_negative_n2(n2[:1], 0, dz=dz.reshape(nz, -1))
    assert_bit_equal(sp1.reshape(4, 6), o.calc_wave_speed(ref[0], dz), "wave speed of one time level")
    if dtype == np.float64 and levels == "uneven":  # the reference's goldens (tests/test_derived.py)
        d = o.generate_test_data()
        g = goldens["stratification"]
        c5, u5, dx5 = core.gradient_coefficients(d["z_l"])
        f5 = lambda a: a.reshape(5, 5, 25)  # noqa: E731
        n5 = h.stratification(f5(d["thetao"]), f5(d["so"]), d["z_l"] * 1.0e4 + 101325.0, c5, u5, dx5)
        assert np.allclose(np.nansum(n5), g["calc_n2_sum"])
        a5, s5 = h.adjust_negative_n2(n5, 1, dz=o.calc_dz(d["z_l"], d["z_i"], d["deptho"]).reshape(5, 25))
        assert np.allclose(np.nansum(a5), g["adjust_negative_n2_sum"])
        assert np.allclose(np.nansum(h.wave_speed_where_time0(n5[0], s5)), g["calc_wave_speed_sum"])
        t5 = h.stratification(f5(d["thetao"]), f5(d["so"]), d["z_l"] * 1.0e4, c5, u5, dx5, func="turner")
        assert np.allclose(np.nansum(t5), g["calc_stability_angle_sum"])
    with pytest.raises(ValueError):  # numpy.gradient's own refusal, restated on the host
        core.gradient_coefficients(z[:2])
