"""GPU: libmomlevel_hip.so against the HOST build of the same ABI (oracle/libmomlevel_host.so) on
identical inputs -- two builder-authored implementations behind one header (separately written,
except mlx_eos_map_promote: its host build compiles the product's own eos_promote.hpp): pointwise
outputs (rho, delta_rho, eta, dz) bit for bit, sums within 1e-12.  A consistency check between the
two builds; what pins either of them to the REFERENCE are the golden vectors (test_host_abi.py,
test_gpu_wright.py, test_gpu_promote.py)."""

import numpy as np
import pytest
import torch

from momlevel_amd import core, synthetic
from oracle import host_abi as h
from oracle import momlevel_numpy as o
from conftest import assert_bit_equal, assert_rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,f32_mode", [(np.float64, "faithful"), (np.float32, "faithful"),
                                            (np.float32, "upcast"),
                                            ((np.float32, np.float64), "faithful"),
                                            ((np.float64, np.float32), "faithful")])
@pytest.mark.parametrize("shape", [(9, 6, 12, 40), (4, 3, 7, 9)])
def test_hip_library_matches_the_host_build(shape, dtype, f32_mode):
    nt, nz, ny, nx = shape
    g = synthetic.make_grid(ny, nx, nz)
    r = np.random.default_rng(17)
    mask = np.isnan(g["volcello"])
    mixed = isinstance(dtype, tuple)  # theta and salinity of different dtypes: K1 / K2 only
    T = np.where(mask[None], np.nan, r.uniform(-2, 32, shape)).astype(dtype[0] if mixed else dtype)
    S = np.where(mask[None], np.nan, r.uniform(30, 40, shape)).astype(dtype[1] if mixed else dtype)
    pres = o.pressure_from_depth(g["z_l"])
    vol = g["volcello"]
    dT, dS, dvol = torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda(), torch.from_numpy(vol).cuda()
    kw = dict(f32_mode=f32_mode)
    for func in () if mixed else ("density", "drho_dtemp", "drho_dsal", "alpha", "beta"):
        assert_bit_equal(core.eos_map(dT, dS, pres, func=func, **kw).cpu().numpy(),
                         h.eos_map(T, S, pres, func=func, **kw), func)
    rows = core.steric_global_decomp(dT, dS, dT[0], dS[0], dvol, pres, **kw).cpu().numpy()
    assert_rel(rows, h.steric_global_decomp(T, S, T[0], S[0], vol, pres, **kw), 1e-12, "rows")
    rho0 = core.eos_map(dT[0], dS[0], pres, **kw)
    rho0m = core.fold_mask(rho0, dvol)
    assert_bit_equal(rho0m.cpu().numpy(), h.fold_mask(rho0.cpu().numpy(), vol), "rho0m")
    d3, e3 = core.steric_local_decomp(dT, dS, dT[0], dS[0], rho0m, dvol[0], pres, -1.0 / 1035.0,
                                      z_i=g["z_i"], deptho=g["deptho"], **kw)
    hd3, he3 = h.steric_local_decomp(T, S, T[0], S[0], rho0m.cpu().numpy(), vol[0], pres,
                                     -1.0 / 1035.0, z_i=g["z_i"], deptho=g["deptho"], **kw)
    assert_bit_equal(d3.cpu().numpy(), hd3, "delta_rho x3")
    assert_bit_equal(e3.cpu().numpy(), he3, "eta x3")
    assert_bit_equal(core.calc_dz(g["z_i"], torch.from_numpy(g["deptho"]).cuda()).cpu().numpy(),
                     h.calc_dz(g["z_i"], g["deptho"]), "calc_dz")
    assert core.nansum(dvol).item() == pytest.approx(h.nansum(vol), rel=1e-13)


@pytest.mark.parametrize("dtype,f32_mode", [(np.float64, "faithful"), (np.float32, "faithful"),
                                            (np.float32, "upcast")])
@pytest.mark.parametrize("levels", ["uneven", "even"])
def test_stratification_matches_the_host_build(dtype, f32_mode, levels):
    """mlx_stratification / mlx_adjust_negative_n2 / mlx_wave_speed_where_time0: the HIP kernels
    against the host build's plain C loops (separately written: oracle/host_abi.c) on the same
    inputs -- N^2, the adjustment and the wave speed bit for bit, the stability angle to the last
    bits of the two arctans.  (What pins them to the REFERENCE: tests/test_gpu_stratification.py.)"""
    nt, nz, ny, nx = 3, 9, 6, 20
    plane = ny * nx
    r = np.random.default_rng(23)
    T = r.uniform(-2, 32, (nt, nz, plane)).astype(dtype)
    S = r.uniform(30, 40, (nt, nz, plane)).astype(dtype)
    land = r.random(plane) < 0.2
    T[..., land] = np.nan
    S[..., land] = np.nan
    z = np.cumsum(2.0 * 1.3 ** np.arange(nz)) if levels == "uneven" else 4.0 * np.arange(nz) + 2.0
    p = z * 1.0e4 + 101325.0
    coef, uniform, two_dx = core.gradient_coefficients(z)
    dT, dS = torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda()
    kw = dict(f32_mode=f32_mode)
    n2 = core.stratification(dT, dS, torch.from_numpy(p).cuda(), z, **kw)
    hn2 = h.stratification(T, S, p, coef, uniform, two_dx, **kw)
    assert_bit_equal(n2.cpu().numpy(), hn2, "n2")
    tu = core.stratification(dT, dS, torch.from_numpy(p).cuda(), z, func="turner", **kw).cpu().numpy()
    htu = h.stratification(T, S, p, coef, uniform, two_dx, func="turner", **kw)
    assert np.array_equal(np.isnan(tu), np.isnan(htu))
    assert np.nanmax(np.abs(tu - htu)) <= 90.0 * 1e-12
    dz = np.abs(r.normal(10.0, 3.0, (nz, plane)))
    adj, speed = core.adjust_negative_n2(n2, 1, dz=torch.from_numpy(dz).cuda())
    hadj, hspeed = h.adjust_negative_n2(hn2, 1, dz=dz)
    assert_bit_equal(adj.cpu().numpy(), hadj, "adjusted")
    assert_bit_equal(speed.cpu().numpy(), hspeed, "speed")
    assert_bit_equal(core.wave_speed_where_time0(n2[0], speed).cpu().numpy(),
                     h.wave_speed_where_time0(hn2[0], hspeed), "speed broadcast")
    one, sp1 = core.adjust_negative_n2(n2[:1], 0, dz=torch.from_numpy(dz).cuda())
    hone, hsp1 = h.adjust_negative_n2(hn2[:1], 0, dz=dz)
    assert_bit_equal(one.cpu().numpy(), hone, "adjusted, z leading")
    assert_bit_equal(sp1.cpu().numpy(), hsp1, "speed, z leading")
    if dtype == np.float64:
        lin = core.stratification(dT, dS, None, z, eos="linear")
        assert_bit_equal(lin.cpu().numpy(), h.stratification(T, S, None, coef, uniform, two_dx, eos="linear"))
