"""GPU: randomised sweep of shapes / dtypes / variants / pressure layouts / NaN patterns through
the C ABI, each case against the oracle.  Deterministic seeds: a failure names its case."""

import os

import numpy as np
import pytest
import torch

from momlevel_amd import core
from oracle import momlevel_numpy as o
from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


def draw(seed):
    r = np.random.default_rng(1000 + seed)
    nt = int(r.integers(1, 41))
    nz = int(r.integers(1, 7))
    ny = int(r.integers(1, 20))
    nx = int(r.choice([1, 2, 3, 4, 6, 8, 12, 16, 31, 64, 130, 256, 515]))
    # dtype, variant, f32 mode and pressure layout are drawn INDEPENDENTLY: seeds 0..5 already
    # cover all six dtype x variant cells, 48 seeds every cell >= 8 times (round 1 tied dtype to
    # the variant and never ran a float32 thermosteric / halosteric case)
    variant = ("steric", "thermosteric", "halosteric")[seed % 3]
    dtype = np.float32 if (seed // 3) % 2 else np.float64
    f32_mode = "upcast" if (seed // 6) % 4 == 3 else "faithful"
    p3d = (seed // 12) % 4 == 1 or seed % 7 == 3
    nan_frac = float(r.choice([0.0, 0.1, 0.5, 0.95]))
    T = r.uniform(-2, 32, (nt, nz, ny, nx)).astype(dtype)
    S = r.uniform(30, 40, (nt, nz, ny, nx)).astype(dtype)
    vol = r.uniform(1e8, 1e12, (nz, ny, nx))
    land = r.uniform(size=(nz, ny, nx)) < nan_frac
    vol[land] = np.nan
    T[:, land] = np.nan
    S[:, land] = np.nan
    z_i = np.concatenate([[0.0], np.cumsum(r.uniform(1.0, 300.0, nz))])
    z_l = 0.5 * (z_i[1:] + z_i[:-1])
    deptho = r.uniform(0.0, z_i[-1] * 1.1, (ny, nx))
    deptho[r.uniform(size=(ny, nx)) < 0.1] = np.nan
    pres = o.pressure_from_depth(z_l)
    if p3d:
        pres = pres[:, None, None] + r.normal(0.0, 300.0, (1, ny, nx))
    return dict(nt=nt, nz=nz, ny=ny, nx=nx, dtype=dtype, variant=variant, p3d=p3d,
                f32_mode=f32_mode, T=T, S=S,
                vol=vol, z_i=z_i, z_l=z_l, deptho=deptho, pres=pres)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MOMLEVEL_SWEEP_SEEDS", "48"))))
def test_random_case(seed):
    c = draw(seed)
    T, S, vol, pres = c["T"], c["S"], c["vol"], c["pres"]
    mode = c["f32_mode"]
    dT32, dS32 = T, S  # what goes to the device
    if mode == "upcast" and c["dtype"] == np.float32:
        T, S = T.astype(np.float64), S.astype(np.float64)  # the oracle sees the upcast values
    pb = pres if c["p3d"] else pres[:, None, None]
    rho0 = o.wright_density(T[0], S[0], pb)
    Th, Sh = T, S
    if c["variant"] == "thermosteric":
        Sh = S[0]
    elif c["variant"] == "halosteric":
        Th = T[0]
    rho = o.wright_density(Th, Sh, pb)
    rho = np.broadcast_to(rho, T.shape)
    ref_masso = np.nansum(rho * vol, axis=(1, 2, 3))
    drho_ref = np.where(~np.isnan(vol), rho - rho0, np.nan)
    dz = o.calc_dz(c["z_l"], c["z_i"], c["deptho"])
    eta_ref = np.where(~np.isnan(vol[0]), (-1.0 / 1035.0) * np.nansum(dz * drho_ref, axis=1), np.nan)

    Td = dT32 if c["variant"] != "halosteric" else dT32[0]
    Sd = dS32 if c["variant"] != "thermosteric" else dS32[0]
    dT, dS = torch.from_numpy(np.ascontiguousarray(Td)).cuda(), torch.from_numpy(np.ascontiguousarray(Sd)).cuda()
    dvol = torch.from_numpy(vol).cuda()
    tag = f"seed {seed}: {c['variant']} {np.dtype(c['dtype']).name} nt={c['nt']} nz={c['nz']} " \
          f"ny={c['ny']} nx={c['nx']} p3d={c['p3d']} f32_mode={mode}"

    got_rho = core.eos_map(dT, dS, pres, f32_mode=mode).cpu().numpy()
    assert_bit_equal(np.broadcast_to(got_rho, T.shape), rho, tag + " rho")

    masso = core.steric_global_masso(dT, dS, dvol, pres, f32_mode=mode).cpu().numpy()
    assert masso.shape == (c["nt"],)
    scale = np.nansum(np.abs(rho * vol), axis=(1, 2, 3))
    assert np.all(np.abs(masso - ref_masso) <= 1e-12 * np.maximum(scale, 1e-300)), tag + " masso"

    rho0m = core.fold_mask(torch.from_numpy(rho0).cuda(), dvol)
    drho, eta = core.steric_local(dT, dS, rho0m, dvol[0], pres, -1.0 / 1035.0,
                                  z_i=c["z_i"], deptho=c["deptho"], f32_mode=mode)
    assert_bit_equal(drho.cpu().numpy(), drho_ref, tag + " delta_rho")
    assert_bit_equal(eta.cpu().numpy(), eta_ref, tag + " eta")


def test_sweep_covers_every_dtype_variant_cell():
    """the draw itself: all 6 dtype x variant cells, both f32 modes for every variant"""
    n = int(os.environ.get("MOMLEVEL_SWEEP_SEEDS", "48"))
    cells = {(np.dtype(c["dtype"]).name, c["variant"]) for c in map(draw, range(min(n, 12)))}
    assert len(cells) == 6
    if n >= 48:
        up = {c["variant"] for c in map(draw, range(n))
              if c["f32_mode"] == "upcast" and c["dtype"] == np.float32}
        assert up == {"steric", "thermosteric", "halosteric"}
