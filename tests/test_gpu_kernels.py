"""GPU: the kernels through the C ABI (momlevel_amd.core -> ctypes -> libmomlevel_hip.so):
edge cases, error behaviour, determinism, and parity at BASELINE.json's sizes.

config 2 (360x576x75, nt=12): element-wise parity with the oracle.
config 3 (1440x1080x75): size-independent properties at the full grid (a reduced number of
time steps keeps the test inside the time budget; bench.py runs all 120).
"""

import numpy as np
import pytest
import torch

from momlevel_amd import _lib, core, engine, parallel, synthetic
from oracle import momlevel_numpy as o
from conftest import assert_bit_equal, assert_rel

pytestmark = pytest.mark.gpu

SEED = synthetic.SEED


def make_case(nt, nz, ny, nx, dtype=torch.float64):
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    shape = (nt, nz, ny, nx)
    T = core.synth_field(shape, dtype, seed=SEED, field_id=1, lo=-2.0, scale=34.0, mask3d=vol0)
    S = core.synth_field(shape, dtype, seed=SEED, field_id=2, lo=30.0, scale=10.0, mask3d=vol0)
    pres = o.pressure_from_depth(g["z_l"])
    return g, vol0, T, S, pres


def host_fields(g, shape, t0=0, dtype=np.float64):
    kw = dict(seed=SEED, mask3d=g["volcello"], t0=t0, dtype=dtype)
    return (synthetic.field_numpy(shape, field_id=1, lo=-2.0, scale=34.0, **kw),
            synthetic.field_numpy(shape, field_id=2, lo=30.0, scale=10.0, **kw))


# ---------------------------------------------------------------------------------------------
def test_synth_generator_replays_in_numpy():
    g, vol0, T, S, _ = make_case(3, 5, 12, 20)
    Tn, Sn = host_fields(g, (3, 5, 12, 20))
    assert_bit_equal(T.cpu().numpy(), Tn)
    assert_bit_equal(S.cpu().numpy(), Sn)
    T32 = core.synth_field((3, 5, 12, 20), torch.float32, seed=SEED, field_id=1, lo=-2.0,
                           scale=34.0, mask3d=vol0)
    assert_bit_equal(T32.cpu().numpy(), host_fields(g, (3, 5, 12, 20), dtype=np.float32)[0])


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (2, 3, 1, 7), (5, 2, 3, 3), (3, 4, 2, 1024),
                                   (2, 2, 17, 130), (9, 3, 64, 64), (16, 1, 1, 4098)])
def test_k1_ragged_shapes(shape):
    """empty-ish, odd and tile-straddling planes; nt around the LDS flush period (8)."""
    nt, nz, ny, nx = shape
    r = np.random.default_rng(nx)
    T = r.uniform(-2, 32, shape)
    S = r.uniform(30, 40, shape)
    vol = r.uniform(1e9, 1e12, shape[1:])
    vol[r.uniform(size=vol.shape) < 0.2] = np.nan
    pres = o.pressure_from_depth(np.linspace(1.0, 5000.0, nz))
    got = core.steric_global_masso(torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda(),
                                   torch.from_numpy(vol).cuda(), pres)
    ref = o.calc_masso(o.calc_rho(T, S, pres), vol)
    assert_rel(got.cpu().numpy(), ref, 1e-12, f"masso {shape}")


def test_all_nan_reductions_are_zero():
    shape = (3, 2, 4, 6)
    T = torch.full(shape, float("nan"), dtype=torch.float64, device="cuda")
    vol = torch.full(shape[1:], float("nan"), dtype=torch.float64, device="cuda")
    got = core.steric_global_masso(T, T, vol, np.array([1e5, 2e5]))
    assert (got.cpu().numpy() == 0.0).all()
    assert core.nansum(vol).item() == 0.0


def test_fast_and_generic_paths_agree():
    """a 16-byte-misaligned view must take the scalar kernels and give the same numbers."""
    g, vol0, T, S, pres = make_case(5, 6, 16, 24)
    n = T.numel()
    buf = torch.empty(n + 1, dtype=torch.float64, device="cuda")
    T_mis = buf[1:].view(T.shape)
    T_mis.copy_(T)
    assert T_mis.data_ptr() % 16 == 8
    a = core.steric_global_masso(T, S, vol0, pres).cpu().numpy()
    b = core.steric_global_masso(T_mis, S, vol0, pres).cpu().numpy()
    assert_rel(b, a, 1e-13, "generic vs fast masso")
    ra = core.eos_map(T, S, pres).cpu().numpy()
    rb = core.eos_map(T_mis, S, pres).cpu().numpy()
    assert_bit_equal(rb, ra, "generic vs fast rho")


def test_strided_time_axis_needs_no_copy():
    g, vol0, T, S, pres = make_case(8, 4, 8, 16)
    a = core.steric_global_masso(T[::2], S[::2], vol0, pres).cpu().numpy()
    b = core.steric_global_masso(T[::2].contiguous(), S[::2].contiguous(), vol0, pres).cpu().numpy()
    assert_bit_equal(a, b)


def test_run_to_run_bit_identical():
    g, vol0, T, S, pres = make_case(9, 10, 48, 64)
    runs = [core.steric_global_masso(T, S, vol0, pres).cpu().numpy() for _ in range(3)]
    assert_bit_equal(runs[1], runs[0])
    assert_bit_equal(runs[2], runs[0])


def test_abi_error_codes():
    g, vol0, T, S, pres = make_case(2, 3, 4, 8)
    with pytest.raises(_lib.MomlevelHipError, match="argument error -3"):
        lib = _lib.load()
        rc = lib.mlx_eos_map(T.data_ptr(), S.data_ptr(), 0, vol0.data_ptr(), 1, 7, 0,
                             2, 3, 32, 96, 96, 0, T.data_ptr(), None)
        _lib.check(rc, "mlx_eos_map")
    lib = _lib.load()
    ws = torch.empty(1, dtype=torch.float64, device="cuda")
    rc = lib.mlx_steric_global(T.data_ptr(), S.data_ptr(), 0, vol0.data_ptr(), vol0.data_ptr(), 1,
                               0, 2, 3, 32, 96, 96, 0, ws.data_ptr(), ws.data_ptr(), 8, None)
    assert rc == -4 and "workspace" in _lib.last_error()
    with pytest.raises(TypeError):
        core.steric_global_masso(T.cpu(), S, vol0, pres)
    with pytest.raises(ValueError):
        core.steric_global_masso(T, S, vol0[:, :2], pres)


# ---------------------------------------------------------------------------------------------
# config 2: OM4 1-degree-like grid 360x576x75, 12 steps -- element-wise parity with the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_config2_full_parity():
    nt, nz, ny, nx = 12, 75, 576, 360
    g, vol0, T, S, pres = make_case(nt, nz, ny, nx)
    masso = core.steric_global_masso(T, S, vol0, pres).cpu().numpy()
    rho0 = core.eos_map(T[0], S[0], pres)
    rho0m = core.fold_mask(rho0, vol0)
    drho, eta = core.steric_local(T, S, rho0m, vol0[0], pres, -1.0 / 1035.0,
                                  z_i=g["z_i"], deptho=g["deptho"])
    ref_masso = np.empty(nt)
    dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"])
    wet3 = ~np.isnan(g["volcello"])
    rho0_ref = None
    for t in range(nt):  # slab by slab: keeps the oracle's temporaries at 124 MB each
        Tn, Sn = host_fields(g, (1, nz, ny, nx), t0=t)
        rho = o.calc_rho(Tn[0], Sn[0], pres)
        if t == 0:
            rho0_ref = rho
            assert_bit_equal(rho0.cpu().numpy(), rho, "rho0")
        ref_masso[t] = o.calc_masso(rho, g["volcello"])
        d = np.where(wet3, rho - rho0_ref, np.nan)
        assert_bit_equal(drho[t].cpu().numpy(), d, f"delta_rho t={t}")
        e = np.where(wet3[0], (-1.0 / 1035.0) * np.nansum(dz * d, axis=0), np.nan)
        assert_bit_equal(eta[t].cpu().numpy(), e, f"eta t={t}")
    assert_rel(masso, ref_masso, 1e-10, "masso")
    assert np.max(np.abs(masso - ref_masso) / ref_masso) < 1e-13  # what is actually achieved


# ---------------------------------------------------------------------------------------------
# config 3 grid (1440x1080x75) at full horizontal/vertical size: size-independent properties
# ---------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_config3_grid_properties():
    nt, nz, ny, nx = 4, 75, 1080, 1440
    g, vol0, T, S, pres = make_case(nt, nz, ny, nx)
    m = core.steric_global_masso(T, S, vol0, pres)
    # determinism
    assert torch.equal(m, core.steric_global_masso(T, S, vol0, pres))
    # linearity in vol0: a power-of-two scale is exact in floating point
    m2 = core.steric_global_masso(T, S, vol0 * 2.0, pres)
    assert torch.equal(m2, m * 2.0)
    # reference-state identity: masso0 == masso(t=0), held-field variants agree at t=0
    _rho0, volo, masso0 = engine.reference_state(T[0], S[0], vol0, pres)
    assert masso0.item() == m[0].item()
    assert core.steric_global_masso(T, S[0], vol0, pres)[0].item() == masso0.item()
    assert core.steric_global_masso(T[0], S, vol0, pres)[0].item() == masso0.item()
    # horizontal tiling (the 2x4 multi-GPU decomposition) sums to the whole
    tot = torch.zeros(nt, dtype=torch.float64, device="cuda")
    vtot = 0.0
    for rank in range(8):
        y0, y1, x0, x1 = synthetic.tile_bounds(ny, nx, rank, 8)
        Tt = T[:, :, y0:y1, x0:x1].contiguous()
        St = S[:, :, y0:y1, x0:x1].contiguous()
        vt = vol0[:, y0:y1, x0:x1].contiguous()
        tot += core.steric_global_masso(Tt, St, vt, pres)
        vtot += core.nansum(vt).item()
        del Tt, St, vt
    assert_rel(tot.cpu().numpy(), m.cpu().numpy(), 1e-12, "tiles vs whole")
    assert_rel(vtot, volo.item(), 1e-12, "volo tiles vs whole")
    # oracle spot check on one time slab of a 135x1440 band (all z): masso band + local columns
    y0, y1 = 400, 535
    Tb = T[2:3, :, y0:y1].contiguous()
    Sb = S[2:3, :, y0:y1].contiguous()
    vb = vol0[:, y0:y1].contiguous()
    Tn, Sn, vn = Tb.cpu().numpy(), Sb.cpu().numpy(), vb.cpu().numpy()
    rho = o.calc_rho(Tn, Sn, pres)
    assert_rel(core.steric_global_masso(Tb, Sb, vb, pres).cpu().numpy(), o.calc_masso(rho, vn),
               1e-12, "band masso")
    rho0m = core.fold_mask(core.eos_map(T[0], S[0], pres), vol0)
    drho, eta = core.steric_local(T[2:3], S[2:3], rho0m, vol0[0], pres, -1.0 / 1035.0,
                                  z_i=g["z_i"], deptho=g["deptho"])
    rho0n = o.calc_rho(T[0, :, y0:y1].cpu().numpy(), S[0, :, y0:y1].cpu().numpy(), pres)
    d = np.where(~np.isnan(vn), rho[0] - rho0n, np.nan)
    assert_bit_equal(drho[0, :, y0:y1].cpu().numpy(), d, "band delta_rho")
    dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"][y0:y1])
    e = np.where(~np.isnan(vn[0]), (-1.0 / 1035.0) * np.nansum(dz * d, axis=0), np.nan)
    assert_bit_equal(eta[0, y0:y1].cpu().numpy(), e, "band eta")
    # eta-only mode (delta_rho store skipped) gives the same eta
    _, eta2 = core.steric_local(T[2:3], S[2:3], rho0m, vol0[0], pres, -1.0 / 1035.0,
                                z_i=g["z_i"], deptho=g["deptho"], want_delta_rho=False)
    assert torch.equal(torch.nan_to_num(eta2), torch.nan_to_num(eta))
    # an explicit dz array instead of the on-the-fly calc_dz
    dzd = core.calc_dz(torch.from_numpy(g["z_i"]).cuda(), torch.from_numpy(g["deptho"]).cuda())
    _, eta3 = core.steric_local(T[2:3], S[2:3], rho0m, vol0[0], pres, -1.0 / 1035.0, dz=dzd,
                                want_delta_rho=False)
    assert torch.equal(torch.nan_to_num(eta3), torch.nan_to_num(eta))


def test_single_process_tile_pipeline_matches_labelled_api():
    """parallel.steric_global_tile with world_size 1 == the public steric(domain='global')."""
    g, vol0, T, S, pres = make_case(5, 8, 16, 24)
    out = parallel.steric_global_tile(T, S, vol0, g["areacello"], pres)
    vol4 = np.broadcast_to(g["volcello"], T.shape).copy()
    ref, refst = o.steric(T.cpu().numpy(), S.cpu().numpy(), vol4, g["areacello"], g["z_l"],
                          domain="global")
    assert out["eta"][0] == 0.0
    assert_rel(out["masso"], ref["masso"], 1e-10)
    assert_rel(out["reference_height"], ref["reference_height"], 1e-10)
    assert np.allclose(out["expansion_coeff"], ref["expansion_coeff"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_skip_dry_is_bit_identical(dtype):
    """MLX_FLAG_SKIP_DRY only removes loads of cells that contribute exactly nothing."""
    g, vol0, T, S, pres = make_case(20, 12, 64, 96, dtype)
    # inconsistent masks on purpose: finite theta/S under NaN vol0 and NaN theta over wet cells
    T = T.clone()
    T[:, 0, :4, :8] = 3.0
    T[3, 5, 10:20, 30:50] = float("nan")
    for Tv, Sv in ((T, S), (T, S[0]), (T[0], S)):
        dense = core.steric_global_masso(Tv, Sv, vol0, pres, skip_dry=False)
        sparse = core.steric_global_masso(Tv, Sv, vol0, pres, skip_dry=True)
        assert torch.equal(dense, sparse)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_skip_dry_local_is_bit_identical(dtype):
    g, vol0, T, S, pres = make_case(19, 12, 64, 96, dtype)
    T = T.clone()
    T[:, 0, :4, :8] = 3.0  # finite theta under NaN vol0
    rho0m = core.fold_mask(core.eos_map(T[0], S[0], pres), vol0)
    for Tv, Sv in ((T, S), (T, S[0]), (T[0], S)):
        outs = []
        for skip in (False, True):
            d, e = core.steric_local(Tv, Sv, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=g["z_i"],
                                     deptho=g["deptho"], skip_dry=skip)
            outs.append((d.cpu().numpy(), e.cpu().numpy()))
        assert_bit_equal(outs[1][0], outs[0][0], "delta_rho")
        assert_bit_equal(outs[1][1], outs[0][1], "eta")
        assert np.array_equal(outs[1][0].view(np.int64), outs[0][0].view(np.int64))  # NaN payloads too


def test_local_tiles_need_no_exchange():
    """parallel.steric_local_tile on the 2x4 tiles of a grid == the whole grid, bit for bit."""
    nt, nz, ny, nx = 5, 9, 32, 64
    g, vol0, T, S, pres = make_case(nt, nz, ny, nx)
    zi, dep = g["z_i"], torch.from_numpy(g["deptho"]).cuda()
    for variant in ("steric", "thermosteric", "halosteric"):
        dfull, efull = parallel.steric_local_tile(T, S, vol0, pres, zi, dep, variant=variant)
        for rank in range(8):
            y0, y1, x0, x1 = synthetic.tile_bounds(ny, nx, rank, 8)
            d, e = parallel.steric_local_tile(
                T[:, :, y0:y1, x0:x1].contiguous(), S[:, :, y0:y1, x0:x1].contiguous(),
                vol0[:, y0:y1, x0:x1].contiguous(), pres, zi, dep[y0:y1, x0:x1].contiguous(),
                variant=variant)
            assert_bit_equal(d.cpu().numpy(), dfull[:, :, y0:y1, x0:x1].cpu().numpy(), "delta_rho tile")
            assert_bit_equal(e.cpu().numpy(), efull[:, y0:y1, x0:x1].cpu().numpy(), "eta tile")


def test_empty_record_is_rejected_cleanly():
    """nt = 0 (an empty time axis) is an error, never a launch: ValueError from the Python layer,
    MLX_E_SHAPE from the ABI itself."""
    g, vol0, T, S, pres = make_case(2, 3, 4, 8)
    with pytest.raises(ValueError, match="empty field"):
        core.steric_global_masso(T[:0], S[:0], vol0, pres)
    with pytest.raises(ValueError, match="empty field"):
        core.eos_map(T[:0], S[:0], pres)
    lib = _lib.load()
    ws = torch.empty(16, dtype=torch.float64, device="cuda")
    rc = lib.mlx_steric_global(T.data_ptr(), S.data_ptr(), 0, vol0.data_ptr(), vol0.data_ptr(), 1,
                               0, 0, 3, 32, 96, 96, 0, ws.data_ptr(), ws.data_ptr(), 128, None)
    assert rc == -2  # MLX_E_SHAPE


def test_very_long_record_on_a_tiny_grid():
    """nt = 100 000 (grid.z = 3125 time chunks in K1, 6250 in K2) on a 2x3x4 grid."""
    nt, nz, ny, nx = 100_000, 2, 3, 4
    r = np.random.default_rng(9)
    T = r.uniform(-2, 32, (nt, nz, ny, nx))
    S = r.uniform(30, 40, (nt, nz, ny, nx))
    vol = r.uniform(1e9, 1e11, (nz, ny, nx))
    vol[0, 0, 0] = np.nan
    pres = np.array([1.2e5, 9.0e6])
    dT, dS, dvol = (torch.from_numpy(a).cuda() for a in (T, S, vol))
    rho = o.calc_rho(T, S, pres)
    masso = core.steric_global_masso(dT, dS, dvol, pres).cpu().numpy()
    assert_rel(masso, o.calc_masso(rho, vol), 1e-13, "masso")
    rho0m = core.fold_mask(torch.from_numpy(rho[0]).cuda(), dvol)
    z_i = np.array([0.0, 12.0, 900.0])
    dep = r.uniform(0.0, 1000.0, (ny, nx))
    drho, eta = core.steric_local(dT, dS, rho0m, dvol[0], pres, -1.0 / 1035.0, z_i=z_i, deptho=dep)
    d = np.where(~np.isnan(vol), rho - rho[0], np.nan)
    assert_bit_equal(drho.cpu().numpy(), d, "delta_rho")
    dz = o.calc_dz(0.5 * (z_i[1:] + z_i[:-1]), z_i, dep)
    e = np.where(~np.isnan(vol[0]), (-1.0 / 1035.0) * np.nansum(dz * d, axis=1), np.nan)
    assert_bit_equal(eta.cpu().numpy(), e, "eta")


def test_launch_targets_the_operands_device():
    """kernels go to the stream of the device that owns the tensors, whatever torch's
    current device / stream is.  With one GPU the check is on the stream: operands produced on a
    side stream, launch under that stream, result identical; with >1 GPU also a non-current device."""
    g, vol0, T, S, pres = make_case(5, 3, 8, 16)
    base = core.steric_global_masso(T, S, vol0, pres).cpu().numpy()
    side = torch.cuda.Stream(device=T.device)
    side.wait_stream(torch.cuda.current_stream(T.device))
    with torch.cuda.stream(side):
        got = core.steric_global_masso(T, S, vol0, pres)
    side.synchronize()
    assert np.array_equal(got.cpu().numpy(), base)
    if torch.cuda.device_count() > 1:
        other = torch.device("cuda", 1)
        with torch.cuda.device(0):
            got = core.steric_global_masso(T.to(other), S.to(other), vol0.to(other), pres)
            assert got.device == other
            torch.cuda.synchronize(other)
        assert np.array_equal(got.cpu().numpy(), base)
        with pytest.raises(ValueError):
            core.steric_global_masso(T, S.to(other), vol0, pres)


# ---------------------------------------------------------------------------------------------
# round 2: one-pass decomposition (all variants + heat), fused arithmetic, time-dependent pressure
# ---------------------------------------------------------------------------------------------
def _case_fields(shape, dtype, seed=3):
    nt, nz, ny, nx = shape
    g = synthetic.make_grid(ny, nx, nz)
    r = np.random.default_rng(seed)
    mask = np.isnan(g["volcello"])
    dT, dS = dtype if isinstance(dtype, tuple) else (dtype, dtype)  # (theta, salinity) dtypes
    T = np.where(mask[None], np.nan, r.uniform(-2, 32, shape)).astype(dT)
    S = np.where(mask[None], np.nan, r.uniform(30, 40, shape)).astype(dS)
    return g, T, S


T32_S64, T64_S32 = (np.float32, np.float64), (np.float64, np.float32)  # fields of different dtypes


@pytest.mark.parametrize("skip_dry", [False, True])
@pytest.mark.parametrize("dtype,f32_mode", [(np.float64, "faithful"), (np.float32, "faithful"),
                                            (np.float32, "upcast"), (T32_S64, "faithful"),
                                            (T64_S32, "faithful"), (T32_S64, "upcast")])
@pytest.mark.parametrize("shape", [(37, 5, 12, 40), (9, 3, 7, 9), (3, 4, 2, 1024)])
def test_decomposition_rows_equal_single_variant_launches(shape, dtype, f32_mode, skip_dry):
    """mlx_steric_global_decomp: rows 0-2 bit-identical to three mlx_steric_global calls (same
    tiling, same order), row 3 = sum(theta*vol0) vs numpy; 12x40 / 2x1024 planes take the dwordx4
    kernel, 7x9 the scalar twin; nt=37 spans two 32-step time chunks.  A pair of dtypes = theta and
    salinity stored with different precisions (MLX_DTYPE_T32_S64 / _T64_S32: numpy's promotion per
    sub-expression, exact arithmetic whatever the default policy says; "upcast": both widened)."""
    g, T, S = _case_fields(shape, dtype)
    dT, dS = torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda()
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    pres = o.pressure_from_depth(g["z_l"])
    kw = dict(f32_mode=f32_mode, skip_dry=skip_dry)
    rows = core.steric_global_decomp(dT, dS, dT[0], dS[0], vol0, pres, **kw).cpu().numpy()
    assert rows.shape == (4, shape[0])
    assert np.array_equal(rows[0], core.steric_global_masso(dT, dS, vol0, pres, **kw).cpu().numpy())
    assert np.array_equal(rows[1], core.steric_global_masso(dT, dS[0], vol0, pres, **kw).cpu().numpy())
    assert np.array_equal(rows[2], core.steric_global_masso(dT[0], dS, vol0, pres, **kw).cpu().numpy())
    assert rows[0][0] == rows[1][0] == rows[2][0]  # t=0: every variant sees (theta0, S0)
    heat = np.nansum(T.astype(np.float64) * g["volcello"], axis=(1, 2, 3))
    assert_rel(rows[3], heat, 1e-12, "heat integrand")
    # and against the oracle (the single launches are checked elsewhere; this closes the loop)
    Tn, Sn = (T.astype(np.float64), S.astype(np.float64)) if f32_mode == "upcast" else (T, S)
    pb = pres[:, None, None]
    for row, (a, b) in zip(rows[:3], [(Tn, Sn), (Tn, Sn[0]), (Tn[0], Sn)]):
        ref = np.nansum(np.broadcast_to(o.wright_density(a, b, pb), T.shape) * g["volcello"],
                        axis=(1, 2, 3))
        assert_rel(row, ref, 1e-12, "masso vs oracle")


def test_decomposition_with_a_separate_reference_state():
    """T0/S0 need not be time level 0 of the record (a reference read from an earlier run)"""
    g, T, S = _case_fields((6, 4, 8, 16), np.float64)
    _, T0, S0 = _case_fields((1, 4, 8, 16), np.float64, seed=99)
    dT, dS = torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda()
    dT0, dS0 = torch.from_numpy(T0[0]).cuda(), torch.from_numpy(S0[0]).cuda()
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    pres = o.pressure_from_depth(g["z_l"])
    rows = core.steric_global_decomp(dT, dS, dT0, dS0, vol0, pres).cpu().numpy()
    assert np.array_equal(rows[1], core.steric_global_masso(dT, dS0, vol0, pres).cpu().numpy())
    assert np.array_equal(rows[2], core.steric_global_masso(dT0, dS, vol0, pres).cpu().numpy())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(19, 6, 12, 40), (5, 3, 7, 9)])
def test_fused_arithmetic_meets_the_parity_gate(shape, dtype):
    """MLX_FLAG_FMA (opt-in): rho and masso within 1e-10 relative of the oracle (north_star's fp64
    tolerance), delta_rho / eta within 1e-10 * max|ref|; and the fused kernels agree with EACH
    OTHER bit for bit: masso(t=0) identical across variants and equal to the reference state's,
    delta_rho(t=0) exactly 0, K0 == K2 + rho0."""
    g, T, S = _case_fields(shape, dtype)
    # float32 input (numpy's mixed precision, the default f32_mode): the fused policy keeps numpy's
    # float32 polynomial and fuses the float64 tail only -- the oracle is numpy ON THE float32 ARRAYS
    Tn, Sn = T, S
    dT, dS = torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda()
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    pres = o.pressure_from_depth(g["z_l"])
    pb = pres[:, None, None]
    rho_ref = o.wright_density(Tn, Sn, pb)
    rho = core.eos_map(dT, dS, pres, arith="fused").cpu().numpy()
    assert_rel(rho, rho_ref, 1e-10, "fused rho")
    m = ~np.isnan(rho_ref)
    assert np.max(np.abs(rho[m] - rho_ref[m]) / rho_ref[m]) < 2e-15  # in fact a few ulp
    assert not np.array_equal(rho[m], core.eos_map(dT, dS, pres, arith="exact").cpu().numpy()[m])
    kw = dict(arith="fused")
    rows = core.steric_global_decomp(dT, dS, dT[0], dS[0], vol0, pres, **kw).cpu().numpy()
    singles = [core.steric_global_masso(dT, dS, vol0, pres, **kw),
               core.steric_global_masso(dT, dS[0], vol0, pres, **kw),
               core.steric_global_masso(dT[0], dS, vol0, pres, **kw)]
    for row, single, (a, b) in zip(rows, singles, [(Tn, Sn), (Tn, Sn[0]), (Tn[0], Sn)]):
        assert np.array_equal(row, single.cpu().numpy())
        ref = np.nansum(np.broadcast_to(o.wright_density(a, b, pb), T.shape) * g["volcello"],
                        axis=(1, 2, 3))
        assert_rel(row, ref, 1e-10, "fused masso")
    masso0 = core.steric_global_masso(dT[:1], dS[:1], vol0, pres, **kw).cpu().numpy()[0]
    assert rows[0][0] == rows[1][0] == rows[2][0] == masso0
    # local: rho0 from the fused K0, so delta_rho(t=0) == 0 exactly, for every variant
    rho0 = core.eos_map(dT[0], dS[0], pres, arith="fused")
    rho0m = core.fold_mask(rho0, vol0)
    drho_ref = np.where(~np.isnan(g["volcello"]), rho_ref - rho_ref[0], np.nan)
    dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"])
    for (a, b, an, bn) in [(dT, dS, Tn, Sn), (dT, dS[0], Tn, Sn[0]), (dT[0], dS, Tn[0], Sn)]:
        drho, eta = core.steric_local(a, b, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=g["z_i"],
                                      deptho=g["deptho"], **kw)
        drho, eta = drho.cpu().numpy(), eta.cpu().numpy()
        wet = ~np.isnan(g["volcello"])
        assert np.all(drho[0][wet] == 0.0)
        r = np.broadcast_to(o.wright_density(an, bn, pb), T.shape)
        dref = np.where(wet, r - rho_ref[0], np.nan)
        eref = np.where(wet[0], (-1.0 / 1035.0) * np.nansum(dz * dref, axis=1), np.nan)
        assert np.array_equal(np.isnan(drho), np.isnan(dref))
        assert np.nanmax(np.abs(drho - dref)) <= 1e-10 * np.nanmax(np.abs(dref))
        assert np.array_equal(np.isnan(eta), np.isnan(eref))
        assert np.nanmax(np.abs(eta - eref)) <= 1e-10 * np.nanmax(np.abs(eref))


def test_default_arithmetic_policy(monkeypatch):
    """Round 3: the global sums (K1) default to fused arithmetic -- within the
    north-star gate (<= 1e-10 relative) of the oracle, masso(t=0) == masso0 still exact -- while K0
    and K2 (pointwise outputs) keep numpy's exact arithmetic.  MOMLEVEL_AMD_ARITH overrides."""
    monkeypatch.delenv("MOMLEVEL_AMD_ARITH", raising=False)
    g, vol0, T, S, pres = make_case(4, 3, 8, 16)
    exact = core.steric_global_masso(T, S, vol0, pres, arith="exact").cpu().numpy()
    fused = core.steric_global_masso(T, S, vol0, pres, arith="fused").cpu().numpy()
    default = core.steric_global_masso(T, S, vol0, pres).cpu().numpy()
    assert np.array_equal(default, fused)
    assert_rel(fused, exact, 1e-12)
    ref = o.calc_masso(o.calc_rho(T.cpu().numpy(), S.cpu().numpy(), pres), g["volcello"])
    assert_rel(default, ref, 1e-10, "default (fused) masso vs the oracle")
    rows = core.steric_global_decomp(T, S, T[0], S[0], vol0, pres).cpu().numpy()
    assert np.array_equal(rows, core.steric_global_decomp(T, S, T[0], S[0], vol0, pres,
                                                          arith="fused").cpu().numpy())
    assert rows[0, 0] == rows[1, 0] == rows[2, 0] == default[0]
    # pointwise outputs stay bit-identical to numpy by default
    rho = core.eos_map(T, S, pres).cpu().numpy()
    assert_bit_equal(rho, o.calc_rho(T.cpu().numpy(), S.cpu().numpy(), pres), "default K0")
    # float32 inputs: the same policy -- K1 fused (float32 polynomial as numpy rounds it, fused
    # float64 tail: a few ulp from the exact kernel), K0 / K2 exact
    T32, S32 = T.float(), S.float()
    m32 = core.steric_global_masso(T32, S32, vol0, pres).cpu().numpy()
    assert np.array_equal(m32, core.steric_global_masso(T32, S32, vol0, pres, arith="fused").cpu().numpy())
    assert_rel(m32, core.steric_global_masso(T32, S32, vol0, pres, arith="exact").cpu().numpy(), 1e-13)
    ref32 = o.calc_masso(o.calc_rho(T32.cpu().numpy(), S32.cpu().numpy(), pres), g["volcello"])
    assert_rel(m32, ref32, 1e-12, "default float32 masso vs numpy on the float32 arrays")
    assert_bit_equal(core.eos_map(T32, S32, pres).cpu().numpy(),
                     o.calc_rho(T32.cpu().numpy(), S32.cpu().numpy(), pres), "default K0, float32")
    assert core.arith_default("k1", torch.float64) == core.arith_default("k1", torch.float32) == "fused"
    assert core.arith_default("k0", torch.float64) == core.arith_default("k2", torch.float64) == "exact"
    # the environment overrides the policy for every kernel
    monkeypatch.setenv("MOMLEVEL_AMD_ARITH", "exact")
    assert np.array_equal(core.steric_global_masso(T, S, vol0, pres).cpu().numpy(), exact)
    monkeypatch.setenv("MOMLEVEL_AMD_ARITH", "fused")
    assert np.array_equal(core.steric_global_masso(T, S, vol0, pres).cpu().numpy(), fused)
    assert not np.array_equal(core.eos_map(T, S, pres).cpu().numpy()[~np.isnan(rho)], rho[~np.isnan(rho)])
    monkeypatch.setenv("MOMLEVEL_AMD_ARITH", "sloppy")
    with pytest.raises(ValueError):
        core.steric_global_masso(T, S, vol0, pres)


def test_fused_float32_against_the_reference_vectors(wright_vectors):
    """MLX_FLAG_FMA on float32 theta/S.  With numpy's mixed precision (MLX_DTYPE_F32, the
    default) the float32 polynomial is evaluated exactly as numpy rounds it and only the float64
    tail is fused, so the density is a few float64 ulp from what the reference module computed on
    the same float32 arrays (tests/golden/wright_vectors.npz).  With f32_mode="upcast" it is float64
    arithmetic on the float32 VALUES: a few ulp from that, and the float32 rounding of al0, p0, lam
    (~1e-7) away from the reference -- pinned at 1e-10 < err < 2e-7."""
    v = wright_vectors
    T = torch.from_numpy(v["f32_T"]).cuda()
    S = torch.from_numpy(v["f32_S"]).cuda()
    assert T.dtype == torch.float32
    ref = v["f32_density"]
    m = np.isfinite(ref)
    p = v["blk_p"].reshape(-1)
    fused = core.eos_map(T, S, p, arith="fused").cpu().numpy()
    assert np.array_equal(np.isnan(fused), np.isnan(ref))
    assert np.max(np.abs(fused[m] - ref[m]) / np.abs(ref[m])) < 1e-15
    assert not np.array_equal(fused[m], ref[m])  # (it IS the fused kernel)
    upf = core.eos_map(T, S, p, arith="fused", f32_mode="upcast").cpu().numpy()
    err = np.max(np.abs(upf[m] - ref[m]) / np.abs(ref[m]))
    assert 1e-10 < err < 2e-7, err
    up = o.wright_density(v["f32_T"].astype(np.float64), v["f32_S"].astype(np.float64), v["blk_p"])
    assert_rel(upf, up, 1e-14, "upcast + fused vs float64 arithmetic on the float32 values")
    # held-field operands (thermosteric / halosteric) through K2, fused: delta_rho + rho0 within
    # a few ulp of the reference module's broadcast float32 outputs
    nt, nz, ny, nx = v["f32_T"].shape
    vol = torch.ones((nz, ny, nx), dtype=torch.float64, device="cuda")
    rho0 = core.eos_map(T[0], S[0], p, arith="fused")
    z_i = np.concatenate([[0.0], np.cumsum(np.full(nz, 10.0))])
    for a_, b_, key in ((T, S[0], "f32_density_heldS"), (T[0], S, "f32_density_heldT")):
        drho, _ = core.steric_local(a_, b_, core.fold_mask(rho0, vol), vol[0], p, -1.0 / 1035.0,
                                    z_i=z_i, deptho=np.full((ny, nx), 1e4), arith="fused")
        got = drho.cpu().numpy() + rho0.cpu().numpy()
        assert np.max(np.abs(got - v[key]) / v[key]) < 1e-15, key


@pytest.mark.parametrize("t_chunk", [8, 16, 64, 2040])
def test_time_chunk_hint_never_changes_a_result(t_chunk):
    g, vol0, T, S, pres = make_case(70, 3, 8, 32)
    base = core.steric_global_masso(T, S, vol0, pres).cpu().numpy()
    assert np.array_equal(core.steric_global_masso(T, S, vol0, pres, t_chunk=t_chunk).cpu().numpy(), base)
    assert np.array_equal(core.steric_global_masso(T, S[0], vol0, pres, t_chunk=t_chunk).cpu().numpy(),
                          core.steric_global_masso(T, S[0], vol0, pres).cpu().numpy())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_time_dependent_pressure_in_the_fused_kernels(dtype):
    """MLX_P_FULL4D in K1 / K2 (a patm DataArray with a time dimension, steric.py:58-60,96)"""
    shape = (5, 4, 6, 10)
    g, T, S = _case_fields(shape, dtype)
    r = np.random.default_rng(8)
    patm = 101325.0 + r.normal(0.0, 800.0, (shape[0], 1, shape[2], shape[3]))
    pres = (g["z_l"] * 1.0e4)[None, :, None, None] + patm
    dT, dS = torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda()
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    rho = o.wright_density(T, S, pres)
    assert_bit_equal(core.eos_map(dT, dS, pres).cpu().numpy(), rho, "K0 4-D pressure")
    masso = core.steric_global_masso(dT, dS, vol0, pres).cpu().numpy()
    assert_rel(masso, np.nansum(rho * g["volcello"], axis=(1, 2, 3)), 1e-12, "K1 4-D pressure")
    rho0 = o.wright_density(T[0], S[0], (g["z_l"] * 1.0e4 + 101325.0)[:, None, None])
    rho0m = core.fold_mask(torch.from_numpy(rho0).cuda(), vol0)
    drho, eta = core.steric_local(dT, dS, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=g["z_i"],
                                  deptho=g["deptho"])
    dref = np.where(~np.isnan(g["volcello"]), rho - rho0, np.nan)
    dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"])
    eref = np.where(~np.isnan(g["volcello"][0]), (-1.0 / 1035.0) * np.nansum(dz * dref, axis=1), np.nan)
    assert_bit_equal(drho.cpu().numpy(), dref, "K2 4-D pressure delta_rho")
    assert_bit_equal(eta.cpu().numpy(), eref, "K2 4-D pressure eta")
    rows = core.steric_global_decomp(dT, dS, dT[0], dS[0], vol0, pres).cpu().numpy()
    assert np.array_equal(rows[0], masso)
    assert_rel(rows[1], np.nansum(o.wright_density(T, S[0], pres) * g["volcello"], axis=(1, 2, 3)),
               1e-12, "thermo 4-D pressure")


@pytest.mark.parametrize("skip_dry", [False, True])
@pytest.mark.parametrize("dtype,f32_mode", [(np.float64, "faithful"), (np.float32, "faithful"),
                                            (np.float32, "upcast")])
def test_pressure_field_takes_the_fast_kernels(dtype, f32_mode, skip_dry):
    """MLX_P_FULL3D -- `patm` as a (yh,xh) DataArray makes the pressure a (z,y,x) FIELD
    (steric.py:58-60,96) -- runs on the 16-byte-load kernels since round 4 (template argument P3D:
    each thread keeps / reads its cells' own pressures), not on the scalar twins: K0, K1 and K2,
    every single variant, exact and fused, against numpy; the scalar twin (forced by an unaligned
    pressure view) must agree bit for bit pointwise and to 1e-12 in the sums; masso(t=0) == masso0."""
    shape = (19, 5, 12, 40)  # nt=19: a ragged K2 time block for every blocking (6, 8, 12, 16)
    nt, nz, ny, nx = shape
    g, T, S = _case_fields(shape, dtype)
    r = np.random.default_rng(21)
    p3 = (g["z_l"] * 1.0e4)[:, None, None] + (101325.0 + r.normal(0.0, 700.0, (ny, nx)))[None]
    dT, dS = torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda()
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    dp = torch.from_numpy(p3).cuda()
    raw = torch.empty(p3.size + 1, dtype=torch.float64, device="cuda")
    dp_odd = raw[1:].view(nz, ny, nx)  # 8 bytes off a 16-byte boundary: the scalar twin
    dp_odd.copy_(dp)
    kw = dict(f32_mode=f32_mode, skip_dry=skip_dry)
    if f32_mode == "upcast":
        T, S = T.astype(np.float64), S.astype(np.float64)
    rho = o.wright_density(T, S, p3[None])
    # K0
    got = core.eos_map(dT, dS, dp, f32_mode=f32_mode)
    assert_bit_equal(got.cpu().numpy(), rho, "K0, pressure field")
    assert torch.equal(core.eos_map(dT, dS, dp_odd, f32_mode=f32_mode).nan_to_num(-1.0),
                       got.nan_to_num(-1.0))
    fused = core.eos_map(dT, dS, dp, f32_mode=f32_mode, arith="fused").cpu().numpy()
    assert_rel(fused, rho, 1e-12, "K0 fused, pressure field")
    # K1: every single variant, both policies
    tin = "double" if dtype == np.float64 else "float"
    for name, Tv, Sv, Tn, Sn in (("steric", dT, dS, T, S), ("thermo", dT, dS[0], T, S[0]),
                                 ("halo", dT[0], dS, T[0], S)):
        ref = np.nansum(o.wright_density(Tn, Sn, p3[None] if Tn.ndim == 4 or Sn.ndim == 4 else p3)
                        * g["volcello"], axis=(1, 2, 3))
        for arith, tol in (("exact", 1e-12), ("fused", 1e-10)):
            m = core.steric_global_masso(Tv, Sv, vol0, dp, arith=arith, **kw)
            kern = _lib.last_kernel()
            args = kern[kern.index("<") + 1:-1].split(",")
            # <type, cells per pack, packs, variant, mode, GENERIC, skip, fma, P3D>
            assert kern.startswith(f"k_steric_global<{tin},") and len(args) == 9, kern
            assert args[5] == "false" and args[8] == "true", kern  # the fast kernel, P3D form
            assert_rel(m.cpu().numpy(), ref, tol, f"K1 {name} {arith}, pressure field")
            m_odd = core.steric_global_masso(Tv, Sv, vol0, dp_odd, arith=arith, **kw)
            assert len(_lib.last_kernel().split(",")) == 8  # the scalar twin has no P3D form
            assert_rel(m_odd.cpu().numpy(), m.cpu().numpy(), 1e-12, "scalar twin vs fast kernel")
            m0 = core.steric_global_masso(dT[:1], dS[:1], vol0, dp, arith=arith, **kw)
            if name == "steric":
                assert m0[0] == m[0]  # the reference state's masso0 through the same kernel
    # K2: delta_rho and eta of every variant, bit for bit; eta-only launches agree
    rho0 = o.wright_density(T[0], S[0], p3)
    rho0m = core.fold_mask(core.eos_map(dT[0], dS[0], dp, f32_mode=f32_mode), vol0)
    dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"])
    for name, Tv, Sv, Tn, Sn in (("steric", dT, dS, T, S), ("thermo", dT, dS[0], T, S[:1]),
                                 ("halo", dT[0], dS, T[:1], S)):
        rv = o.wright_density(Tn, Sn, p3[None])
        dref = np.where(~np.isnan(g["volcello"]), rv - rho0, np.nan)
        eref = np.where(~np.isnan(g["volcello"][0]),
                        (-1.0 / 1035.0) * np.nansum(dz * dref, axis=1), np.nan)
        drho, eta = core.steric_local(Tv, Sv, rho0m, vol0[0], dp, -1.0 / 1035.0, z_i=g["z_i"],
                                      deptho=g["deptho"], **kw)
        kern = _lib.last_kernel()
        args = kern[kern.index("<") + 1:-1].split(",")
        assert kern.startswith(f"k_steric_local<{tin},") and len(args) == 9, kern
        assert args[5] == "false" and args[8] == "true", kern
        assert_bit_equal(drho.cpu().numpy(), dref, f"K2 {name} delta_rho, pressure field")
        assert_bit_equal(eta.cpu().numpy(), eref, f"K2 {name} eta, pressure field")
        _, eta_only = core.steric_local(Tv, Sv, rho0m, vol0[0], dp, -1.0 / 1035.0, z_i=g["z_i"],
                                        deptho=g["deptho"], want_delta_rho=False, **kw)
        assert torch.equal(eta_only.nan_to_num(-1.0), eta.nan_to_num(-1.0))
        d_odd, e_odd = core.steric_local(Tv, Sv, rho0m, vol0[0], dp_odd, -1.0 / 1035.0,
                                         z_i=g["z_i"], deptho=g["deptho"], **kw)
        assert len(_lib.last_kernel().split(",")) == 8
        assert torch.equal(d_odd.nan_to_num(-1.0), drho.nan_to_num(-1.0))
        assert torch.equal(e_odd.nan_to_num(-1.0), eta.nan_to_num(-1.0))
    # the all-variants kernels keep the scalar twin for a pressure field: rows equal the launches above
    rows = core.steric_global_decomp(dT, dS, dT[0], dS[0], vol0, dp, arith="exact", **kw)
    assert_rel(rows[0].cpu().numpy(),
               core.steric_global_masso(dT, dS, vol0, dp, arith="exact", **kw).cpu().numpy(), 1e-12,
               "one-pass (scalar twin) vs fast kernel")


def test_stream_probe_adds():
    a = torch.rand(4096 * 6, dtype=torch.float64, device="cuda")
    b = torch.rand(4096 * 6, dtype=torch.float64, device="cuda")
    assert torch.equal(core.stream_probe(a, b), a + b)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("two", [False, True])
def test_stream_probe_mix(dtype, two):
    """the probes with the other local passes' read:write mixes (mlx_stream_probe_mix): one or two
    streams of either dtype in, one float64 stream out, or read-only (nothing written)"""
    n = 4096 * 6 + 12  # whole 16-byte packs, ragged against the block size
    a = torch.rand(n, dtype=dtype, device="cuda")
    b = torch.rand(n, dtype=dtype, device="cuda") if two else None
    want = a.double() + (b.double() if two else 0.0)
    assert torch.equal(core.stream_probe_mix(a, b), want)
    keep = torch.full((1,), 7.0, dtype=torch.float64, device="cuda")
    assert core.stream_probe_mix(a, b, out=keep, write=False) is keep and keep.item() == 7.0
    with pytest.raises(core.MomlevelHipError):
        core.stream_probe_mix(a[:n - 1], None)  # not a whole number of packs


def test_valu_probe_reports_its_instruction_count():
    """mlx_valu_probe: 8192 blocks x 256 threads x 8 fma chains x iters lane-instructions, at a rate
    of the order of the chip's float64 vector peak (256 CUs x 4 SIMDs x 16 lanes per clock)"""
    assert core.valu_probe(16) == 8192 * 256 * 8 * 16
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = core.valu_probe(4096)  # ~2 ms: long enough for the rate to mean something
    e1.record()
    torch.cuda.synchronize()
    rate = n / (e0.elapsed_time(e1) * 1e-3)
    assert 5e12 < rate < 6e13, rate


# ---------------------------------------------------------------------------------------------
# full-size launches (BASELINE.json configs[2] and configs[4]); memory is freed between them
# ---------------------------------------------------------------------------------------------
def _free_hbm():
    import gc

    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.timeout(1200)
def test_config3_full_launch_every_time_chunk():
    """The headline launch itself: 1440x1080x75, 120 steps resident (224 GB), ONE K1 launch of
    4 time chunks (blockIdx.z) x 75 levels x 760 tiles.  A whole-grid time slab from EVERY chunk
    -- first, interior and last steps -- is compared with the oracle; round 1 only checked chunk 0
    at this size."""
    _free_hbm()
    free, _ = torch.cuda.mem_get_info()
    nt, nz, ny, nx = 120, 75, 1080, 1440
    if free < 2 * nt * nz * ny * nx * 8 + (20 << 30):
        pytest.skip("needs 244 GB of free HBM")
    g, vol0, T, S, pres = make_case(nt, nz, ny, nx)
    m = core.steric_global_masso(T, S, vol0, pres, skip_dry=False)
    assert torch.equal(m, core.steric_global_masso(T, S, vol0, pres, skip_dry=True))
    m = m.cpu().numpy()
    for t in (0, 31, 45, 64, 95, 96, 119):  # chunks 0,0,1,2,2,3,3
        rho = o.calc_rho(T[t].cpu().numpy(), S[t].cpu().numpy(), pres)
        ref = o.calc_masso(rho, g["volcello"])
        assert abs(m[t] - ref) <= 1e-12 * abs(ref), f"t={t}: {m[t]!r} vs {ref!r}"
        del rho
    # the chunked, tiled walk of the same record (what --gpus N runs) gives the same numbers
    out = parallel.steric_global_tile_streamed((T, S), vol0, g["areacello"], pres, steps=50,
                                               skip_dry=False)["steric"]
    assert np.array_equal(out["masso"], m) and out["eta"][0] == 0.0
    del T, S
    _free_hbm()


@pytest.mark.timeout(1200)
def test_config5_f32_properties():
    """BASELINE.json configs[4]: float32 theta/S at 0.25 degree, steric + thermosteric +
    halosteric (+ heat content) from ONE pass.  Size-independent properties on the full grid plus an
    oracle check of one whole time slab per variant, in numpy's float32 mixed precision."""
    _free_hbm()
    nt, nz, ny, nx = 8, 75, 1080, 1440
    g, vol0, T, S, pres = make_case(nt, nz, ny, nx, dtype=torch.float32)
    rows = core.steric_global_decomp(T, S, T[0], S[0], vol0, pres)
    # determinism; dry-line skipping changes nothing
    assert torch.equal(rows, core.steric_global_decomp(T, S, T[0], S[0], vol0, pres))
    assert torch.equal(rows, core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, skip_dry=False))
    # one pass == three launches, bit for bit; masso0 == masso(t=0) for all three variants
    assert torch.equal(rows[0], core.steric_global_masso(T, S, vol0, pres))
    assert torch.equal(rows[1], core.steric_global_masso(T, S[0], vol0, pres))
    assert torch.equal(rows[2], core.steric_global_masso(T[0], S, vol0, pres))
    _rho0, volo, masso0 = engine.reference_state(T[0], S[0], vol0, pres)
    assert rows[0, 0].item() == rows[1, 0].item() == rows[2, 0].item() == masso0.item()
    # linearity in vol0 (power of two: exact)
    assert torch.equal(core.steric_global_decomp(T, S, T[0], S[0], vol0 * 4.0, pres), rows * 4.0)
    rows = rows.cpu().numpy()
    # oracle, one whole slab (t=5): numpy on float32 arrays = the reference's arithmetic
    t = 5
    Tn, Sn, T0n, S0n = (x.cpu().numpy() for x in (T[t], S[t], T[0], S[0]))
    assert Tn.dtype == np.float32
    for row, (a, b) in zip(rows[:3], [(Tn, Sn), (Tn, S0n), (T0n, Sn)]):
        ref = o.calc_masso(o.calc_rho(a, b, pres), g["volcello"])
        assert abs(row[t] - ref) <= 1e-12 * abs(ref)
    heat = np.nansum(Tn.astype(np.float64) * g["volcello"])
    assert abs(rows[3][t] - heat) <= 1e-12 * abs(heat)
    # the tolerance study in one line each: upcast and fused arithmetic vs the faithful result
    up = core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, f32_mode="upcast",
                                   arith="exact").cpu().numpy()
    fu = core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, arith="fused").cpu().numpy()
    ex = core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, arith="exact").cpu().numpy()
    upf = core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, f32_mode="upcast",
                                    arith="fused").cpu().numpy()
    assert np.array_equal(fu, rows)                                # the default IS the fused policy
    assert np.max(np.abs(fu[:3] - ex[:3]) / ex[:3]) < 1e-12      # float32 polynomial kept: ~1e-15
    assert np.max(np.abs(up[:3] - ex[:3]) / ex[:3]) < 2e-7       # upcast: float32 polynomial rounding
    assert np.max(np.abs(upf[:3] - up[:3]) / up[:3]) < 1e-12     # upcast fused == upcast to ~1e-15
    assert np.array_equal(fu[3], rows[3]) and np.array_equal(up[3], rows[3])
    # local variant on a 135-row band of one step, float32, vs the oracle (bit-exact)
    y0, y1 = 400, 535
    rho0m = core.fold_mask(core.eos_map(T[0], S[0], pres), vol0)
    for a, b, an, bn in [(T[t:t + 1], S[0], Tn, S0n), (T[0], S[t:t + 1], T0n, Sn)]:
        drho, eta = core.steric_local(a, b, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=g["z_i"],
                                      deptho=g["deptho"])
        vn = g["volcello"][:, y0:y1]
        rho = o.calc_rho(an[:, y0:y1], bn[:, y0:y1], pres)
        rho0n = o.calc_rho(T0n[:, y0:y1], S0n[:, y0:y1], pres)
        d = np.where(~np.isnan(vn), rho - rho0n, np.nan)
        assert_bit_equal(drho[0, :, y0:y1].cpu().numpy(), d, "f32 band delta_rho")
        dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"][y0:y1])
        e = np.where(~np.isnan(vn[0]), (-1.0 / 1035.0) * np.nansum(dz * d, axis=0), np.nan)
        assert_bit_equal(eta[0, y0:y1].cpu().numpy(), e, "f32 band eta")
    del T, S
    _free_hbm()


@pytest.mark.parametrize("want_delta_rho", [True, False])
@pytest.mark.parametrize("dtype,f32_mode,arith", [(np.float64, "faithful", "exact"),
                                                  (np.float32, "faithful", "exact"),
                                                  (np.float32, "upcast", "exact"),
                                                  (np.float64, "faithful", "fused"),
                                                  (np.float32, "faithful", "fused"),
                                                  (T32_S64, "faithful", "exact"),
                                                  (T64_S32, "faithful", "exact"),
                                                  (T64_S32, "upcast", "exact")])
@pytest.mark.parametrize("shape", [(19, 5, 12, 40), (9, 3, 7, 9), (5, 4, 6, 10)])
def test_local_decomposition_fields_equal_single_variant_launches(shape, dtype, f32_mode, arith,
                                                                   want_delta_rho):
    """mlx_steric_local_decomp: the three delta_rho / eta fields from ONE pass over theta/S are
    bit-identical to three mlx_steric_local calls (12x40: dwordx4 / float2 kernels, nt=19 = ragged
    8+8+3; 7x9 and 6x10: the scalar twin), and -- exact arithmetic -- to the oracle."""
    g, T, S = _case_fields(shape, dtype)
    dT, dS = torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda()
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    pres = o.pressure_from_depth(g["z_l"])
    kw = dict(f32_mode=f32_mode, arith=arith, z_i=g["z_i"], deptho=g["deptho"])
    rho0m = core.fold_mask(core.eos_map(dT[0], dS[0], pres, f32_mode=f32_mode, arith=arith), vol0)
    d3, e3 = core.steric_local_decomp(dT, dS, dT[0], dS[0], rho0m, vol0[0], pres, -1.0 / 1035.0,
                                      want_delta_rho=want_delta_rho, **kw)
    assert (d3 is None) == (not want_delta_rho)
    Tn, Sn = (T.astype(np.float64), S.astype(np.float64)) if f32_mode == "upcast" else (T, S)
    pb = pres[:, None, None]
    rho0 = o.wright_density(Tn[0], Sn[0], pb)
    dz = o.calc_dz(g["z_l"], g["z_i"], g["deptho"])
    wet = ~np.isnan(g["volcello"])
    for i, (a, b, an, bn) in enumerate([(dT, dS, Tn, Sn), (dT, dS[0], Tn, Sn[0]),
                                        (dT[0], dS, Tn[0], Sn)]):
        d1, e1 = core.steric_local(a, b, rho0m, vol0[0], pres, -1.0 / 1035.0,
                                   want_delta_rho=want_delta_rho, **kw)
        assert torch.equal(torch.nan_to_num(e3[i], nan=-7.0), torch.nan_to_num(e1, nan=-7.0))
        if want_delta_rho:
            assert torch.equal(torch.nan_to_num(d3[i], nan=-7.0), torch.nan_to_num(d1, nan=-7.0))
        if arith == "exact":
            dref = np.where(wet, np.broadcast_to(o.wright_density(an, bn, pb), T.shape) - rho0, np.nan)
            eref = np.where(wet[0], (-1.0 / 1035.0) * np.nansum(dz * dref, axis=1), np.nan)
            assert_bit_equal(e3[i].cpu().numpy(), eref, f"one-pass eta, variant {i}")
            if want_delta_rho:
                assert_bit_equal(d3[i].cpu().numpy(), dref, f"one-pass delta_rho, variant {i}")
    # strided outputs: the kernel writes straight into a slice of the full-record tensors
    full_e = torch.full((3, shape[0] + 4) + shape[2:], -1.0, dtype=torch.float64, device="cuda")
    core.steric_local_decomp(dT, dS, dT[0], dS[0], rho0m, vol0[0], pres, -1.0 / 1035.0,
                             want_delta_rho=False, eta_out=full_e[:, 2:2 + shape[0]], **kw)
    assert torch.equal(torch.nan_to_num(full_e[:, 2:2 + shape[0]], nan=-7.0),
                       torch.nan_to_num(e3, nan=-7.0))
    assert torch.all(full_e[:, :2] == -1.0) and torch.all(full_e[:, 2 + shape[0]:] == -1.0)


def _cancelling_cells(arith, dtype=np.float64):
    """Operands on which Wright's density is +-inf: (theta, S) pairs and pressures within a few
    hundred ulps of the root of the denominator lam + al0 * (p + p0) (a negative pressure near
    -8.7e8 Pa: no ocean, but a legal operand), evaluated BY THE KERNEL in the given arithmetic.
    -> (T, S, p3d) of shape (1,1,ny,nx) / (1,ny,nx) and the density K0 returns on them."""
    pairs = [(8.308624195915929, 18.13991557922606), (30.69634458456875, 28.991597630941346),
             (18.842112235803377, 36.69190819163611), (26.555971715067898, 20.379835260860375),
             (27.802261278163737, 18.87638877435161), (10.0, 35.0), (2.5, 34.25), (21.0, 31.5)]
    nx = 1024
    T = np.empty((1, 1, len(pairs), nx))
    S = np.empty_like(T)
    p = np.empty((1, len(pairs), nx))
    for j, (t, s) in enumerate(pairs):
        al0, p0, lam = o._wright_terms(np.float64(t), np.float64(s))
        root = -lam / al0 - p0
        T[0, 0, j], S[0, 0, j] = t, s
        p[0, j] = root + (np.arange(nx) - nx // 2) * np.spacing(root)
    rho = core.eos_map(torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda(),
                       torch.from_numpy(p).cuda(), arith=arith).cpu().numpy()
    return T, S, p, rho


@pytest.mark.parametrize("skip_dry", [False, True])
@pytest.mark.parametrize("arith", ["exact", "fused"])
def test_fused_sum_skips_what_the_reference_skips(arith, skip_dry):
    """derived.py:435-438: masso = (rho * volcello).sum(skipna) -- a term is skipped when the PRODUCT
    is NaN.  The fused K1 (the product default) accumulates fma(rho, vol, c) under "neither operand is
    NaN", which is the same test except for inf * 0: rho = +-inf on a zero-volume cell (VERDICT r5
    weak 1e).  K1 now turns zero volumes into NaN volumes as it loads them -- a zero-volume cell
    never changes the reference's sum -- so both arithmetics return what the reference's expression
    returns on the kernel's own densities: rho = +-inf with volume 0 (skipped), NaN (skipped) and 2
    (the sum is +-inf, as numpy's).  Infinite densities exist in EXACT arithmetic only (see below):
    there the cells are planted; under the fused policy the zero / NaN volume equivalence is checked
    on ordinary cells.  Left outside the contract, and said so in DESIGN 3.1: rho = +-0 on a cell of
    INFINITE volume under the fused policy (the exact policy follows there too)."""
    T, S, p, rho = _cancelling_cells(arith)
    hits = np.argwhere(np.isinf(rho[0, 0]))
    if arith == "exact":
        assert len(hits) >= 3, "no exact cancellation found"
    else:
        # the fused denominator fma(al0, p + p0, lam) is rounded ONCE: it vanishes only where the
        # exact value does, and no candidate within 512 ulps of the root of eight (theta, S) pairs
        # does -- an infinite density is not constructible there; the volume classes are checked on
        # ordinary cells (and on whatever non-finite densities the scan did produce)
        assert len(hits) == 0
    ny, nx = rho.shape[2:]
    Td, Sd, pd = (torch.from_numpy(a).cuda() for a in (T, S, p))

    def masso(vol):
        got = core.steric_global_masso(Td, Sd, torch.from_numpy(vol).cuda(), pd, arith=arith,
                                       skip_dry=skip_dry).cpu().numpy()
        with np.errstate(all="ignore"):
            want = o.calc_masso(rho, vol)  # the reference's expression on the kernel's densities
        return got, want

    finite = np.isfinite(rho[0, 0])
    assert finite.sum() > rho[0, 0].size // 2
    base = np.where(finite, 1.0, np.nan)[None]  # (1, ny, nx): every non-finite density masked out
    got, want = masso(base)
    assert np.isfinite(got).all()
    assert_rel(got, want, 1e-12, "finite cells only")
    cells = [tuple(h) for h in hits] + [(j, 3 + 11 * j) for j in range(ny)]
    results = {}
    for fill in (np.nan, 0.0, -0.0):  # +-inf * +-0 = NaN: skipped, like a NaN volume; x * +-0 = +-0
        vol = base.copy()
        for (j, i) in cells:
            vol[0, j, i] = fill
        results[repr(fill)], want2 = masso(vol)
        assert np.isfinite(want2).all()
        assert_rel(results[repr(fill)], want2, 1e-12, f"volume {fill!r} on the chosen cells")
    assert np.array_equal(results["0.0"], results["nan"]) and np.array_equal(results["-0.0"], results["nan"])
    if len(hits):  # (only the infinite-density cells were changed: the sum is the base case's)
        vol = base.copy()
        for (j, i) in hits:
            vol[0, j, i] = 0.0
        got2, want2 = masso(vol)
        assert np.array_equal(want2, want) and np.array_equal(got2, got)
    # zero volumes on ordinary cells change nothing either (+-0 terms), whatever the arithmetic
    vol = base.copy()
    vol[0, :, ::7] = np.where(finite[:, ::7], 0.0, np.nan)
    got3, want3 = masso(vol)
    assert_rel(got3, want3, 1e-12, "zero volumes on ordinary cells")
    # an infinite density on a cell WITH volume is added, as numpy adds it: the sum is +-inf
    if len(hits):
        j, i = hits[0]
        vol = base.copy()
        vol[0, j, i] = 2.0
        got4, want4 = masso(vol)
        assert np.isinf(want4).all() and np.array_equal(got4, want4)
    # the one class left out under the fused policy: rho == +-0 on a cell of infinite volume
    # (p = -p0: the numerator vanishes) -- the exact policy follows the reference there as well
    if arith == "exact":
        al0, p0, lam = o._wright_terms(T[0, 0, 0, 0], S[0, 0, 0, 0])
        pz = p.copy()
        pz[0, 0, 5] = -p0
        rz = core.eos_map(Td, Sd, torch.from_numpy(pz).cuda(), arith="exact").cpu().numpy()
        assert rz[0, 0, 0, 5] == 0.0
        vol = np.where(np.isfinite(rz[0, 0]), 1.0, np.nan)[None]
        vol[0, 0, 5] = np.inf
        got5 = core.steric_global_masso(Td, Sd, torch.from_numpy(vol).cuda(),
                                        torch.from_numpy(pz).cuda(), arith="exact",
                                        skip_dry=skip_dry).cpu().numpy()
        with np.errstate(all="ignore"):
            want5 = o.calc_masso(rz, vol)
        assert np.isfinite(want5).all()
        assert_rel(got5, want5, 1e-12, "rho = 0 on an infinite volume, exact policy")
