"""GPU: the EOS kernel (K0) behind momlevel_amd.eos.wright -- mirrors the reference's
tests/test_wright.py, then tightens it to bit-exactness against the reference module's own
outputs (tests/golden/wright_vectors.npz) and the oracle."""

import numpy as np
import pytest
import torch

from momlevel_amd.eos.wright import alpha, beta, density, drho_dsal, drho_dtemp
from momlevel_amd.eos import linear
from oracle import momlevel_numpy as o
from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu

rng = np.random.default_rng(123)
thetao = rng.normal(15.0, 5.0, (5, 5))
so = rng.normal(35.0, 1.5, (5, 5))
pressure = rng.normal(2000.0, 500.0, (5, 5))

FUNCS = {"density": density, "drho_dtemp": drho_dtemp, "drho_dsal": drho_dsal,
         "alpha": alpha, "beta": beta}


# ---- the reference's own tests (tests/test_wright.py:11-137), np.allclose defaults ---------
def test_wright_density_scalar():
    assert np.allclose(density(18.0, 35.0, 200000.0), 1025.359957453976)


def test_wright_density_3D(goldens):
    assert np.allclose(density(thetao, so, pressure), np.array(goldens["wright_5x5"]["density"]))


def test_wright_drho_dtemp(goldens):
    assert np.allclose(drho_dtemp(18.0, 35.0, 200000.0), -0.24680005918175105)
    assert np.allclose(drho_dtemp(thetao, so, pressure), np.array(goldens["wright_5x5"]["drho_dtemp"]))


def test_wright_drho_dsal(goldens):
    assert np.allclose(drho_dsal(18.0, 35.0, 200000.0), 0.7652676800174607)
    assert np.allclose(drho_dsal(thetao, so, pressure), np.array(goldens["wright_5x5"]["drho_dsal"]))


def test_wright_alpha_beta(goldens):
    assert np.allclose(alpha(18.0, 35.0, 200000.0), 0.0002406960183958898)
    assert np.allclose(beta(18.0, 35.0, 200000.0), 0.0007463405162784603)
    assert np.allclose(beta(thetao, so, pressure), np.array(goldens["wright_5x5"]["beta"]))


# ---- bit-exact against the reference module's outputs ------------------------------------
def test_scalars_bit_exact(wright_vectors):
    T, S, p = wright_vectors["scalar_args"]
    got = np.array([f(T, S, p) for f in (density, drho_dtemp, drho_dsal, alpha, beta)])
    assert_bit_equal(got, wright_vectors["scalar_out"], "scalars")
    assert isinstance(density(T, S, p), np.float64)


@pytest.mark.parametrize("tag", ["tw", "rnd", "blk"])
@pytest.mark.parametrize("func", list(FUNCS))
def test_vectors_bit_exact(wright_vectors, tag, func):
    v = wright_vectors
    got = FUNCS[func](v[f"{tag}_T"], v[f"{tag}_S"], v[f"{tag}_p"])
    # 'rnd' holds NaN/inf/zero edge cases; 'blk' is the calc_rho shape (nt,nz,ny,nx) x (nz,1,1)
    assert_bit_equal(got, v[f"{tag}_{func}"], f"{tag}/{func}")


def test_float32_inputs_follow_numpy_mixed_precision(wright_vectors):
    v = wright_vectors
    got = density(v["f32_T"], v["f32_S"], v["blk_p"])
    assert got.dtype == np.float64
    assert_bit_equal(got, v["f32_density"], "float32 theta/S, float64 p")


@pytest.mark.parametrize("func", ["drho_dtemp", "drho_dsal", "alpha", "beta"])
def test_float32_derivatives_follow_numpy_mixed_precision(wright_vectors, func):
    v = wright_vectors
    got = FUNCS[func](v["f32_T"], v["f32_S"], v["blk_p"])
    assert_bit_equal(got, v[f"f32_{func}"], f"float32 {func}")


def test_float32_upcast_mode_is_plain_float64(wright_vectors):
    from momlevel_amd import core

    v = wright_vectors
    T = torch.from_numpy(v["f32_T"]).cuda()
    S = torch.from_numpy(v["f32_S"]).cuda()
    got = core.eos_map(T, S, v["blk_p"].reshape(-1), f32_mode="upcast").cpu().numpy()
    ref = o.wright_density(v["f32_T"].astype(np.float64), v["f32_S"].astype(np.float64), v["blk_p"])
    assert_bit_equal(got, ref, "upcast")


@pytest.mark.parametrize("shape", [(1,), (7,), (63,), (64,), (65,), (1023,), (3, 5, 7), (2, 3, 5, 8)])
def test_ragged_shapes_and_broadcasting(shape):
    r = np.random.default_rng(sum(shape))
    T = r.uniform(-2, 32, shape)
    S = r.uniform(30, 40, shape)
    p = r.uniform(1e5, 6e7, shape)
    assert_bit_equal(density(T, S, p), o.wright_density(T, S, p))
    assert_bit_equal(density(T, 35.0, 2.0e5), o.wright_density(T, 35.0, 2.0e5))
    assert_bit_equal(density(T, S[..., :1], p), o.wright_density(T, S[..., :1], p))


def test_device_tensors_stay_on_device():
    T = torch.from_numpy(thetao).cuda()
    out = density(T, torch.from_numpy(so).cuda(), torch.from_numpy(pressure).cuda())
    assert isinstance(out, torch.Tensor) and out.is_cuda
    assert_bit_equal(out.cpu().numpy(), o.wright_density(thetao, so, pressure))


def test_linear_eos():
    got = linear.density(thetao, so)
    assert_bit_equal(got, o.linear_density(thetao, so))


def test_large_host_arrays_are_chunked(monkeypatch):
    """calc_rho-style calls on host arrays above the pipeline limit walk the leading axis in
    pieces: upload of piece k+1, kernel of piece k and result download of piece k-1 overlap
    (hostio.Uploader / Downloader).  Same bits as numpy whatever the piece size; operands that
    broadcast along the leading axis, python-float pressures and float32-throughout results
    (numpy's promotion) included."""
    from momlevel_amd import hostio
    from momlevel_amd.eos import _dispatch

    monkeypatch.setattr(_dispatch, "_HOST_PIPELINE_ELEMS", 1000)
    monkeypatch.setattr(_dispatch, "_HOST_CHUNK_ELEMS", 1000)
    r = np.random.default_rng(5)
    T = r.uniform(-2, 32, (7, 5, 6, 10))
    S = r.uniform(30, 40, (7, 5, 6, 10))
    pz = np.linspace(1e5, 5e7, 5)[:, None, None]
    assert_bit_equal(density(T, S, pz), o.wright_density(T, S, pz))
    p4 = r.uniform(1e5, 5e7, (7, 5, 6, 10))
    assert_bit_equal(alpha(T, S, p4), o.wright_alpha(T, S, p4))
    assert_bit_equal(density(T, S[0], 2.0e5), o.wright_density(T, S[0], 2.0e5))
    # pieces large enough for the staging ring (0.96 MB each), float64 and float32 fields
    used = []
    real = hostio._enqueue_download
    monkeypatch.setattr(hostio, "_enqueue_download",
                        lambda out, dev, *a: used.append(out.nbytes) or real(out, dev, *a))
    monkeypatch.setattr(_dispatch, "_HOST_CHUNK_ELEMS", 200_000)
    T = r.uniform(-2, 32, (6, 40, 50, 60))
    S = r.uniform(30, 40, (6, 40, 50, 60))
    pz = np.linspace(1e5, 5e7, 40)[:, None, None]
    assert_bit_equal(density(T, S, pz), o.wright_density(T, S, pz))
    assert len(used) == 6 and all(n == 40 * 50 * 60 * 8 for n in used)
    T32, S32 = T.astype(np.float32), S.astype(np.float32)
    got = density(T32, S32, pz)  # float32 fields, float64 pressure: float64 out
    assert got.dtype == np.float64
    assert_bit_equal(got, o.wright_density(T32, S32, pz))
    got = density(T32, S32, 2.0e5)  # python-float pressure: float32 throughout, as numpy
    assert got.dtype == np.float32
    assert_bit_equal(got, o.wright_density(T32, S32, 2.0e5))


# ---- held-field (thermosteric / halosteric) kernels vs the reference module's vectors ---------
@pytest.mark.parametrize("f32_mode", ["faithful", "upcast"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("held", ["S", "T"])
def test_held_field_kernels_match_reference_vectors(wright_vectors, held, prec, f32_mode):
    """K1 / K2 with HOLD=2 (S held: thermosteric) and HOLD=1 (theta held: halosteric) hoist the
    held field's share of al0/p0/lam out of the time loop (eos_device.hpp, held-field hoisting).
    delta_rho + rho0 must reproduce eos/wright.py:44-48 on the broadcast operands bit for bit --
    float64, and float32 in numpy's mixed precision (faithful); upcast is float64 arithmetic on
    the float32 values."""
    from momlevel_amd import core

    if prec == "f64" and f32_mode == "upcast":
        pytest.skip("f32_mode only matters for float32 inputs")
    v = wright_vectors
    T, S = (v["f32_T"], v["f32_S"]) if prec == "f32" else (v["blk_T"], v["blk_S"])
    p = v["blk_p"]
    if prec == "f32" and f32_mode == "upcast":
        Tn, Sn = T.astype(np.float64), S.astype(np.float64)
        ref = o.wright_density(Tn, Sn[0], p) if held == "S" else o.wright_density(Tn[0], Sn, p)
        rho0 = o.wright_density(Tn[0], Sn[0], p)
    else:
        ref = v[f"{prec}_density_held{held}"]
        rho0 = o.wright_density(T[0], S[0], p)
    nt, nz, ny, nx = T.shape
    dT, dS = torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda()
    Tv, Sv = (dT, dS[0]) if held == "S" else (dT[0], dS)
    vol = np.random.default_rng(5).uniform(1e8, 1e12, (nz, ny, nx))
    dvol = torch.from_numpy(vol).cuda()
    pz = p.reshape(-1)
    # K2: delta_rho = rho - rho0, elementwise -> bit-exact
    rho0m = core.fold_mask(torch.from_numpy(rho0).cuda(), dvol)
    z_i = np.concatenate([[0.0], np.cumsum(np.full(nz, 10.0))])
    drho, eta = core.steric_local(Tv, Sv, rho0m, dvol[0], pz, -1.0 / 1035.0, z_i=z_i,
                                  deptho=np.full((ny, nx), 1e4), f32_mode=f32_mode)
    assert_bit_equal(drho.cpu().numpy(), ref - rho0, f"K2 held {held} {prec} {f32_mode}")
    eta_ref = (-1.0 / 1035.0) * np.sum(10.0 * (ref - rho0), axis=1)
    assert_bit_equal(eta.cpu().numpy(), eta_ref, f"K2 eta held {held} {prec} {f32_mode}")
    # K1: masso(t) = sum(rho * vol) -- order of summation differs, 1e-12
    masso = core.steric_global_masso(Tv, Sv, dvol, pz, f32_mode=f32_mode).cpu().numpy()
    mref = np.sum(ref * vol, axis=(1, 2, 3))
    assert np.max(np.abs(masso - mref) / mref) < 1e-12
    # K1 with a ONE-HOT volcello (1.0 in one cell, NaN elsewhere): masso(t) is rho of that cell,
    # so the hoisted arithmetic of the dwordx4 kernels is seen bit for bit through the reduction
    for (z, j) in [(0, 0), (3, 17), (nz - 1, ny * nx - 1)]:
        hot = np.full((nz, ny * nx), np.nan)
        hot[z, j] = 1.0
        for skip in (False, True):
            # (arith="exact": K1's default on float64 input is the fused policy since round 3)
            one = core.steric_global_masso(Tv, Sv, torch.from_numpy(hot.reshape(nz, ny, nx)).cuda(),
                                           pz, f32_mode=f32_mode, skip_dry=skip,
                                           arith="exact").cpu().numpy()
            assert_bit_equal(one, ref.reshape(nt, nz, -1)[:, z, j], f"K1 one-hot held {held}")


def test_bare_1d_pressure_follows_numpy_alignment():
    """eos.wright.density claims numpy broadcasting: a (n,) pressure against (n,n,n) fields aligns
    with the LAST axis (x), as in the reference's eos/wright.py; only (nz,1,1) / (1,nz,1,1) is a
    z profile."""
    r = np.random.default_rng(11)
    T, S = r.uniform(0, 30, (5, 5, 5)), r.uniform(30, 40, (5, 5, 5))
    p = r.uniform(1e5, 5e7, 5)
    assert_bit_equal(density(T, S, p), o.wright_density(T, S, p), "(n,) -> x")
    assert_bit_equal(density(T, S, p[:, None, None]), o.wright_density(T, S, p[:, None, None]),
                     "(n,1,1) -> z")
    T4, S4 = r.uniform(0, 30, (3, 5, 5, 5)), r.uniform(30, 40, (3, 5, 5, 5))
    p4 = p.reshape(1, 5, 1, 1)
    assert_bit_equal(density(T4, S4, p4), o.wright_density(T4, S4, p4), "(1,n,1,1) -> z")
    with pytest.raises(Exception):
        density(T[:, :, :4], S[:, :, :4], p)  # numpy would refuse to broadcast (5,) against x=4


def test_4d_z_profile_takes_the_profile_kernel(monkeypatch):
    """calc_rho on 4-D fields hands (1,nz,1,1): it must reach the kernel as MLX_P_ZPROF (nz
    doubles), not as a materialised 4-D pressure (+8 B/cell)"""
    from momlevel_amd import core, _lib

    T = torch.rand((2, 3, 4, 8), dtype=torch.float64, device="cuda")
    pt, mode = core._pressure(np.arange(3.0).reshape(1, 3, 1, 1), 2, 3, 4, 8, T.device, True)
    assert mode == _lib.P_ZPROF and tuple(pt.shape) == (3,)


# ---- the linear EOS, complete (reference tests/test_linear.py + the reference module's vectors) --
def test_linear_eos_reference_tests(goldens):
    g = goldens["linear"]
    assert np.allclose(linear.density(18.0, 35.0, 200000.0), 1024.4)
    assert np.allclose(linear.density(thetao, so, pressure), np.array(g["density_5x5"]))
    assert np.allclose(linear.drho_dtemp(18.0, 35.0, 200000.0), -0.2)
    assert np.allclose(linear.drho_dtemp(thetao, so, pressure), -0.2)
    assert np.allclose(linear.drho_dsal(18.0, 35.0, 200000.0), 0.8)
    assert np.allclose(linear.alpha(18.0, 35.0, 200000.0), 0.0001952362358453729)
    assert np.allclose(linear.alpha(thetao, so, pressure), np.array(g["alpha_5x5"]))
    assert np.allclose(linear.beta(18.0, 35.0, 200000.0), 0.0007809449433814916)
    assert np.allclose(linear.beta(thetao, so, pressure), np.array(g["beta_5x5"]))


@pytest.mark.parametrize("tag", ["tw", "blk", "f32"])
@pytest.mark.parametrize("func", ["density", "alpha", "beta"])
def test_linear_eos_bit_exact(wright_vectors, tag, func):
    """bit for bit against the outputs of the reference's eos/linear.py, float64 and float32
    (numpy: full_like(T) / density(T, S) stays float32; the kernel returns it widened)"""
    v = wright_vectors
    T, S = {"tw": ("tw_T", "tw_S"), "blk": ("blk_T", "blk_S"), "f32": ("f32_T", "f32_S")}[tag]
    got = getattr(linear, func)(v[T], v[S], None)
    assert_bit_equal(got, v[f"lin_{tag}_{func}"].astype(np.float64), f"linear {tag}/{func}")


@pytest.mark.parametrize("tag", ["tw", "f32"])
def test_linear_density_with_a_reference_density_bit_exact(wright_vectors, tag):
    """eos/linear.py:55-56 with rho_ref given: (1000 - rho_ref) + ((-0.2*T) + (0.8*S)), the constant
    formed first; python-float and numpy-scalar reference densities, float64 and float32 fields;
    value AND result dtype against the reference module's own outputs (rounds
    3-5 returned density(T, S) - rho_ref, other bits)."""
    v = wright_vectors
    T, S = (v["tw_T"], v["tw_S"]) if tag == "tw" else (v["f32_T"], v["f32_S"])
    py, py2, n64, n32 = (float(x) for x in v["linref_values"])
    for rk, rv in {"py": py, "py2": py2, "np64": np.float64(n64), "np32": np.float32(n32)}.items():
        want = v[f"linref_{tag}_{rk}"]
        got = linear.density(T, S, None, rv)
        assert got.dtype == want.dtype, (tag, rk, got.dtype)
        assert_bit_equal(got, want, f"linear density with rho_ref ({rk}), {tag}")
        dev = linear.density(torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda(), None, rv)
        assert dev.is_cuda
        assert_bit_equal(dev.cpu().numpy(), want, f"device operands, rho_ref ({rk}), {tag}")
    assert_bit_equal(linear.density(T, S, 2.0e5, None), v[f"lin_{tag}_density"].astype(np.float64)
                     if tag == "tw" else linear.density(T, S))  # rho_ref=None: unchanged


def test_calc_alpha_beta_with_the_linear_eos():
    from momlevel_amd import derived
    from momlevel_amd.labeled import DataArray

    r = np.random.default_rng(2)
    T = DataArray(r.uniform(0, 30, (3, 4, 5, 6)), ("time", "z_l", "yh", "xh"))
    S = DataArray(r.uniform(30, 40, (3, 4, 5, 6)), ("time", "z_l", "yh", "xh"))
    p = DataArray(np.linspace(1e5, 4e7, 4), ("z_l",))
    a = derived.calc_alpha(T, S, p, eos="linear")
    b = derived.calc_beta(T, S, p, eos="linear")
    assert_bit_equal(a.values, o.linear_alpha(T.values, S.values))
    assert_bit_equal(b.values, o.linear_beta(T.values, S.values))


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_guarded_reciprocal_falls_back_on_pathological_operands(wright_vectors, prec):
    """Round 3: the exact kernels take a scale-free reciprocal under a per-wave guard
    (eos_device.hpp).  Operands no ocean produces -- temperatures of 1e150, pressures of 1e300 or
    1e-300, float32 values that overflow the float32 polynomial -- must send their wave to the IEEE
    division and come out exactly as momlevel computes them (inf, 0, NaN included), next to ordinary
    cells in the same wave.  Expected values: the REFERENCE module's own outputs on these inputs
    (tests/golden/wright_vectors.npz, patho_*); K0, K2 (delta_rho) and the one-hot K1 sum all see
    the same bits."""
    from momlevel_amd import core

    v = wright_vectors
    T, S, ref, pz = v[f"patho_{prec}_T"], v[f"patho_{prec}_S"], v[f"patho_{prec}_density"], v["patho_p"]
    nt, nz, ny, nx = T.shape
    assert np.isnan(ref).any() and T.dtype == (np.float64 if prec == "f64" else np.float32)
    dT, dS = torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda()
    got = core.eos_map(dT, dS, pz).cpu().numpy()
    assert_bit_equal(got, ref, f"K0 on pathological operands, {prec}")
    with np.errstate(all="ignore"):  # the oracle agrees with the reference there too
        assert_bit_equal(o.wright_density(T, S, pz[:, None, None]), ref, "oracle vs reference")
        # K2: delta_rho = rho - rho0 with rho0 = an ordinary slab
        rho0 = o.wright_density(np.full((nz, ny, nx), 10.0), np.full((nz, ny, nx), 35.0),
                                pz[:, None, None])
        dref = ref - rho0
    vol = torch.ones((nz, ny, nx), dtype=torch.float64, device="cuda")
    drho, _ = core.steric_local(dT, dS, core.fold_mask(torch.from_numpy(rho0).cuda(), vol), vol[0],
                                pz, -1.0 / 1035.0, z_i=np.array([0.0, 10.0, 20.0, 30.0]),
                                deptho=np.full((ny, nx), 1e4))
    assert_bit_equal(drho.cpu().numpy(), dref, "K2 delta_rho on pathological operands")
    # K1 (exact) through a one-hot volcello at three of the planted cells
    for (z, y, x) in [(0, 0, 3), (1, 1, 8), (2, 6, 16)]:
        hot = np.full((nz, ny, nx), np.nan)
        hot[z, y, x] = 1.0
        one = core.steric_global_masso(dT, dS, torch.from_numpy(hot).cuda(), pz,
                                       arith="exact").cpu().numpy()
        want = np.where(np.isnan(ref[:, z, y, x]), 0.0, ref[:, z, y, x])  # skipna
        assert np.array_equal(one, want), (z, y, x, one, want)
