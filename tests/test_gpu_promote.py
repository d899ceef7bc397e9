"""GPU: the EOS functions under numpy's type promotion -- every dtype combination the reference's
numpy expressions accept (mlx_eos_map_promote, csrc/eos_promote.hpp).

The expected values are what the REFERENCE's own eos/wright.py and eos/linear.py returned for each
of T, S, p being a float64 array, a float32 array or a python float (tests/golden/make_golden.py
section 8; 26 combinations x 8 functions, NaN / inf / zero operands included): bit for bit, and the
result dtype numpy gave.  Then the public front ends (eos.wright / eos.linear functions, calc_rho,
calc_pdens, inverse_barometer) on the combinations real data produces -- float32 MOM6 fields with a
python-float pressure stay float32 throughout -- against the numpy oracle."""

import numpy as np
import pytest
import torch

from momlevel_amd import core, derived
from momlevel_amd.eos import linear, wright
from momlevel_amd.labeled import DataArray
from oracle import momlevel_numpy as o
from conftest import MIX_FUNCS, MIX_KINDS, assert_bit_equal, mixed_operands

pytestmark = pytest.mark.gpu

ORACLE = {"density": o.wright_density, "drho_dtemp": o.wright_drho_dtemp,
          "drho_dsal": o.wright_drho_dsal, "alpha": o.wright_alpha, "beta": o.wright_beta}


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda() if isinstance(x, np.ndarray) else x


@pytest.mark.parametrize("kinds", MIX_KINDS)
def test_c_abi_against_the_reference_vectors(wright_vectors, kinds):
    ops = [_dev(x) for x in mixed_operands(wright_vectors, kinds)]
    for name in MIX_FUNCS:
        want = wright_vectors[f"mix_{kinds}_{name}"]
        eos, func = ("linear", name[4:]) if name.startswith("lin_") else ("wright", name)
        out = core.eos_map_promote(*ops, eos=eos, func=func)
        got = out.cpu().numpy()
        assert got.dtype == want.dtype, (kinds, name)  # the kernel stores numpy's result dtype
        assert_bit_equal(got, np.broadcast_to(want, got.shape), f"{kinds} {name}")


@pytest.mark.parametrize("kinds", MIX_KINDS)
def test_public_functions_take_what_numpy_takes(wright_vectors, kinds):
    """momlevel_amd.eos.wright / eos.linear called the way the reference's modules are: numpy
    arrays and python floats in, numpy's values and numpy's dtype out."""
    ops = mixed_operands(wright_vectors, kinds)
    fns = {"density": wright.density, "drho_dtemp": wright.drho_dtemp,
           "drho_dsal": wright.drho_dsal, "alpha": wright.alpha, "beta": wright.beta,
           "lin_density": linear.density, "lin_alpha": linear.alpha, "lin_beta": linear.beta}
    for name in MIX_FUNCS:
        want = wright_vectors[f"mix_{kinds}_{name}"]
        got = np.asarray(fns[name](*ops))  # (python floats only: a scalar, as numpy)
        assert got.dtype == want.dtype and got.shape == want.shape, (kinds, name, got.dtype)
        assert_bit_equal(got, want, f"{kinds} {name}")


@pytest.mark.parametrize("kinds,tuned_dtype", [("ddd", "f64"), ("ffd", "f32")])
def test_promote_kernel_agrees_with_the_tuned_kernel(wright_vectors, kinds, tuned_dtype):
    """The two combinations the steric path streams have their own tuned kernels (mlx_eos_map):
    same bits from both implementations."""
    T, S, p = mixed_operands(wright_vectors, kinds)
    n = T.size
    for func in ORACLE:
        tuned = core.eos_map(_dev(T.reshape(1, 1, n)), _dev(S.reshape(1, 1, n)),
                             _dev(p.reshape(1, 1, n)), func=func).reshape(n).cpu().numpy()
        out = core.eos_map_promote(_dev(T), _dev(S), _dev(p), func=func)
        assert out.dtype == torch.float64
        assert_bit_equal(out.cpu().numpy(), tuned, f"{kinds} {func}")


@pytest.mark.parametrize("n", [1, 3, 4, 5, 255, 1021, 1024, 4099])
def test_ragged_sizes_and_unaligned_views(n):
    """The kernel reads and writes four cells per thread with 16-byte accesses when every operand
    is 16-byte aligned; the last partial group, and views that start in the middle of a 16-byte
    line, go cell by cell.  Same bits either way."""
    rng = np.random.default_rng(n)
    T = rng.uniform(-2, 30, n + 3).astype(np.float32)
    S = rng.uniform(30, 40, n + 3)
    p = rng.uniform(1e5, 5e7, n + 3).astype(np.float32)
    dT, dS, dp = (torch.from_numpy(x).cuda() for x in (T, S, p))
    for off in (0, 1, 3):  # element offsets 1 and 3: 4 / 12 bytes (float32), 8 / 24 (float64)
        for ops, want in (((dT[off:off + n], dS[off:off + n], dp[off:off + n]),
                           o.wright_density(T[off:off + n], S[off:off + n], p[off:off + n])),
                          ((dT[off:off + n], dT[off:off + n] + 30, 2.0e7),
                           o.wright_density(T[off:off + n], T[off:off + n] + np.float32(30), 2.0e7)),
                          ((dT[off:off + n], dS[off:off + 1], dp[off:off + n]),
                           o.wright_density(T[off:off + n], S[off:off + 1], p[off:off + n]))):
            got = core.eos_map_promote(*ops).cpu().numpy()
            assert got.dtype == want.dtype and got.shape == (n,)
            assert_bit_equal(got, want, f"n={n} offset={off}")


def test_numpy_scalars_are_not_weak():
    """np.float64(p) is a float64 OPERAND (NEP 50): float32 fields then promote where it enters,
    exactly as with a float64 array -- unlike the python float, which leaves everything float32."""
    rng = np.random.default_rng(3)
    T = rng.uniform(-2, 30, (4, 6)).astype(np.float32)
    S = rng.uniform(30, 40, (4, 6)).astype(np.float32)
    for p in (2.0e7, np.float64(2.0e7), np.float32(2.0e7), np.array(2.0e7), 20000000):
        want = o.wright_density(T, S, p)
        got = wright.density(T, S, p)
        assert got.dtype == want.dtype, type(p)
        assert_bit_equal(got, want, repr(type(p)))
    assert wright.density(T, S, 2.0e7).dtype == np.float32
    assert wright.density(T, S, np.float64(2.0e7)).dtype == np.float64
    # 0-d numpy scalars for everything: numpy returns a numpy scalar of the promoted dtype
    got = wright.density(np.float32(10.0), np.float32(35.0), 2.0e7)
    want = o.wright_density(np.float32(10.0), np.float32(35.0), 2.0e7)
    assert type(got) is type(want) is np.float32 and got == want


def test_integer_fields_compute_as_float64_and_float16_is_refused():
    """numpy promotes ``int_array * python_float`` to float64 (the python float is weak only among
    floats), so an integer or boolean field behaves like its float64 conversion in every
    sub-expression -- here against numpy itself, with a float32 partner field and every pressure
    kind.  float16 fields, whose part of the polynomial numpy would evaluate IN float16, raise."""
    rng = np.random.default_rng(5)
    S = rng.uniform(30, 40, (4, 6)).astype(np.float32)
    for dt in (np.int16, np.uint8, np.int64, np.bool_):
        T = rng.integers(0, 2 if dt is np.bool_ else 30, (4, 6)).astype(dt)
        for p in (2.0e7, np.float32(2.0e7), np.full((4, 6), 2.0e7)):
            for name, fn in ORACLE.items():
                want = fn(T, S, p)
                got = getattr(wright, name)(T, S, p)
                assert got.dtype == want.dtype == np.float64, (dt, name)
                assert_bit_equal(got, want, f"{name}, T {np.dtype(dt).name}")
    with pytest.raises(TypeError, match="float16"):
        wright.density(S.astype(np.float16), S, 2.0e7)


def test_broadcasting_with_mixed_dtypes():
    """(time, z, y, x) float32 theta against a float64 (z, y, x) salinity slab and a float32
    (z, 1, 1) pressure profile: numpy's broadcasting and promotion together."""
    rng = np.random.default_rng(11)
    T = rng.uniform(-2, 30, (3, 5, 4, 7)).astype(np.float32)
    S = rng.uniform(30, 40, (5, 4, 7))
    p = (np.linspace(0, 5.0e7, 5).astype(np.float32) + np.float32(101325.0))[:, None, None]
    T[1, 2, 3, 4] = np.nan
    for name, fn in ORACLE.items():
        want = fn(T, S, p)
        got = getattr(wright, name)(T, S, p)
        assert got.shape == want.shape and got.dtype == want.dtype
        assert_bit_equal(got, want, name)
    # float32 everything, float32 profile: all float32
    S32 = S.astype(np.float32)
    want = o.wright_density(T, S32, p)
    got = wright.density(T, S32, p)
    assert got.dtype == want.dtype == np.float32
    assert_bit_equal(got, want, "all float32")


def test_device_tensors_stay_on_the_device():
    rng = np.random.default_rng(12)
    T = rng.uniform(-2, 30, (6, 9)).astype(np.float32)
    S = rng.uniform(30, 40, (6, 9))
    out = wright.density(torch.from_numpy(T).cuda(), torch.from_numpy(S).cuda(), 3.0e7)
    assert isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float64
    assert_bit_equal(out.cpu().numpy(), o.wright_density(T, S, 3.0e7), "device mixed")
    out32 = wright.alpha(torch.from_numpy(T).cuda(), torch.from_numpy(S.astype(np.float32)).cuda(), 3.0e7)
    assert out32.dtype == torch.float32
    assert_bit_equal(out32.cpu().numpy(), o.wright_alpha(T, S.astype(np.float32), 3.0e7), "device f32")


def test_calc_pdens_on_float32_fields_is_float32():
    """derived.py:477: the pressure is the python float (level*1e4 + patm), so on MOM6's float32
    output numpy evaluates potential density in float32 from end to end -- and so does this."""
    rng = np.random.default_rng(13)
    dims = ("time", "z_l", "yh", "xh")
    T = rng.uniform(-2, 30, (2, 4, 5, 6)).astype(np.float32)
    S = rng.uniform(30, 40, (2, 4, 5, 6)).astype(np.float32)
    for level in (0.0, 2000.0):
        got = derived.calc_pdens(DataArray(T, dims), DataArray(S, dims), level=level)
        want = o.wright_density(T, S, (level * 1.0e4) + 101325)
        assert want.dtype == np.float32
        assert got.values.dtype == np.float32 and got.dims == dims
        assert_bit_equal(got.values, want, f"calc_pdens level {level}")
    # calc_rho with a float64 z-profile: the steric path's combination, float64 out (tuned kernel)
    p = DataArray(np.linspace(101325.0, 4.0e7, 4), ("z_l",))
    rho = derived.calc_rho(DataArray(T, dims), DataArray(S, dims), p)
    assert rho.values.dtype == np.float64
    assert_bit_equal(rho.values, o.wright_density(T, S, p.values[:, None, None]), "calc_rho f32/f64")
    # ... and with theta float32, salinity float64
    rho = derived.calc_rho(DataArray(T, dims), DataArray(S.astype(np.float64), dims), p)
    assert_bit_equal(rho.values, o.wright_density(T, S.astype(np.float64), p.values[:, None, None]),
                     "calc_rho mixed")
    for fn, ofn in ((derived.calc_alpha, o.wright_alpha), (derived.calc_beta, o.wright_beta)):
        got = fn(DataArray(T, dims), DataArray(S, dims), 101325.0)
        want = ofn(T, S, 101325.0)
        assert got.values.dtype == want.dtype == np.float32
        assert_bit_equal(got.values, want, fn.__name__)


def test_inverse_barometer_promotes_like_xarray():
    """dynamic.py:34-36 on float32 surface fields: pso a float32 array (all float32), a float64
    array (promotes where it enters), a python float."""
    from momlevel_amd import inverse_barometer

    rng = np.random.default_rng(14)
    dims = ("time", "yh", "xh")
    T = rng.uniform(-2, 30, (3, 5, 6)).astype(np.float32)
    S = rng.uniform(30, 40, (3, 5, 6)).astype(np.float32)
    pso64 = 101325.0 + rng.normal(0, 800.0, (5, 6))
    for eos, dens in (("Wright", o.wright_density), ("linear", lambda T, S, p: o.linear_density(T, S))):
        for pso in (pso64, pso64.astype(np.float32), 101325.0):
            arg = DataArray(pso, ("yh", "xh")) if isinstance(pso, np.ndarray) else pso
            got = inverse_barometer(DataArray(T, dims), DataArray(S, dims), arg, gravity=9.81,
                                    equation_of_state=eos)
            pb = pso[None] if isinstance(pso, np.ndarray) else pso
            want = pb * (-1.0 / (dens(T, S, pb) * 9.81))
            assert got.values.dtype == want.dtype, (eos, type(pso), getattr(pso, "dtype", None))
            assert got.dims == dims and got.name == "ibh"
            assert_bit_equal(got.values, want, f"ibh {eos}")


def test_argument_errors():
    t = torch.zeros(8, dtype=torch.float32, device="cuda")
    with pytest.raises(TypeError):
        core.eos_map_promote(1.0, 2.0, 3.0)                      # no array at all
    with pytest.raises(ValueError):
        core.eos_map_promote(t, t[:3], 1.0)                      # neither n nor 1 elements
    with pytest.raises(TypeError):
        core.eos_map_promote(t.to(torch.float16), t, 1.0)
    with pytest.raises(TypeError):
        core.eos_map_promote(t, t, None)                         # Wright needs a pressure
    out = core.eos_map_promote(t, t, None, eos="linear")  # the linear EOS does not
    assert out.dtype == torch.float32 and out.shape == (8,) and torch.all(out == 1000.0)
