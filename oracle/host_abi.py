"""ctypes binding of oracle/libmomlevel_host.so, the HOST build of the C ABI.  TEST INFRASTRUCTURE ONLY.

Same symbols and argument lists as libmomlevel_hip.so (the table is the product's own
``momlevel_amd._lib.SIGNATURES``), host pointers instead of device pointers.  The numpy wrappers
below exist for the tests; nothing under momlevel_amd/ imports this module.
"""

import ctypes
import os
import subprocess

import numpy as np

from momlevel_amd import _lib as abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmomlevel_host.so")
_DT = {np.dtype("float64"): abi.DTYPE_F64, np.dtype("float32"): abi.DTYPE_F32}
F32_MODES = {"faithful": abi.DTYPE_F32, "upcast": abi.DTYPE_F32_UPCAST}


def build(force=False):
    deps = [os.path.join(HERE, "host_abi.c"), os.path.join(HERE, "host_promote.cpp"),
            os.path.join(HERE, "..", "include", "momlevel_hip.h"),
            os.path.join(HERE, "..", "momlevel_amd", "csrc", "eos_promote.hpp")]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(map(os.path.getmtime, deps)):
        subprocess.run(["make", "-C", HERE, "-B", "libmomlevel_host.so"], check=True,
                       capture_output=True)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(LIB)
        for name, (restype, argtypes) in abi.SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError = the host build misses a symbol of the ABI
            fn.restype, fn.argtypes = restype, argtypes
        if lib.mlx_version() != abi.ABI_VERSION:
            raise RuntimeError("host ABI build is out of date: make -C oracle -B")
        _lib = lib
    return _lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().mlx_last_error(buf, 512)
    return buf.value.decode()


def _p(a):
    return None if a is None else a.ctypes.data


def _c(a, dtype=None):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _pair(T, S, f32_mode):
    T, S = _c(T), _c(S)
    assert T.dtype in _DT and S.dtype in _DT
    nz, ny, nx = T.shape[-3:]
    nt = max(T.shape[0] if T.ndim == 4 else 1, S.shape[0] if S.ndim == 4 else 1)
    n3 = nz * ny * nx
    sT = n3 if T.ndim == 4 else 0
    sS = n3 if S.ndim == 4 else 0
    if T.dtype != S.dtype:  # theta / salinity of different dtypes (K1 / K2 only)
        dt = abi.DTYPE_T32_S64 if T.dtype == np.float32 else abi.DTYPE_T64_S32
    else:
        dt = abi.DTYPE_F64 if T.dtype == np.float64 else F32_MODES[f32_mode]
    return T, S, nt, nz, ny, nx, sT, sS, dt


def _pressure(p, nt, nz, ny, nx):
    p = np.asarray(p, dtype=np.float64)
    if p.size == 1:
        return _c(p.reshape(1)), abi.P_SCALAR
    if p.shape in ((nz,), (nz, 1, 1)):
        return _c(p.reshape(nz)), abi.P_ZPROF
    if p.ndim <= 3:
        return _c(np.broadcast_to(p, (nz, ny, nx))), abi.P_FULL3D
    return _c(np.broadcast_to(p, (nt, nz, ny, nx))), abi.P_FULL4D


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} -> {rc}: {last_error()}")


def eos_map(T, S, p, eos="wright", func="density", f32_mode="faithful"):
    T, S, nt, nz, ny, nx, sT, sS, dt = _pair(T, S, f32_mode)
    pp, pm = _pressure(p, nt, nz, ny, nx)
    out = np.empty((nt, nz, ny, nx))
    _check(load().mlx_eos_map(_p(T), _p(S), dt, _p(pp), pm, abi.EOS_IDS[eos], abi.FUNC_IDS[func],
                              nt, nz, ny * nx, sT, sS, 0, _p(out), None), "mlx_eos_map")
    return out


def eos_map_promote(T, S, p, eos="wright", func="density", gravity=9.8):
    """mlx_eos_map_promote on host operands: python floats are weak scalars, float32 / float64
    arrays of one common size (or one element) keep their dtype.  Returns numpy's values in
    numpy's result dtype."""
    keep, args, n = [], [], 1
    for x in (T, S, p):
        if x is None:
            args += [None, abi.KIND_WEAK, 0]
        elif isinstance(x, (bool, int, float)) and not isinstance(x, np.generic):
            w = ctypes.c_double(float(x))
            keep.append(w)
            args += [ctypes.addressof(w), abi.KIND_WEAK, 0]
        else:
            a = np.ascontiguousarray(x)
            assert a.dtype in _DT
            keep.append(a)
            n = max(n, a.size)
            args += [a.ctypes.data, abi.KIND_F32 if a.dtype == np.float32 else abi.KIND_F64, a]
    args = [(1 if (v.size == n and n > 1) else 0) if isinstance(v, np.ndarray) else v for v in args]
    out = np.empty(n)
    kind = ctypes.c_int(-1)
    fid = {"inverse_barometer": abi.FUNC_IBH, "density_ref": abi.FUNC_DENSITY_REF}.get(func)
    if fid is None:
        fid = abi.FUNC_IDS[func]
    _check(load().mlx_eos_map_promote(*args, abi.EOS_IDS[eos], fid, float(gravity), n, _p(out),
                                      ctypes.byref(kind), None), "mlx_eos_map_promote")
    return out.view(np.float32)[:n].copy() if kind.value == abi.KIND_F32 else out


def steric_global(T, S, vol0, p, eos="wright", f32_mode="faithful", flags=0):
    T, S, nt, nz, ny, nx, sT, sS, dt = _pair(T, S, f32_mode)
    pp, pm = _pressure(p, nt, nz, ny, nx)
    vol0 = _c(vol0, np.float64)
    out = np.empty(nt)
    _check(load().mlx_steric_global(_p(T), _p(S), dt, _p(vol0), _p(pp), pm, abi.EOS_IDS[eos], nt, nz,
                                    ny * nx, sT, sS, flags, _p(out), None, 0, None),
           "mlx_steric_global")
    return out


def steric_global_decomp(T, S, T0, S0, vol0, p, eos="wright", f32_mode="faithful"):
    T, S, nt, nz, ny, nx, sT, sS, dt = _pair(T, S, f32_mode)
    pp, pm = _pressure(p, nt, nz, ny, nx)
    vol0, T0, S0 = _c(vol0, np.float64), _c(T0, T.dtype), _c(S0, S.dtype)
    out = np.empty((4, nt))
    _check(load().mlx_steric_global_decomp(_p(T), _p(S), _p(T0), _p(S0), dt, _p(vol0), _p(pp), pm,
                                           abi.EOS_IDS[eos], nt, nz, ny * nx, sT, sS, 0, _p(out),
                                           None, 0, None), "mlx_steric_global_decomp")
    return out


def fold_mask(rho0, vol0):
    rho0, vol0 = _c(rho0, np.float64), _c(vol0, np.float64)
    out = np.empty_like(rho0)
    _check(load().mlx_fold_mask(_p(rho0), _p(vol0), rho0.size, _p(out), None), "mlx_fold_mask")
    return out


def steric_local(T, S, rho0m, vol0_surface, p, neg_inv_rhozero, z_i=None, deptho=None, dz=None,
                 eos="wright", f32_mode="faithful", want_delta_rho=True):
    T, S, nt, nz, ny, nx, sT, sS, dt = _pair(T, S, f32_mode)
    pp, pm = _pressure(p, nt, nz, ny, nx)
    rho0m, surf = _c(rho0m, np.float64), _c(vol0_surface, np.float64)
    dz, z_i, deptho = _c(dz, np.float64), _c(z_i, np.float64), _c(deptho, np.float64)
    drho = np.empty((nt, nz, ny, nx)) if want_delta_rho else None
    eta = np.empty((nt, ny, nx))
    _check(load().mlx_steric_local(_p(T), _p(S), dt, _p(rho0m), _p(surf), _p(dz), _p(z_i), _p(deptho),
                                   _p(pp), pm, abi.EOS_IDS[eos], float(neg_inv_rhozero), nt, nz,
                                   ny * nx, sT, sS, 0, _p(drho), _p(eta), None), "mlx_steric_local")
    return drho, eta


def steric_local_decomp(T, S, T0, S0, rho0m, vol0_surface, p, neg_inv_rhozero, z_i=None,
                        deptho=None, dz=None, eos="wright", f32_mode="faithful"):
    T, S, nt, nz, ny, nx, sT, sS, dt = _pair(T, S, f32_mode)
    pp, pm = _pressure(p, nt, nz, ny, nx)
    rho0m, surf = _c(rho0m, np.float64), _c(vol0_surface, np.float64)
    T0, S0 = _c(T0, T.dtype), _c(S0, S.dtype)
    dz, z_i, deptho = _c(dz, np.float64), _c(z_i, np.float64), _c(deptho, np.float64)
    drho = np.empty((3, nt, nz, ny, nx))
    eta = np.empty((3, nt, ny, nx))
    _check(load().mlx_steric_local_decomp(
        _p(T), _p(S), _p(T0), _p(S0), dt, _p(rho0m), _p(surf), _p(dz), _p(z_i), _p(deptho), _p(pp),
        pm, abi.EOS_IDS[eos], float(neg_inv_rhozero), nt, nz, ny * nx, sT, sS, 0, _p(drho),
        drho[0].size, _p(eta), eta[0].size, None), "mlx_steric_local_decomp")
    return drho, eta


def nansum(x):
    x = _c(x, np.float64)
    out = np.empty(1)
    _check(load().mlx_nansum(_p(x), x.size, _p(out), None, 0, None), "mlx_nansum")
    return out[0]


def calc_dz(z_i, depth, top=0.0, bottom=None, fraction=False):
    z_i, depth = _c(z_i, np.float64), _c(depth, np.float64)
    nz = z_i.size - 1
    out = np.empty((nz,) + depth.shape)
    _check(load().mlx_calc_dz(_p(z_i), _p(depth), nz, depth.size, float(top),
                              0.0 if bottom is None else float(bottom), int(bottom is not None),
                              int(bool(fraction)), _p(out), None), "mlx_calc_dz")
    return out


def group_weighted_mean(x, w, group_len):
    x, w = _c(x, np.float64), _c(w, np.float64)
    ngroups = x.shape[0] // group_len
    out = np.empty((ngroups,) + x.shape[1:])
    _check(load().mlx_group_weighted_mean(_p(x), _p(w), ngroups, group_len, x[0].size, _p(out), None),
           "mlx_group_weighted_mean")
    return out


def stratification(T, S, p, coef, uniform, two_dx, func="n2", eos="wright", gravity=-9.8,
                   f32_mode="faithful"):
    """mlx_stratification of the host build on (nt, nz, plane) arrays; ``p``: None, (nz,),
    (nz, plane) or (nt, nz, plane); ``coef`` / ``uniform`` / ``two_dx`` as
    momlevel_amd.core.gradient_coefficients returns them."""
    T, S = _c(T), _c(S)
    nt, nz, plane = T.shape
    dt = abi.DTYPE_F64 if T.dtype == np.float64 else F32_MODES[f32_mode]
    strides = (0, 0, 0)
    if p is not None:
        p = _c(p, np.float64)
        strides = {(nz,): (0, 1, 0), (nz, plane): (0, plane, 1),
                   (nt, nz, plane): (nz * plane, plane, 1)}.get(p.shape, (0, 0, 0))
    coef = _c(coef, np.float64)
    out = np.empty((nt, nz, plane), dtype=np.float64)
    rc = load().mlx_stratification(_p(T), _p(S), dt, _p(p), *strides, abi.EOS_IDS[eos],
                                   {"n2": abi.STRAT_N2, "turner": abi.STRAT_TURNER}[func], _p(coef),
                                   int(uniform), float(two_dx), float(gravity), nt, nz, plane,
                                   _p(out), None)
    _check(rc, "mlx_stratification")
    return out


def adjust_negative_n2(n2, lead0_rows, dz=None):
    """(adjusted, speed or None) of the host build's mlx_adjust_negative_n2; n2 (nt, nz, plane)."""
    n2 = _c(n2, np.float64)
    nt, nz, plane = n2.shape
    adjusted = np.empty_like(n2)
    speed = None if dz is None else np.empty((nt, plane), dtype=np.float64)
    dz = _c(dz, np.float64)
    rc = load().mlx_adjust_negative_n2(_p(n2), nt, nz, plane, int(lead0_rows), _p(dz), _p(adjusted),
                                       _p(speed), None)
    _check(rc, "mlx_adjust_negative_n2")
    return adjusted, speed


def wave_speed_where_time0(n2_t0, speed):
    n2_t0, speed = _c(n2_t0, np.float64), _c(speed, np.float64)
    nz, plane = n2_t0.shape
    nt = speed.shape[0]
    out = np.empty((nz, plane, nt), dtype=np.float64)
    _check(load().mlx_wave_speed_where_time0(_p(n2_t0), _p(speed), nt, nz, plane, _p(out), None),
           "mlx_wave_speed_where_time0")
    return out
