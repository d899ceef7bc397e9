"""Array-level API: the HIP kernels on torch device tensors.

torch is used only as the device-array container (allocation, streams); every
number is produced by libmomlevel_hip.so.  All functions enqueue on torch's
current stream OF THE DEVICE THAT OWNS THE OPERANDS (which need not be torch's current
device) and return device tensors without synchronising.

Layout: ``(time, z_l, yh, xh)`` C-contiguous, x fastest.  A
``(z_l, yh, xh)`` tensor passed where a 4-D field is expected is broadcast over
time (the held field of the thermosteric / halosteric variants).
"""

import numpy as np
import torch

from . import _lib, hostio
from ._lib import (
    DTYPE_F32,
    DTYPE_F32_UPCAST,
    DTYPE_F64,
    DTYPE_T32_S64,
    DTYPE_T64_S32,
    EOS_IDS,
    FUNC_IDS,
    P_FULL3D,
    P_FULL4D,
    P_SCALAR,
    P_ZPROF,
    MomlevelHipError,
)

F32_MODES = {"faithful": DTYPE_F32, "upcast": DTYPE_F32_UPCAST}
# K1's default time steps per block (csrc/momlevel_hip.hip kTChunk / kTChunkHeld; reported by
# bench.py, kept in step by tests/test_host_logic.py)
K1_TCHUNK = {"steric": 32, "held": 64}
ARITH_FLAGS = {"exact": 0, "fused": _lib.FLAG_FMA}


def arith_default(kernel="k0", dtype=None):
    """Arithmetic of the Wright density when the caller does not choose (``arith=None``).

    "exact": numpy's operator-for-operator evaluation, bit-identical to the reference.
    "fused": MLX_FLAG_FMA -- contracted multiply-adds and a refined hardware reciprocal: on float64 theta/S the
    whole expression (<= 2 ulp from numpy on rho, two thirds of the VALU work per cell); on float32
    theta/S in numpy's mixed precision (``f32_mode="faithful"``) the float32 polynomial is kept
    exactly as numpy rounds it and only the float64 tail is fused (a few float64 ulp from numpy's
    own value on float32 input); with ``f32_mode="upcast"`` it is float64 arithmetic on the
    float32 values.  Parity gate either way: 1e-10 relative.

    Default policy (MOMLEVEL_AMD_ARITH unset):
      * K1, the global sums (``domain="global"``: masso(t), src/momlevel/steric.py:134-147) ->
        "fused".  No global result was ever bit-identical to numpy -- the order of summation over
        (z,y,x) already differs at the 1e-15 level -- so exact arithmetic bought nothing there and
        kept the held-field variants, the one-pass decomposition and every float32 sum on the fp64
        VALU bound.  masso(t=0) == masso0 and steric[t=0] == 0.0 hold exactly in this mode too (one
        expression tree in every kernel).
      * K0 / K2, the pointwise outputs (rho, delta_rho, local eta) -> "exact": those ARE
        bit-identical to numpy and stay so.
    MOMLEVEL_AMD_ARITH=exact|fused overrides the policy for every kernel and dtype.
    """
    import os

    mode = os.environ.get("MOMLEVEL_AMD_ARITH")
    if mode is None:
        return "fused" if kernel == "k1" else "exact"
    if mode not in ARITH_FLAGS:
        raise ValueError(f"MOMLEVEL_AMD_ARITH must be 'exact' or 'fused', got '{mode}'")
    return mode


def _arith_flag(arith, kernel="k0", dtype=None):
    """``dtype``: the MLX_DTYPE_* code of the launch (or a torch dtype).  theta and salinity of
    different dtypes have exact kernels only: the default policy and MOMLEVEL_AMD_ARITH fall back to
    "exact" there, an explicit arith="fused" is an error."""
    if dtype in (DTYPE_T32_S64, DTYPE_T64_S32):
        if arith == "fused":
            raise ValueError("arith='fused' is not available for thetao / so of different dtypes")
        return 0
    if arith is None:
        arith = arith_default(kernel, dtype)
    try:
        return ARITH_FLAGS[arith]
    except KeyError:
        raise ValueError(f"arith must be 'exact' or 'fused', got '{arith}'") from None


def require_device():
    """The product path needs the HIP library AND a GPU; fail loudly otherwise."""
    _lib.load()
    if not torch.cuda.is_available():
        raise MomlevelHipError(
            "no HIP device visible: momlevel_amd computes on MI355X only (no CPU fallback)"
        )


def _stream(device):
    """Raw hipStream_t of torch's current stream ON ``device`` (not on the current device: the
    operands may live on another GPU of the node)."""
    return torch.cuda.current_stream(device).cuda_stream


def _on(device):
    """Context for a launch: kernels are enqueued with ``device`` current, so that the library's
    hipLaunchKernelGGL targets the GPU that owns the operands whatever torch's current device is."""
    return torch.cuda.device(device)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _f64(x, device):
    """Small time-invariant operand -> contiguous float64 device tensor."""
    if isinstance(x, torch.Tensor) and x.is_cuda:
        return x.to(device=device, dtype=torch.float64).contiguous()
    if not isinstance(x, torch.Tensor):
        x = np.asarray(x, dtype=np.float64)
    return hostio.to_device(x, device, torch.float64).contiguous()  # host data: owned staging


def _dtype_code(t, f32_mode):
    if t.dtype == torch.float64:
        return DTYPE_F64
    if t.dtype == torch.float32:
        return F32_MODES[f32_mode]
    raise TypeError(f"thetao/so must be float64 or float32, got {t.dtype}")


def _field(x, nz, ny, nx, name):
    """Validate a streamed field; return (tensor, nt or None, time stride in elements)."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise TypeError(f"{name} must be a CUDA/HIP torch tensor")
    if x.dim() == 3:
        if tuple(x.shape) != (nz, ny, nx):
            raise ValueError(f"{name} has shape {tuple(x.shape)}, expected {(nz, ny, nx)}")
        return x.contiguous(), None, 0
    if x.dim() != 4 or tuple(x.shape[1:]) != (nz, ny, nx):
        raise ValueError(f"{name} has shape {tuple(x.shape)}, expected (nt,{nz},{ny},{nx})")
    inner_ok = x.stride(3) == 1 and x.stride(2) == nx and x.stride(1) == ny * nx
    if not inner_ok or x.stride(0) < 0:
        x = x.contiguous()
    return x, x.shape[0], (x.stride(0) if x.shape[0] > 1 else nz * ny * nx)


def _pair(T, S, f32_mode):
    """Common shape logic of the (thetao, so) pair."""
    for name, x in (("thetao", T), ("so", S)):
        if x.dtype not in (torch.float32, torch.float64):
            raise TypeError(f"{name} must be float64 or float32, got {x.dtype}")
    mixed = T.dtype != S.dtype
    if mixed and f32_mode != "faithful":  # "upcast": float64 arithmetic on the stored values
        T, S, mixed = T.double(), S.double(), False
    if (isinstance(T, torch.Tensor) and isinstance(S, torch.Tensor) and T.is_cuda and S.is_cuda
            and T.device != S.device):
        raise ValueError(f"thetao is on {T.device} but so on {S.device}: operands of one call "
                         "must live on one GPU")
    shape3 = tuple(T.shape[-3:])
    nz, ny, nx = shape3
    T, ntT, sT = _field(T, nz, ny, nx, "thetao")
    S, ntS, sS = _field(S, nz, ny, nx, "so")
    if ntT is not None and ntS is not None and ntT != ntS:
        raise ValueError("thetao and so disagree on the number of time steps")
    nt = ntT if ntT is not None else ntS
    squeeze = nt is None
    if squeeze:
        nt = 1
    if nt == 0 or nz * ny * nx == 0:
        raise ValueError(f"empty field: shape {(nt, nz, ny, nx)} has no cells")
    if mixed:
        # numpy evaluates each field's part of the polynomial in that field's precision and joins
        # them in float64 (csrc/eos_device.hpp wright_density_mixed); exact kernels only
        dt = DTYPE_T32_S64 if T.dtype == torch.float32 else DTYPE_T64_S32
    else:
        dt = _dtype_code(T, f32_mode)
    return T, S, nt, nz, ny, nx, sT, sS, dt, squeeze


def _pressure(p, nt, nz, ny, nx, device, allow4d):
    """Classify the pressure operand -> (tensor, p_mode)."""
    if p is None:
        return None, P_SCALAR
    p = _f64(p, device)
    if p.numel() == 1:
        return p.reshape(1), P_SCALAR
    while p.dim() > 3 and p.shape[0] == 1:  # (1,nz,1,1): calc_rho's 4-D broadcast of a z profile
        p = p[0]
    shape = tuple(p.shape)
    if shape in ((nz,), (nz, 1, 1)):  # NB at THIS level a bare (nz,) is a z profile by contract
        return p.reshape(nz), P_ZPROF
    if shape == (nz, ny, nx):
        return p, P_FULL3D
    if allow4d and shape == (nt, nz, ny, nx):
        return p, P_FULL4D
    # anything else that broadcasts against (nz,ny,nx): materialise it (tiny vs the 4-D fields)
    try:
        return p.expand(nz, ny, nx).contiguous(), P_FULL3D
    except RuntimeError:
        pass
    if allow4d:
        return p.expand(nt, nz, ny, nx).contiguous(), P_FULL4D
    raise ValueError(f"pressure of shape {shape} does not broadcast to {(nz, ny, nx)}")


def eos_map(T, S, p, eos="wright", func="density", f32_mode="faithful", arith=None):
    """K0: pointwise EOS function over a (nt,nz,ny,nx) or (nz,ny,nx) grid -> float64.
    ``arith``: "exact" | "fused" | None = arith_default("k0", dtype) = exact; applies to the Wright
    density only."""
    require_device()
    T, S, nt, nz, ny, nx, sT, sS, dt, squeeze = _pair(T, S, f32_mode)
    pt, p_mode = _pressure(p, nt, nz, ny, nx, T.device, allow4d=True)
    if dt in (DTYPE_T32_S64, DTYPE_T64_S32):
        # thetao and so of different dtypes: numpy's promotion per sub-expression is the promote
        # kernel's job (one cell per thread over the broadcast operands); exact arithmetic only --
        # an explicit arith="fused" is an error here as it is in K1 / K2
        _arith_flag(arith, "k0", dt)
        full = (nt, nz, ny, nx)
        ops = [(x if x.dim() == 4 else x.unsqueeze(0)).expand(full).reshape(-1) for x in (T, S)]
        if pt is not None and pt.numel() > 1:
            pt = {P_ZPROF: lambda q: q.reshape(1, nz, 1, 1), P_FULL3D: lambda q: q.reshape(1, nz, ny, nx),
                  P_FULL4D: lambda q: q}[p_mode](pt).expand(full).reshape(-1)
        out = eos_map_promote(ops[0], ops[1], pt, eos=eos, func=func).reshape(full)
        return out[0] if squeeze else out
    flags = (_arith_flag(arith, "k0", dt)
             if (func == "density" and eos.lower() == "wright") else 0)
    out = torch.empty((nt, nz, ny, nx), dtype=torch.float64, device=T.device)
    with _on(T.device):
        rc = _lib.load().mlx_eos_map(
            _ptr(T), _ptr(S), dt, _ptr(pt), p_mode, EOS_IDS[eos.lower()], FUNC_IDS[func],
            nt, nz, ny * nx, sT, sS, flags, _ptr(out), _stream(T.device),
        )
    _lib.check(rc, "mlx_eos_map")
    return out[0] if squeeze else out


def inverse_barometer(T, S, p, gravity=9.8, eos="wright", f32_mode="faithful"):
    """pso * (-1 / (rho(T,S,pso) * gravity)) on a (nt,nz,ny,nx)/(nz,ny,nx) grid -> float64."""
    require_device()
    T, S, nt, nz, ny, nx, sT, sS, dt, squeeze = _pair(T, S, f32_mode)
    pt, p_mode = _pressure(p, nt, nz, ny, nx, T.device, allow4d=True)
    out = torch.empty((nt, nz, ny, nx), dtype=torch.float64, device=T.device)
    with _on(T.device):
        rc = _lib.load().mlx_inverse_barometer(
            _ptr(T), _ptr(S), dt, _ptr(pt), p_mode, EOS_IDS[eos.lower()], float(gravity),
            nt, nz, ny * nx, sT, sS, _ptr(out), _stream(T.device),
        )
    _lib.check(rc, "mlx_inverse_barometer")
    return out[0] if squeeze else out


def eos_map_promote(T, S, p, eos="wright", func="density", gravity=9.8):
    """K0 under numpy's type promotion (mlx_eos_map_promote; csrc/eos_promote.hpp) for the dtype
    combinations mlx_eos_map does not cover.  Each operand is a python float / int (a WEAK scalar:
    it takes the dtype of the arrays it meets) or a float32 / float64 device tensor of n elements or
    of ONE element (used for every cell); ``p`` may be None for the linear EOS.  ``func`` may also
    be "inverse_barometer" (``gravity`` a python float) or, for the linear EOS, "density_ref" (``p``
    then carries the constant term RHO_T0_S0 - rho_ref of eos/linear.py:55).  Returns a device tensor of n elements in
    numpy's result dtype (float32 when no float64 array takes part) holding numpy's values."""
    import ctypes

    require_device()
    if p is None and (eos.lower() != "linear" or func in ("inverse_barometer", "density_ref")):
        raise TypeError("p must not be None (only the linear EOS ignores the pressure)")
    tensors = [x for x in (T, S, p) if isinstance(x, torch.Tensor)]
    if not tensors:
        raise TypeError("at least one operand must be a device tensor")
    device = tensors[0].device
    n = max(int(x.numel()) for x in tensors)
    keep, args = [], []
    for name, x in (("T", T), ("S", S), ("p", p)):
        if x is None:
            if name != "p":
                raise TypeError(f"{name} must not be None")
            args += [None, _lib.KIND_WEAK, 0]
        elif isinstance(x, torch.Tensor):
            if not x.is_cuda or x.device != device:
                raise ValueError("operands of one call must live on one GPU")
            if x.dtype not in (torch.float32, torch.float64):
                raise TypeError(f"{name} must be float32 or float64, not {x.dtype}")
            if x.numel() not in (1, n):
                raise ValueError(f"{name} has {x.numel()} elements, expected 1 or {n}")
            x = x.contiguous()
            keep.append(x)
            args += [x.data_ptr(), _lib.KIND_F32 if x.dtype == torch.float32 else _lib.KIND_F64,
                     1 if x.numel() == n and n > 1 else 0]
        else:
            w = ctypes.c_double(float(x))
            keep.append(w)
            args += [ctypes.addressof(w), _lib.KIND_WEAK, 0]
    fid = {"inverse_barometer": _lib.FUNC_IBH, "density_ref": _lib.FUNC_DENSITY_REF}.get(func)
    if fid is None:
        fid = FUNC_IDS[func]
    out = torch.empty(n, dtype=torch.float64, device=device)
    kind = ctypes.c_int(-1)
    with _on(device):
        rc = _lib.load().mlx_eos_map_promote(*args, EOS_IDS[eos.lower()], fid, float(gravity), n,
                                             _ptr(out), ctypes.byref(kind), _stream(device))
    _lib.check(rc, "mlx_eos_map_promote")
    if kind.value == _lib.KIND_F32:  # the kernel stored n float32 values at the start of the buffer
        return out.view(torch.float32)[:n]
    return out


def gradient_coefficients(z):
    """numpy.gradient's per-level coefficients for ``edge_order=2`` along a coordinate ``z`` (what
    xarray's ``differentiate(zcoord, edge_order=2)`` evaluates; derived.py:399-400, :752-753),
    computed on the host from the coordinate with numpy's own expressions, operator for operator.
    Returns (coef (nz,3) float64, uniform, two_dx): level k's derivative is
    ``a*f[k-1] + b*f[k] + c*f[k+1]`` (level 0: on levels 0,1,2; level nz-1: on the last three);
    for evenly spaced levels -- numpy's test: every np.diff equals the first -- the interior is
    ``(f[k+1] - f[k-1]) / two_dx`` instead and only the edge rows of ``coef`` are used."""
    z = np.asarray(z)
    if z.ndim != 1:
        raise ValueError("distances must be either scalars or 1d")
    n = z.shape[0]
    if n < 3:
        raise ValueError("Shape of array too small to calculate a numerical gradient, "
                         "at least (edge_order + 1) elements are required.")
    if np.issubdtype(z.dtype, np.integer):
        z = z.astype(np.float64)
    diffx = np.diff(z)
    coef = np.zeros((n, 3), dtype=np.float64)
    if (diffx == diffx[0]).all():
        dx = diffx[0]
        coef[0] = (-1.5 / dx, 2.0 / dx, -0.5 / dx)
        coef[-1] = (0.5 / dx, -2.0 / dx, 1.5 / dx)
        return coef, True, float(2.0 * dx)
    dx1, dx2 = diffx[0:-1], diffx[1:]
    coef[1:-1, 0] = -(dx2) / (dx1 * (dx1 + dx2))
    coef[1:-1, 1] = (dx2 - dx1) / (dx1 * dx2)
    coef[1:-1, 2] = dx1 / (dx2 * (dx1 + dx2))
    dx1, dx2 = diffx[0], diffx[1]
    coef[0] = (-(2.0 * dx1 + dx2) / (dx1 * (dx1 + dx2)), (dx1 + dx2) / (dx1 * dx2),
               -dx1 / (dx2 * (dx1 + dx2)))
    dx1, dx2 = diffx[-2], diffx[-1]
    coef[-1] = ((dx2) / (dx1 * (dx1 + dx2)), -(dx2 + dx1) / (dx1 * dx2),
                (2.0 * dx2 + dx1) / (dx2 * (dx1 + dx2)))
    return coef, False, 0.0


def stratification(T, S, p, z, func="n2", eos="wright", gravity=-9.8, f32_mode="faithful"):
    """mlx_stratification: N^2 (``func="n2"``, derived.py:328-411) or the stability angle
    (``"turner"``, derived.py:714-766) in one pass over device fields laid out (nt, nz, plane)
    -- z the middle axis, nt / plane the products of the dimensions before / after it.  ``z`` the
    level coordinate (nz,); ``p`` None (linear EOS), one value, (nz,) or a (nt, nz, plane) /
    (nz, plane) float64 device tensor.  Returns (nt, nz, plane) float64."""
    require_device()
    if not (isinstance(T, torch.Tensor) and isinstance(S, torch.Tensor) and T.is_cuda and S.is_cuda):
        raise TypeError("thetao and so must be CUDA/HIP torch tensors")
    if T.dim() != 3 or T.shape != S.shape:
        raise ValueError(f"thetao {tuple(T.shape)} and so {tuple(S.shape)} must both be (nt, nz, plane)")
    if T.dtype != S.dtype:
        raise TypeError("thetao and so of different dtypes: convert one of them")
    dt = _dtype_code(T, f32_mode)
    T, S = T.contiguous(), S.contiguous()
    nt, nz, plane = (int(v) for v in T.shape)
    if nt == 0 or plane == 0:
        raise ValueError(f"empty field: shape {(nt, nz, plane)} has no cells")
    coef, uniform, two_dx = gradient_coefficients(z)
    if coef.shape[0] != nz:
        raise ValueError("when 1d, distances must match the length of the corresponding dimension")
    if eos.lower() == "linear":
        pt, strides = None, (0, 0, 0)
    else:
        if p is None:
            raise TypeError("p must not be None for the Wright EOS")
        pt = _f64(p, T.device)
        shape = tuple(pt.shape)
        if pt.numel() == 1:
            pt, strides = pt.reshape(1), (0, 0, 0)
        elif shape == (nz,):
            strides = (0, 1, 0)
        elif shape == (nz, plane):
            strides = (0, plane, 1)
        elif shape == (nt, nz, plane):
            strides = (nz * plane, plane, 1)
        else:
            raise ValueError(f"pressure of shape {shape} is none of (), ({nz},), ({nz},{plane}), "
                             f"({nt},{nz},{plane})")
    coef_dev = _f64(coef, T.device)
    out = torch.empty((nt, nz, plane), dtype=torch.float64, device=T.device)
    with _on(T.device):
        rc = _lib.load().mlx_stratification(
            _ptr(T), _ptr(S), dt, _ptr(pt), *strides, EOS_IDS[eos.lower()],
            {"n2": _lib.STRAT_N2, "turner": _lib.STRAT_TURNER}[func], _ptr(coef_dev),
            int(uniform), float(two_dx), float(gravity), nt, nz, plane, _ptr(out),
            _stream(T.device))
    _lib.check(rc, "mlx_stratification")
    return out


def adjust_negative_n2(n2, lead0_rows, dz=None, want_adjusted=True):
    """mlx_adjust_negative_n2 on a (nt, nz, plane) float64 device field (see the header for
    ``lead0_rows``).  Returns (adjusted or None, speed or None): ``speed`` (nt, plane), the column
    sum of derived.py:822, when ``dz`` (nz, plane) is given."""
    require_device()
    if n2.dim() != 3:
        raise ValueError("n2 must be (nt, nz, plane)")
    n2 = n2.to(torch.float64).contiguous()
    nt, nz, plane = (int(v) for v in n2.shape)
    adjusted = torch.empty_like(n2) if want_adjusted else None
    speed = dzt = None
    if dz is not None:
        dzt = _f64(dz, n2.device)
        if tuple(dzt.shape) != (nz, plane):
            raise ValueError(f"dz has shape {tuple(dzt.shape)}, expected {(nz, plane)}")
        speed = torch.empty((nt, plane), dtype=torch.float64, device=n2.device)
    with _on(n2.device):
        rc = _lib.load().mlx_adjust_negative_n2(_ptr(n2), nt, nz, plane, int(lead0_rows),
                                                _ptr(dzt), _ptr(adjusted), _ptr(speed),
                                                _stream(n2.device))
    _lib.check(rc, "mlx_adjust_negative_n2")
    return adjusted, speed


def wave_speed_where_time0(n2_t0, speed):
    """mlx_wave_speed_where_time0: (nz, plane) condition field x (nt, plane) speeds ->
    (nz, plane, nt)."""
    require_device()
    nz, plane = (int(v) for v in n2_t0.shape)
    nt = int(speed.shape[0])
    out = torch.empty((nz, plane, nt), dtype=torch.float64, device=speed.device)
    n2_t0, speed = n2_t0.contiguous(), speed.contiguous()
    with _on(speed.device):
        rc = _lib.load().mlx_wave_speed_where_time0(_ptr(n2_t0), _ptr(speed), nt, nz, plane,
                                                    _ptr(out), _stream(speed.device))
    _lib.check(rc, "mlx_wave_speed_where_time0")
    return out


def skip_dry_default():
    """Land / sub-bottom skipping is exact, so it is on unless MOMLEVEL_AMD_SKIP_DRY=0."""
    import os

    return os.environ.get("MOMLEVEL_AMD_SKIP_DRY", "1") != "0"


def _launch_flags(skip_dry, arith, t_chunk, kernel, dtype):
    if skip_dry is None:
        skip_dry = skip_dry_default()
    flags = (_lib.FLAG_SKIP_DRY if skip_dry else 0) | _arith_flag(arith, kernel, dtype)
    if t_chunk:
        flags |= _lib.flag_tchunk(t_chunk)
    return flags


def steric_global_masso(T, S, vol0, p, eos="wright", f32_mode="faithful", events=None,
                        skip_dry=None, arith=None, t_chunk=0):
    """K1: masso[t] = sum_{z,y,x} rho(T,S,p) * vol0  (skipna) -> (nt,) float64.

    ``events=(start, end)``: two ``torch.cuda.Event(enable_timing=True)`` recorded on the
    launch stream immediately around the kernel launches (bench.py's per-launch timing).
    ``skip_dry``: MLX_FLAG_SKIP_DRY (None = the default policy, on); results are bit-identical
    either way.  ``arith``: "exact" | "fused" (None = arith_default("k1"): fused).  ``t_chunk``: tuning
    hint, time steps per block (multiple of 8; 0 = library default); never changes a result.
    ``p`` may be time dependent, (nt,nz,ny,nx)-broadcastable (a DataArray ``patm``).
    """
    require_device()
    T, S, nt, nz, ny, nx, sT, sS, dt, _ = _pair(T, S, f32_mode)
    flags = _launch_flags(skip_dry, arith, t_chunk, "k1", dt)
    vol0 = _f64(vol0, T.device)
    if tuple(vol0.shape) != (nz, ny, nx):
        raise ValueError(f"vol0 has shape {tuple(vol0.shape)}, expected {(nz, ny, nx)}")
    pt, p_mode = _pressure(p, nt, nz, ny, nx, T.device, allow4d=True)
    lib = _lib.load()
    nbytes = lib.mlx_steric_global_workspace_bytes(nt, nz, ny * nx)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=T.device)
    out = torch.empty(nt, dtype=torch.float64, device=T.device)
    with _on(T.device):
        stream = torch.cuda.current_stream(T.device)
        if events is not None:
            events[0].record(stream)
        rc = lib.mlx_steric_global(
            _ptr(T), _ptr(S), dt, _ptr(vol0), _ptr(pt), p_mode, EOS_IDS[eos.lower()],
            nt, nz, ny * nx, sT, sS, flags,
            _ptr(out), _ptr(ws), nbytes, stream.cuda_stream,
        )
        if events is not None:
            events[1].record(stream)
    _lib.check(rc, "mlx_steric_global")
    return out


DECOMP_ROWS = ("steric", "thermosteric", "halosteric", "heat")


def steric_global_decomp(T, S, T0, S0, vol0, p, eos="wright", f32_mode="faithful", events=None,
                         skip_dry=None, arith=None, t_chunk=0):
    """K1, all variants in one pass over theta/S: (4, nt) float64, rows DECOMP_ROWS =
    masso of steric / thermosteric (S held at S0) / halosteric (theta held at T0), and
    sum(theta*vol0) (the heat-content integrand; an extension, not in momlevel).  Rows 0-2 are
    bit-identical to three steric_global_masso calls."""
    require_device()
    T, S, nt, nz, ny, nx, sT, sS, dt, _ = _pair(T, S, f32_mode)
    flags = _launch_flags(skip_dry, arith, t_chunk, "k1", dt)
    if T.dim() != 4 or S.dim() != 4:
        raise ValueError("steric_global_decomp streams both fields: thetao and so must be 4-D")
    dev = T.device
    T0 = T0.to(device=dev, dtype=T.dtype).contiguous()
    S0 = S0.to(device=dev, dtype=S.dtype).contiguous()
    if tuple(T0.shape) != (nz, ny, nx) or tuple(S0.shape) != (nz, ny, nx):
        raise ValueError(f"T0 and S0 must be {(nz, ny, nx)}")
    vol0 = _f64(vol0, dev)
    if tuple(vol0.shape) != (nz, ny, nx):
        raise ValueError(f"vol0 has shape {tuple(vol0.shape)}, expected {(nz, ny, nx)}")
    pt, p_mode = _pressure(p, nt, nz, ny, nx, dev, allow4d=True)
    lib = _lib.load()
    nbytes = lib.mlx_steric_global_decomp_workspace_bytes(nt, nz, ny * nx)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
    out = torch.empty((4, nt), dtype=torch.float64, device=dev)
    with _on(dev):
        stream = torch.cuda.current_stream(dev)
        if events is not None:
            events[0].record(stream)
        rc = lib.mlx_steric_global_decomp(
            _ptr(T), _ptr(S), _ptr(T0), _ptr(S0), dt, _ptr(vol0), _ptr(pt), p_mode,
            EOS_IDS[eos.lower()], nt, nz, ny * nx, sT, sS, flags, _ptr(out), _ptr(ws), nbytes,
            stream.cuda_stream,
        )
        if events is not None:
            events[1].record(stream)
    _lib.check(rc, "mlx_steric_global_decomp")
    return out


def stream_probe(a, b, out=None):
    """Measurement aid: out = a + b with K2's 16-byte nt loads/stores (16 B read + 8 B written per
    element); see mlx_stream_probe."""
    require_device()
    if out is None:
        out = torch.empty_like(a)
    with _on(a.device):
        rc = _lib.load().mlx_stream_probe(_ptr(a), _ptr(b), a.numel(), _ptr(out),
                                          _stream(a.device))
    _lib.check(rc, "mlx_stream_probe")
    return out


def stream_probe_mix(a, b=None, out=None, write=True):
    """Measurement aid: one (``b`` None) or two float32 / float64 streams in, one float64 stream out
    (``write``) or none -- the read:write mix of a held-field / float32 local pass; see
    mlx_stream_probe_mix.  Returns ``out`` (a 1-element tensor, untouched, when ``write`` is False)."""
    require_device()
    dt = _dtype_code(a, "faithful")
    if b is not None and (b.dtype != a.dtype or b.numel() != a.numel()):
        raise ValueError("a and b must agree in dtype and size")
    if out is None:
        out = torch.empty(a.shape if write else (1,), dtype=torch.float64, device=a.device)
    with _on(a.device):
        rc = _lib.load().mlx_stream_probe_mix(_ptr(a), _ptr(b), dt, a.numel(), _ptr(out),
                                              int(bool(write)), _stream(a.device))
    _lib.check(rc, "mlx_stream_probe_mix")
    return out


def valu_probe(iters=4096, device="cuda"):
    """Measurement aid: enqueue the float64 VALU issue-rate probe (mlx_valu_probe) on ``device``'s
    current stream; returns the number of v_fma_f64 lane-instructions the launch issues."""
    require_device()
    device = torch.device(device)
    import ctypes

    out = torch.zeros(1, dtype=torch.float64, device=device)
    n = ctypes.c_int64(0)
    with _on(device):
        rc = _lib.load().mlx_valu_probe(int(iters), _ptr(out), ctypes.byref(n), _stream(device))
    _lib.check(rc, "mlx_valu_probe")
    return int(n.value)


def fold_mask(rho0, vol0):
    """rho0m = where(vol0 notnull, rho0, NaN) -- prepared once per reference state."""
    require_device()
    rho0 = _f64(rho0, rho0.device)
    vol0 = _f64(vol0, rho0.device)
    out = torch.empty_like(rho0)
    with _on(rho0.device):
        rc = _lib.load().mlx_fold_mask(_ptr(rho0), _ptr(vol0), rho0.numel(), _ptr(out),
                                       _stream(rho0.device))
    _lib.check(rc, "mlx_fold_mask")
    return out


def steric_local(T, S, rho0m, vol0_surface, p, neg_inv_rhozero, dz=None, z_i=None,
                 deptho=None, eos="wright", f32_mode="faithful", want_delta_rho=True,
                 delta_rho_out=None, eta_out=None, skip_dry=None, arith=None):
    """K2: (delta_rho (nt,nz,ny,nx) or None, eta (nt,ny,nx)).  ``skip_dry``, ``arith``: see
    steric_global_masso.  ``p`` may be time dependent (4-D)."""
    require_device()
    T, S, nt, nz, ny, nx, sT, sS, dt, _ = _pair(T, S, f32_mode)
    flags = _launch_flags(skip_dry, arith, 0, "k2", dt)
    dev = T.device
    rho0m = _f64(rho0m, dev)
    vol0_surface = _f64(vol0_surface, dev)
    if tuple(rho0m.shape) != (nz, ny, nx) or tuple(vol0_surface.shape) != (ny, nx):
        raise ValueError("rho0m must be (nz,ny,nx) and vol0_surface (ny,nx)")
    pt, p_mode = _pressure(p, nt, nz, ny, nx, dev, allow4d=True)
    if dz is not None:
        dz = _f64(dz, dev)
        if tuple(dz.shape) != (nz, ny, nx):
            raise ValueError("dz must be (nz,ny,nx)")
    else:
        z_i = _f64(z_i, dev)
        deptho = _f64(deptho, dev)
        if z_i.numel() != nz + 1 or tuple(deptho.shape) != (ny, nx):
            raise ValueError("z_i must have nz+1 entries and deptho be (ny,nx)")
    drho = None
    if want_delta_rho:
        drho = delta_rho_out
        if drho is None:
            drho = torch.empty((nt, nz, ny, nx), dtype=torch.float64, device=dev)
    eta = eta_out if eta_out is not None else torch.empty(
        (nt, ny, nx), dtype=torch.float64, device=dev
    )
    with _on(dev):
        rc = _lib.load().mlx_steric_local(
            _ptr(T), _ptr(S), dt, _ptr(rho0m), _ptr(vol0_surface), _ptr(dz), _ptr(z_i),
            _ptr(deptho), _ptr(pt), p_mode, EOS_IDS[eos.lower()], float(neg_inv_rhozero),
            nt, nz, ny * nx, sT, sS, flags,
            _ptr(drho), _ptr(eta), _stream(dev),
        )
    _lib.check(rc, "mlx_steric_local")
    return drho, eta


LOCAL_DECOMP_ROWS = ("steric", "thermosteric", "halosteric")


def steric_local_decomp(T, S, T0, S0, rho0m, vol0_surface, p, neg_inv_rhozero, dz=None, z_i=None,
                        deptho=None, eos="wright", f32_mode="faithful", want_delta_rho=True,
                        delta_rho_out=None, eta_out=None, skip_dry=None, arith=None):
    """K2, all variants in one pass over theta/S: (delta_rho (3,nt,nz,ny,nx) or None,
    eta (3,nt,ny,nx)), variant order LOCAL_DECOMP_ROWS; each field bit-identical to its
    steric_local call.  ``delta_rho_out`` / ``eta_out``: optional (3, nt, ...) float64 device
    tensors (or views whose variant axis has any stride, e.g. ``full[:, t0:t1]``)."""
    require_device()
    T, S, nt, nz, ny, nx, sT, sS, dt, _ = _pair(T, S, f32_mode)
    flags = _launch_flags(skip_dry, arith, 0, "k2", dt)
    if T.dim() != 4 or S.dim() != 4:
        raise ValueError("steric_local_decomp streams both fields: thetao and so must be 4-D")
    dev = T.device
    T0 = T0.to(device=dev, dtype=T.dtype).contiguous()
    S0 = S0.to(device=dev, dtype=S.dtype).contiguous()
    if tuple(T0.shape) != (nz, ny, nx) or tuple(S0.shape) != (nz, ny, nx):
        raise ValueError(f"T0 and S0 must be {(nz, ny, nx)}")
    rho0m = _f64(rho0m, dev)
    vol0_surface = _f64(vol0_surface, dev)
    if tuple(rho0m.shape) != (nz, ny, nx) or tuple(vol0_surface.shape) != (ny, nx):
        raise ValueError("rho0m must be (nz,ny,nx) and vol0_surface (ny,nx)")
    pt, p_mode = _pressure(p, nt, nz, ny, nx, dev, allow4d=True)
    if dz is not None:
        dz = _f64(dz, dev)
        if tuple(dz.shape) != (nz, ny, nx):
            raise ValueError("dz must be (nz,ny,nx)")
    else:
        z_i = _f64(z_i, dev)
        deptho = _f64(deptho, dev)
        if z_i.numel() != nz + 1 or tuple(deptho.shape) != (ny, nx):
            raise ValueError("z_i must have nz+1 entries and deptho be (ny,nx)")

    def variant_major(x, shape):
        """(3, nt, ...) float64 device tensor whose per-variant fields are contiguous"""
        if tuple(x.shape) != shape or x.dtype != torch.float64 or x.device != dev:
            raise ValueError(f"output must be a float64 {shape} tensor on {dev}")
        if not x[0].is_contiguous() or x.stride(0) < x[0].numel():
            raise ValueError("each variant's field must be contiguous")
        return x

    drho = None
    if want_delta_rho:
        drho = delta_rho_out if delta_rho_out is not None else torch.empty(
            (3, nt, nz, ny, nx), dtype=torch.float64, device=dev)
        variant_major(drho, (3, nt, nz, ny, nx))
    eta = eta_out if eta_out is not None else torch.empty((3, nt, ny, nx), dtype=torch.float64,
                                                           device=dev)
    variant_major(eta, (3, nt, ny, nx))
    with _on(dev):
        rc = _lib.load().mlx_steric_local_decomp(
            _ptr(T), _ptr(S), _ptr(T0), _ptr(S0), dt, _ptr(rho0m), _ptr(vol0_surface), _ptr(dz),
            _ptr(z_i), _ptr(deptho), _ptr(pt), p_mode, EOS_IDS[eos.lower()],
            float(neg_inv_rhozero), nt, nz, ny * nx, sT, sS, flags,
            _ptr(drho), drho.stride(0) if drho is not None else 0, _ptr(eta), eta.stride(0),
            _stream(dev),
        )
    _lib.check(rc, "mlx_steric_local_decomp")
    return drho, eta


def nansum(x):
    """skipna sum of a float64 device tensor -> 0-d device tensor."""
    require_device()
    x = _f64(x, x.device).reshape(-1)
    lib = _lib.load()
    nbytes = lib.mlx_nansum_workspace_bytes(x.numel())
    ws = torch.empty(max(nbytes // 8, 1), dtype=torch.float64, device=x.device)
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    with _on(x.device):
        rc = lib.mlx_nansum(_ptr(x), x.numel(), _ptr(out), _ptr(ws), nbytes, _stream(x.device))
    _lib.check(rc, "mlx_nansum")
    return out[0]


def masso(rho, vol):
    """Standalone calc_masso: rho (nt, n3), vol (n3,) or (nt, n3) -> sum(rho*vol) [skipna], (nt,)."""
    require_device()
    rho = _f64(rho, rho.device)
    vol = _f64(vol, rho.device)
    if rho.dim() != 2:
        raise ValueError("rho must be (nt, n3)")
    nt, n3 = rho.shape
    if tuple(vol.shape) == (n3,):
        vstride = 0
    elif tuple(vol.shape) == (nt, n3):
        vstride = n3
    else:
        raise ValueError("vol must be (n3,) or (nt, n3)")
    lib = _lib.load()
    out = torch.empty(nt, dtype=torch.float64, device=rho.device)
    step = 32768  # the kernel's grid.y carries the time axis (<= 65535)
    nbytes = lib.mlx_steric_global_workspace_bytes(min(nt, step), 1, n3)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=rho.device)
    for t0 in range(0, nt, step):
        t1 = min(t0 + step, nt)
        with _on(rho.device):
            rc = lib.mlx_masso(_ptr(rho[t0:t1]), _ptr(vol[t0:t1] if vstride else vol), t1 - t0,
                               n3, vstride, _ptr(out[t0:t1]), _ptr(ws), nbytes,
                               _stream(rho.device))
        _lib.check(rc, "mlx_masso")
    return out


def group_weighted_mean(x, w, group_len, out=None):
    """Per-group weighted mean over the leading axis (annual_average's arithmetic): x (nt, ...)
    with nt = ngroups*group_len, w (nt,) -> (ngroups, ...); NaNs carry no weight."""
    require_device()
    x = _f64(x, x.device)
    w = _f64(w, x.device).reshape(-1)
    nt = x.shape[0]
    if nt % group_len or w.numel() != nt:
        raise ValueError("the time axis must hold whole groups and one weight per step")
    ngroups = nt // group_len
    n = x[0].numel()
    if out is None:
        out = torch.empty((ngroups,) + tuple(x.shape[1:]), dtype=torch.float64, device=x.device)
    with _on(x.device):
        rc = _lib.load().mlx_group_weighted_mean(_ptr(x), _ptr(w), ngroups, group_len, n,
                                                 _ptr(out), _stream(x.device))
    _lib.check(rc, "mlx_group_weighted_mean")
    return out


def calc_dz(z_i, depth, top=0.0, bottom=None, fraction=False):
    """derived.calc_dz core on device -> (nz, ny, nx)."""
    require_device()
    depth = _f64(depth, depth.device if isinstance(depth, torch.Tensor) else "cuda")
    z_i = _f64(z_i, depth.device)
    nz = z_i.numel() - 1
    ny, nx = depth.shape
    out = torch.empty((nz, ny, nx), dtype=torch.float64, device=depth.device)
    with _on(depth.device):
        rc = _lib.load().mlx_calc_dz(
            _ptr(z_i), _ptr(depth), nz, ny * nx, float(top),
            float(bottom) if bottom is not None else 0.0, int(bottom is not None),
            int(bool(fraction)), _ptr(out), _stream(depth.device),
        )
    _lib.check(rc, "mlx_calc_dz")
    return out


def synth_field(shape, dtype=torch.float64, *, seed, field_id, lo, scale, mask3d=None,
                t0=0, global_hw=None, origin=(0, 0), device="cuda", out=None):
    """Counter-based synthetic field (bench / full-size tests); see synthetic.py."""
    require_device()
    nt, nz, ny, nx = shape
    NY, NX = global_hw if global_hw is not None else (ny, nx)
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=device)
    code = DTYPE_F64 if out.dtype == torch.float64 else DTYPE_F32
    if mask3d is not None:
        mask3d = _f64(mask3d, out.device)
    with _on(out.device):
        rc = _lib.load().mlx_synth_field(
            _ptr(out), code, nt, nz, ny, nx, t0, NY, NX, origin[0], origin[1], seed, field_id,
            float(lo), float(scale), _ptr(mask3d), _stream(out.device),
        )
    _lib.check(rc, "mlx_synth_field")
    return out
