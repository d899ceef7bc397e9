"""numpy / torch / scalar front end of the pointwise EOS kernel (K0, mlx_eos_map).

The reference's EOS functions are numpy ufunc expressions with numpy broadcasting
(src/momlevel/eos/wright.py:23-165).  Here the same call signature routes to the
HIP kernel: host arrays are uploaded, evaluated on the MI355X and copied back;
device tensors stay on the device.  There is no host arithmetic fallback.

Being numpy expressions, the reference's functions compute in whatever dtypes numpy's promotion
gives their sub-expressions.  float64 fields, and float32 theta/S with a float64 pressure (the
steric path), run on the tuned kernel (mlx_eos_map).  Every other combination -- a python-float or
float32 pressure on float32 fields (calc_pdens on MOM6 output: the WHOLE expression is float32 and
so is the result), theta and salinity of different dtypes, python floats for theta or salinity --
runs on mlx_eos_map_promote, which restates the promotion rules exactly (csrc/eos_promote.hpp).
"""

import numpy as np
import torch

from .. import core, hostio
from ..labeled import check_field_dtype


def _as_tensor(x, device):
    if isinstance(x, torch.Tensor):
        x = x.to(device)
        return x if x.dtype in (torch.float32, torch.float64) else x.double()  # ints: see _kind
    a = hostio.as_plain(x)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return hostio.to_device(a, device)  # through our own page-locked staging


def _is_lazy(x):
    """array-shaped, readable only by slicing (dask / netCDF4 / h5py-like): never np.asarray'ed
    whole here -- the piecewise evaluation reads it piece by piece"""
    return (not isinstance(x, (np.ndarray, np.generic, torch.Tensor, list, tuple, bool, int, float))
            and all(hasattr(x, a) for a in ("shape", "dtype", "__getitem__")))


def _shape(x):
    return tuple(x.shape) if _is_lazy(x) else np.shape(x)


def _is_weak(x):
    """A python float / int: numpy (>= 2, NEP 50) lets it take the dtype of the arrays it meets.
    numpy scalars (np.float32(1.0), np.float64(1.0)) are NOT weak: they count as arrays."""
    return isinstance(x, (bool, int, float)) and not isinstance(x, np.generic)


def _kind(x):
    """"weak" | "f32" | "f64": how the operand enters numpy's promotion (None for a missing p).
    Integer and boolean ARRAYS compute as float64: numpy (2.x, NEP 50) promotes ``int_array *
    python_float`` to float64 -- the python float is weak only among floats -- so every
    sub-expression of such a field is float64, exactly as if it had been converted first (checked
    against numpy for bool / int16 / uint8 / int32 / int64: tests/test_host_logic.py).  float16
    arrays are refused: numpy would evaluate their part of the polynomial IN float16 (the weak
    constants take the array's dtype), which no kernel here reproduces."""
    if x is None:
        return None
    if _is_weak(x):
        return "weak"
    dt = x.dtype if isinstance(x, torch.Tensor) or _is_lazy(x) else hostio.as_plain(x).dtype
    name = check_field_dtype(dt, "operands")  # (byte order does not matter: ">f4" is float32)
    return "f32" if name == "float32" else "f64"


def _tuned_kernel_covers(kT, kS, kp, eos):
    """mlx_eos_map computes float64 fields (any pressure: it widens exactly) and, for the Wright EOS,
    float32 fields with a float64 pressure -- float64 out in both cases, as numpy.  Python floats
    for BOTH fields are float64 arithmetic too.  (The linear EOS on float32 fields is float32
    throughout in numpy, result included: the promote kernel.)"""
    if kT == "weak" and kS == "weak":
        return kp != "f32"
    if kT == "f64" and kS == "f64":
        return True
    return eos != "linear" and kT == "f32" and kS == "f32" and kp == "f64"


# host arrays above this size are evaluated in pieces along their leading axis, the pieces'
# uploads, kernels and result downloads overlapping (the link is full duplex)
_HOST_PIPELINE_ELEMS = 1 << 26
_HOST_CHUNK_ELEMS = 1 << 25  # 256 MiB of float64 per operand and piece


def _evaluate_host_chunked(eos, func, T, S, p, gravity):
    """Large host arrays: walk the leading axis of the broadcast shape in pieces.  Piece k+1 is
    staged and uploaded by a worker thread (hostio.Uploader) while piece k's kernel runs and piece
    k-1's result leaves on another (hostio.Downloader): both directions of the host link are busy
    at once, and the device never holds more than a few pieces (the result is a host array
    anyway)."""
    arrs = [x if (_is_weak(x) or _is_lazy(x)) else np.asarray(x)
            for x in (T, S, p if p is not None else 0.0)]
    shape = np.broadcast_shapes(*(_shape(a) for a in arrs))
    rows = max(1, _HOST_CHUNK_ELEMS // max(1, int(np.prod(shape[1:]))))
    bounds = [(i0, min(i0 + rows, shape[0])) for i0 in range(0, shape[0], rows)]
    device = torch.device("cuda", torch.cuda.current_device())
    main = torch.cuda.current_stream(device)

    def part(a, i0, i1):  # slice the leading axis unless the operand broadcasts along it
        if len(_shape(a)) == len(shape) and a.shape[0] == shape[0] and shape[0] > 1:
            return hostio.leading_slice(a, i0, i1)  # (a lazy operand is READ piece by piece, by the upload worker)
        return hostio.as_plain(a[...]) if _is_lazy(a) else a

    def pieces(i0, i1):  # (operands of the piece, which of them travel)
        ops = [part(arrs[0], i0, i1), part(arrs[1], i0, i1), None if p is None else part(arrs[2], i0, i1)]
        return ops, [k for k, x in enumerate(ops) if x is not None and not _is_weak(x)]

    out = None
    up = hostio.Uploader(device)
    try:
        with hostio.Downloader(device) as results:
            ops, travel = pieces(*bounds[0])
            nxt = up.submit([ops[k] for k in travel])
            for n, (i0, i1) in enumerate(bounds):
                tensors, ready = nxt.result()  # (re-raises what the worker raised)
                cur, cur_travel = ops, travel
                if n + 1 < len(bounds):
                    ops, travel = pieces(*bounds[n + 1])
                    nxt = up.submit([ops[k] for k in travel])
                main.wait_event(ready)
                for k, t in zip(cur_travel, tensors):
                    cur[k] = t
                res = evaluate(eos, func, cur[0], cur[1], cur[2], gravity=gravity)  # device tensor
                if out is None:
                    # (a mapping of our own when large, pooled across calls: hostio.result_array)
                    out = hostio.result_array(shape, np.float32 if res.dtype == torch.float32
                                              else np.float64)
                results.submit([(out[i0:i1], res.reshape(out[i0:i1].shape))])
    finally:
        up.close()
    return out


def _evaluate_promoted(eos, func, T, S, p, gravity, device, on_device, scalar_in):
    """The dtype combinations of numpy's promotion that the tuned kernel does not cover."""
    ops, shapes = [], []
    for x in (T, S, p):
        if x is None or _is_weak(x):
            ops.append(x)
            continue
        t = _as_tensor(x, device)
        if t.dtype not in (torch.float32, torch.float64):
            t = t.double()
        ops.append(t)
        shapes.append(tuple(t.shape))
    shape = torch.broadcast_shapes(*shapes)
    flat = []
    for t in ops:
        if isinstance(t, torch.Tensor):
            t = t.reshape(1) if t.numel() == 1 else t.expand(shape).contiguous().reshape(-1)
        flat.append(t)
    out = core.eos_map_promote(flat[0], flat[1], flat[2], eos=eos, func=func,
                               gravity=9.8 if gravity is None else gravity).reshape(shape)
    if on_device:
        return out
    res = hostio.to_host(out)
    if scalar_in:
        return res.dtype.type(res.reshape(()))
    return res


def evaluate(eos, func, T, S, p, gravity=None):
    """f(T, S, p) with numpy broadcasting; returns the kind of array it was given."""
    core.require_device()
    # a numpy masked array (a netCDF4 read) means NaN where it is masked -- what the reference's
    # functions are handed once xarray has wrapped it (derived.py:597-639 via apply_ufunc)
    T, S, p = (hostio.as_plain(x) if isinstance(x, np.ma.MaskedArray) else x for x in (T, S, p))
    if not any(isinstance(x, torch.Tensor) for x in (T, S, p)):
        shape = np.broadcast_shapes(*(_shape(x) for x in (T, S, p) if x is not None))
        if len(shape) >= 1 and int(np.prod(shape)) > _HOST_PIPELINE_ELEMS and shape[0] > 1:
            return _evaluate_host_chunked(eos, func, T, S, p, gravity)
        T, S, p = (hostio.as_plain(x[...]) if _is_lazy(x) else x for x in (T, S, p))
    on_device = any(isinstance(x, torch.Tensor) and x.is_cuda for x in (T, S, p))
    scalar_in = all(np.ndim(x) == 0 and not isinstance(x, torch.Tensor) for x in (T, S, p))
    device = next(
        (x.device for x in (T, S, p) if isinstance(x, torch.Tensor) and x.is_cuda),
        torch.device("cuda", torch.cuda.current_device()),
    )
    if func == "density_ref" or not _tuned_kernel_covers(_kind(T), _kind(S), _kind(p), eos):
        return _evaluate_promoted(eos, func, T, S, p, gravity, device, on_device, scalar_in)
    Tt, St = _as_tensor(T, device), _as_tensor(S, device)
    pt = None if p is None else _as_tensor(p, device).double()

    shape = torch.broadcast_shapes(Tt.shape, St.shape, pt.shape if pt is not None else ())
    Tb = Tt.expand(shape).contiguous()
    Sb = St.expand(shape).contiguous()

    def kernel(T4, S4, pp):
        if func == "inverse_barometer":  # dynamic.py:34-36, fused into the EOS kernel
            return core.inverse_barometer(T4, S4, pp, gravity=gravity, eos=eos)
        return core.eos_map(T4, S4, pp, eos=eos, func=func)

    out = None
    if pt is not None and len(shape) >= 3 and pt.numel() > 1:
        nz = shape[-3]
        pshape = tuple(pt.shape)
        while len(pshape) > 3 and pshape[0] == 1:
            pshape = pshape[1:]
        if pshape == (nz, 1, 1):
            # the calc_rho layout: (…, z, y, x) fields with a z-profile pressure (nz,1,1) or
            # (1,…,nz,1,1).  A bare (nz,) is NOT taken here: numpy aligns it with the LAST axis
            # (x), and so does the general path below (ambiguous only when nx == nz)
            lead = int(np.prod(shape[:-3])) if len(shape) > 3 else 1
            T4 = Tb.reshape((lead,) + tuple(shape[-3:]))
            S4 = Sb.reshape((lead,) + tuple(shape[-3:]))
            out = kernel(T4, S4, pt.reshape(nz)).reshape(shape)
    if out is None:
        n = int(np.prod(shape)) if len(shape) else 1
        T3, S3 = Tb.reshape(1, 1, n), Sb.reshape(1, 1, n)
        if pt is None or pt.numel() == 1:
            pp = pt
        else:
            pp = pt.expand(shape).contiguous().reshape(1, 1, n)
        out = kernel(T3, S3, pp).reshape(shape)

    if on_device:
        return out
    res = hostio.to_host(out)
    if scalar_in:
        return np.float64(res.reshape(()))
    return res
