"""numpy / torch / scalar front end of the pointwise EOS kernel (K0, mlx_eos_map).

The reference's EOS functions are numpy ufunc expressions with numpy broadcasting
(src/momlevel/eos/wright.py:23-165).  Here the same call signature routes to the
HIP kernel: host arrays are uploaded, evaluated on the MI355X and copied back;
device tensors stay on the device.  There is no host arithmetic fallback.
"""

import numpy as np
import torch

from .. import core


def _as_tensor(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device)
    a = np.asarray(x)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def evaluate(eos, func, T, S, p, gravity=None):
    """f(T, S, p) with numpy broadcasting; returns the kind of array it was given."""
    core.require_device()
    on_device = any(isinstance(x, torch.Tensor) and x.is_cuda for x in (T, S, p))
    scalar_in = all(np.ndim(x) == 0 and not isinstance(x, torch.Tensor) for x in (T, S, p))
    device = next(
        (x.device for x in (T, S, p) if isinstance(x, torch.Tensor) and x.is_cuda),
        torch.device("cuda", torch.cuda.current_device()),
    )
    Tt, St = _as_tensor(T, device), _as_tensor(S, device)
    if Tt.dtype != St.dtype:  # mixed precision inputs: evaluate in float64
        Tt, St = Tt.double(), St.double()
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
        if tuple(pt.shape) in ((nz,), (nz, 1, 1)) and pt.numel() == nz:
            # the calc_rho layout: (…, z, y, x) fields with a z-profile pressure
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
    res = out.cpu().numpy()
    if scalar_in:
        return np.float64(res.reshape(()))
    return res
