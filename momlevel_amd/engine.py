"""Device-side orchestration of the steric path on plain tensors.

This is the layer between the labelled front end (steric.py / reference.py /
derived.py, which mirror the reference's signatures) and the kernels (core.py).
It owns the time-chunk streaming of host-resident inputs: the reference hands
``steric()`` numpy-backed xarray objects of any size, so (time, z, y, x) fields
are moved to HBM a few time steps at a time (sized from free HBM), the kernels of
chunk k overlapping the upload of chunk k+1, while the time-invariant reference
state stays resident on the device.
"""

import warnings

import numpy as np
import torch

from . import core

_GIB = 1 << 30


def _is_device(x):
    return isinstance(x, torch.Tensor) and x.is_cuda


def device_of(*arrays):
    for a in arrays:
        if _is_device(a):
            return a.device
    core.require_device()
    return torch.device("cuda", torch.cuda.current_device())


def to_device(x, device, dtype=None):
    """Small or already-resident operand -> device tensor (no copy if resident)."""
    if isinstance(x, torch.Tensor):
        t = x.to(device)
    else:
        t = _host_tensor(x).to(device)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t


def _stream_dtype(x):
    """float32 fields stay float32 in HBM (half the bytes); everything else -> float64."""
    dt = x.dtype if isinstance(x, torch.Tensor) else np.asarray(x).dtype
    if str(dt) in ("torch.float32", "float32"):
        return torch.float32
    return torch.float64


def chunk_steps(nt, bytes_per_step, device, budget_bytes=None):
    """Time steps per chunk so that the double-buffered chunk fits the HBM budget."""
    if budget_bytes is None:
        free, _total = torch.cuda.mem_get_info(device)
        budget_bytes = min(free // 3, 24 * _GIB)
    return int(max(1, min(nt, budget_bytes // max(1, 2 * bytes_per_step))))


def _host_tensor(a, dtype=None):
    """numpy array (or view) -> CPU torch tensor sharing its memory when it can."""
    a = np.asarray(a)
    if dtype is not None and a.dtype != dtype:
        a = a.astype(dtype)
    if a.dtype.byteorder not in ("=", "|"):
        a = a.astype(a.dtype.newbyteorder("="))
    if not a.flags["C_CONTIGUOUS"]:
        a = np.ascontiguousarray(a)
    with warnings.catch_warnings():  # read-only views (e.g. broadcast volcello) are only read
        warnings.simplefilter("ignore", UserWarning)
        return torch.from_numpy(a)


class TimeChunks:
    """Iterate a (nt, nz, ny, nx) field pair in device-resident time chunks.

    Device-resident fields are sliced (no copy).  Host (numpy) fields are copied chunk by chunk
    straight from the caller's memory into a fresh device tensor: on the MI355X hosts a
    pageable hipMemcpy already runs at the PCIe Gen5 rate (56 GB/s measured, the same as from
    pinned memory), so a staging copy would only halve the rate.  The copy call blocks the
    host while the PREVIOUS chunk's kernels run asynchronously, so upload and compute overlap;
    compute is ~400x faster than the link anyway.  A (nz,ny,nx) operand (a held field) is
    uploaded once and yielded unchanged with every chunk.
    """

    def __init__(self, T, S, device, steps=None, extra_bytes_per_step=0):
        self.device = device
        self.fields = [T, S]
        self.nt = max(f.shape[0] for f in self.fields if f.ndim == 4) if any(
            f.ndim == 4 for f in self.fields
        ) else 1
        self.resident = [(f.ndim == 3) or _is_device(f) for f in self.fields]
        per_step = extra_bytes_per_step
        for f, res in zip(self.fields, self.resident):
            if f.ndim == 4 and not res:
                itemsize = 4 if _stream_dtype(f) == torch.float32 else 8
                per_step += int(np.prod(f.shape[1:])) * itemsize
        if steps is None:
            steps = chunk_steps(self.nt, per_step, device) if per_step else self.nt
        self.steps = max(1, min(int(steps), self.nt))
        self._held = [
            to_device(f, device, _stream_dtype(f)) if f.ndim == 3 else None for f in self.fields
        ]

    def _upload(self, f, t0, t1):
        dt = _stream_dtype(f)
        src = f[t0:t1]
        if isinstance(src, torch.Tensor):
            return src.to(device=self.device, dtype=dt)
        host = _host_tensor(src, np.float32 if dt == torch.float32 else np.float64)
        dev = torch.empty(host.shape, dtype=dt, device=self.device)
        dev.copy_(host)
        return dev

    def __iter__(self):
        for t0 in range(0, self.nt, self.steps):
            t1 = min(t0 + self.steps, self.nt)
            cur = []
            for f, res, held in zip(self.fields, self.resident, self._held):
                if f.ndim == 3:
                    cur.append(held)
                elif res:
                    cur.append(f[t0:t1])
                else:
                    cur.append(self._upload(f, t0, t1))
            yield t0, t1, cur[0], cur[1]


# ---------------------------------------------------------------------------------------
# reference state (src/momlevel/reference.py:48-85)
# ---------------------------------------------------------------------------------------
def reference_state(T0, S0, vol0, pres, eos="wright", f32_mode="faithful", with_masso=True):
    """rho0 (nz,ny,nx), volo, masso0 as device tensors; rhoga is masso0/volo (host).

    ``with_masso=False`` skips masso0 (returns None): a caller that is about to run K1 over a
    record whose first step IS this reference slab takes masso0 = masso(t=0) from that launch --
    the same kernel, tiling and operands, hence the same bits -- instead of a second pass.
    """
    dev = device_of(T0, S0, vol0)
    T0 = to_device(T0, dev, _stream_dtype(T0))
    S0 = to_device(S0, dev, _stream_dtype(S0))
    vol0 = to_device(vol0, dev, torch.float64)
    rho0 = core.eos_map(T0, S0, pres, eos=eos, f32_mode=f32_mode)
    volo = core.nansum(vol0)
    if not with_masso:
        return rho0, volo, None
    # masso0 through the SAME kernel/tiling as the time loop (a 1-step 4-D view), so that
    # masso(t=0) == masso0 bit for bit and steric[t=0] is exactly 0, as in the reference
    masso0 = core.steric_global_masso(
        T0.unsqueeze(0), S0.unsqueeze(0), vol0, pres, eos=eos, f32_mode=f32_mode
    )[0]
    return rho0, volo, masso0


# ---------------------------------------------------------------------------------------
# global variant (src/momlevel/steric.py:134-147)
# ---------------------------------------------------------------------------------------
def global_masso(T, S, vol0, pres, eos="wright", f32_mode="faithful", steps=None,
                 events=None, skip_dry=None):
    """masso(t) for every time step of (T, S) -> (nt,) float64 device tensor.

    Device-resident fields are processed by ONE K1 launch over all time steps; host
    fields in time chunks.  ``events``: see core.steric_global_masso (single launch only).
    """
    dev = device_of(T, S, vol0)
    vol0 = to_device(vol0, dev, torch.float64)
    pres = to_device(pres, dev, torch.float64)
    chunks = TimeChunks(T, S, dev, steps=steps)
    if chunks.steps >= chunks.nt:
        for _t0, _t1, Tc, Sc in chunks:
            return core.steric_global_masso(Tc, Sc, vol0, pres, eos=eos, f32_mode=f32_mode,
                                            events=events, skip_dry=skip_dry)
    out = torch.empty(chunks.nt, dtype=torch.float64, device=dev)
    for t0, t1, Tc, Sc in chunks:
        out[t0:t1] = core.steric_global_masso(Tc, Sc, vol0, pres, eos=eos, f32_mode=f32_mode,
                                              skip_dry=skip_dry)
    return out


def global_finalize(masso, volo, rhoga, area_sum):
    """steric.py:136-142 on host scalars / a (nt,) vector: (reference_height, eta(t))."""
    masso = np.asarray(masso, dtype=np.float64)
    expansion_coeff = np.log(rhoga / (masso / volo))
    reference_height = volo / area_sum
    return reference_height, reference_height * expansion_coeff, expansion_coeff


# ---------------------------------------------------------------------------------------
# local variant (src/momlevel/steric.py:150-166)
# ---------------------------------------------------------------------------------------
_PINNED_OUTPUT_LIMIT = 32 * _GIB


def _host_output(shape):
    """float64 host array for results copied back from the device.  Page-locked when it is not
    huge: the D2H copy then runs at the link rate and the pages are already resident (a fresh
    np.empty pays a page fault per 4 KiB on first touch).  The numpy view keeps the pinned
    tensor alive."""
    nbytes = int(np.prod(shape)) * 8
    if 0 < nbytes <= _PINNED_OUTPUT_LIMIT:
        try:
            return torch.empty(shape, dtype=torch.float64, pin_memory=True).numpy()
        except RuntimeError:
            pass
    return np.empty(shape, dtype=np.float64)

def local_steric(T, S, rho0, vol0, pres, rhozero, z_i=None, deptho=None, dz=None,
                 eos="wright", f32_mode="faithful", want_delta_rho=True, out_host=None,
                 steps=None):
    """delta_rho (nt,nz,ny,nx) [optional] and eta (nt,ny,nx).

    ``out_host=True`` returns numpy arrays filled chunk by chunk (for host inputs
    larger than HBM); otherwise device tensors.
    """
    dev = device_of(T, S, rho0, vol0)
    vol0 = to_device(vol0, dev, torch.float64)
    rho0 = to_device(rho0, dev, torch.float64)
    pres = to_device(pres, dev, torch.float64)
    rho0m = core.fold_mask(rho0, vol0)
    surface = vol0[0].contiguous()
    if dz is not None:
        dz = to_device(dz, dev, torch.float64)
    else:
        z_i = to_device(z_i, dev, torch.float64)
        deptho = to_device(deptho, dev, torch.float64)
    neg_inv = -1.0 / float(rhozero)
    nz, ny, nx = tuple(vol0.shape)
    n3 = nz * ny * nx
    if out_host is None:
        out_host = not (_is_device(T) or _is_device(S))
    extra = n3 * 8 if (want_delta_rho and out_host) else 0
    chunks = TimeChunks(T, S, dev, steps=steps, extra_bytes_per_step=extra)
    nt = chunks.nt
    if out_host:
        eta = _host_output((nt, ny, nx))
        drho = _host_output((nt, nz, ny, nx)) if want_delta_rho else None
    else:
        eta = torch.empty((nt, ny, nx), dtype=torch.float64, device=dev)
        drho = (
            torch.empty((nt, nz, ny, nx), dtype=torch.float64, device=dev)
            if want_delta_rho else None
        )
    for t0, t1, Tc, Sc in chunks:
        if out_host:
            d, e = core.steric_local(Tc, Sc, rho0m, surface, pres, neg_inv, dz=dz, z_i=z_i,
                                     deptho=deptho, eos=eos, f32_mode=f32_mode,
                                     want_delta_rho=want_delta_rho)
            # straight into the caller-visible arrays (one D2H pass, no intermediate copy)
            torch.from_numpy(eta[t0:t1]).copy_(e)
            if want_delta_rho:
                torch.from_numpy(drho[t0:t1]).copy_(d)
        else:
            core.steric_local(Tc, Sc, rho0m, surface, pres, neg_inv, dz=dz, z_i=z_i,
                              deptho=deptho, eos=eos, f32_mode=f32_mode,
                              want_delta_rho=want_delta_rho,
                              delta_rho_out=drho[t0:t1] if want_delta_rho else None,
                              eta_out=eta[t0:t1])
    return drho, eta
