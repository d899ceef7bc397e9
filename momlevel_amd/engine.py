"""Device-side orchestration of the steric path on plain tensors.

This is the layer between the labelled front end (steric.py / reference.py /
derived.py, which mirror the reference's signatures) and the kernels (core.py).
It owns the time-chunk streaming of host-resident inputs: the reference hands
``steric()`` numpy-backed xarray objects of any size, so (time, z, y, x) fields
are moved to HBM a few time steps at a time (sized from free HBM), the kernels of
chunk k overlapping the upload of chunk k+1, while the time-invariant reference
state stays resident on the device.  Every byte moves through hostio.py: staging and
result buffers of our own, never a GPU mapping of the caller's memory.
"""

import warnings

import numpy as np
import torch

from . import core, hostio
from .labeled import check_field_dtype, dtype_name

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
    return hostio.to_device(x, device, dtype)


def _stream_dtype(x):
    """float32 fields stay float32 in HBM (half the bytes); everything else -> float64."""
    dt = x.dtype if hasattr(x, "dtype") else np.asarray(x).dtype  # never reads a lazy array
    # (any byte order; float16 / long double fields are refused, not upcast: check_field_dtype)
    return torch.float32 if check_field_dtype(dt) == "float32" else torch.float64


def chunk_steps(nt, bytes_per_step, device, budget_bytes=None):
    """Time steps per chunk so that the double-buffered chunk fits the HBM budget."""
    if budget_bytes is None:
        free, _total = torch.cuda.mem_get_info(device)
        budget_bytes = min(free // 3, 4 * _GIB)  # small chunks: deeper H2D|compute|D2H overlap
    return int(max(1, min(nt, budget_bytes // max(1, 2 * bytes_per_step))))


def _host_tensor(a, dtype=None):
    """numpy array (or view) -> CPU torch tensor sharing its memory when it can."""
    a = hostio.as_plain(a)  # (a masked array -- a netCDF4 slice -- means NaN where masked)
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

    Device-resident fields are sliced (no copy).  Lazy fields (dask / netCDF4 / h5py / zarr
    arrays, anything sliceable that is not numpy) are READ chunk by chunk -- ``as_plain(f[t0:t1])``:
    masked elements of a netCDF4 slice become NaN, as in the xarray objects the reference sees --
    so the host never holds more than the chunks in flight.  Host (numpy) fields are copied
    chunk by chunk into a fresh device tensor through hostio's page-locked staging ring, on a
    copy stream: the host fills one staging buffer while the DMA engine drains another, chunk
    k+1 uploads while chunk k's kernels run and its results download on a third stream.  The
    caller's memory is only ever READ BY THE HOST -- rounds 1-2 page-locked it in place
    (hipHostRegister) and died twice of GPU page faults on heap addresses, DESIGN.md section 7.
    A (nz,ny,nx) operand (a held field) is uploaded once and yielded with every chunk.
    """

    def __init__(self, T, S, device, steps=None, extra_bytes_per_step=0, ramp=False,
                 first_step=None):
        """``ramp``: a SHORT first chunk (a quarter of the others, at least one step) when the fields
        are streamed from the host -- for the passes whose results go back to the host: the first
        result starts down the link a few milliseconds after the first upload instead of after a
        whole chunk's, and the download direction is the one that bounds such a call.
        ``first_step``: ``(T[0], S[0])`` as (z,y,x) DEVICE tensors (or None each) when the caller
        has them already -- steric() made its reference state from time level 0 of this very
        record: the first chunk is then that ONE step and is not uploaded a second time."""
        self.device = device
        self.ramp = bool(ramp)
        self._first_step = list(first_step) if first_step is not None else [None, None]
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
        # Uploads run on their own stream AND are staged by a worker thread: while the main thread
        # enqueues chunk k's kernels and result downloads, the worker copies chunk k+1 into the
        # staging ring piece by piece and enqueues its DMAs -- H2D(k+1), kernels(k) and D2H(k) all
        # overlap, and the main thread never blocks in a memcpy.
        self._copy_stream = None if all(self.resident) else torch.cuda.Stream(device=device)
        self._ring = None if all(self.resident) else hostio.new_ring()
        self._main = None  # the consumer's stream, captured when the iteration starts

    def _upload(self, f, t0, t1):
        """(worker thread) -> (device tensor, event that completes its upload)"""
        dt = _stream_dtype(f)
        src = f[t0:t1]
        if isinstance(src, torch.Tensor):
            if src.is_cuda:  # (resident fields never come here; another GPU's tensor might)
                return src.to(device=self.device, dtype=dt), None
            # a CPU torch tensor is host memory like any numpy array: same staging path, same
            # stream and event discipline (never the runtime's on-the-fly pinning of caller memory)
            src = src.detach().contiguous().numpy()
        # (a masked slice -- a netCDF4 read -- travels as data + mask: NaN is written under the mask
        #  while each piece is copied into the staging ring, hostio.split_masked)
        src, mask = hostio.split_masked(src, np.dtype(np.float32 if dt == torch.float32 else np.float64))
        host = _host_tensor(src)
        with torch.cuda.device(self.device):
            # the chunk belongs to the CONSUMER's stream (the caching allocator ties a block to the
            # stream that was current when it was allocated -- in this worker thread that would be
            # the default stream, whatever stream the consumer runs on)
            with torch.cuda.stream(self._main):
                dev = torch.empty(host.shape, dtype=dt, device=self.device)
            # `dev` may reuse memory the consumer's stream is done with (the allocator is
            # stream-ordered): everything enqueued there so far goes first
            self._copy_stream.wait_stream(self._main)
            with hostio.roctx_range(f"stage+H2D steps {t0}:{t1} ({host.numel() * host.element_size() >> 20} MiB)"):
                hostio.upload(host, dev, stream=self._copy_stream, ring=self._ring, mask=mask)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
            dev.record_stream(self._copy_stream)
        return dev, ev

    def _have_first_step(self):
        """is time level 0 of every field that would be uploaded on the device already?"""
        need = [i for i, (f, res) in enumerate(zip(self.fields, self.resident)) if not res]
        return bool(need) and all(
            _is_device(self._first_step[i]) and self._first_step[i].dtype == _stream_dtype(self.fields[i])
            and tuple(self._first_step[i].shape) == tuple(self.fields[i].shape[1:]) for i in need)

    def _stage(self, t0, t1):
        cur, events = [], []
        known = (t0, t1) == (0, 1) and self._have_first_step()
        for i, (f, res, held) in enumerate(zip(self.fields, self.resident, self._held)):
            if f.ndim == 3:
                cur.append(held)
            elif res:
                cur.append(f[t0:t1])
            elif known:
                cur.append(self._first_step[i].unsqueeze(0))  # (already there: nothing to move)
            else:
                dev, ev = self._upload(f, t0, t1)
                cur.append(dev)
                if ev is not None:
                    events.append(ev)
        return cur, events

    def bounds(self):
        """[(t0, t1)] of the chunks, in iteration order"""
        first = 0
        if self.ramp and not all(self.resident) and 1 < self.steps < self.nt:
            first = 1 if self._have_first_step() else max(1, self.steps // 4)
        out = [(0, first)] if first else []
        return out + [(t0, min(t0 + self.steps, self.nt)) for t0 in range(first, self.nt, self.steps)]

    def __iter__(self):
        bounds = self.bounds()
        if not bounds:
            return
        self._main = torch.cuda.current_stream(self.device)
        if all(self.resident):  # nothing to move: plain slicing
            for t0, t1 in bounds:
                cur, _ = self._stage(t0, t1)
                yield t0, t1, cur[0], cur[1]
            return
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(1, thread_name_prefix="mlx-upload") as worker:
            nxt = worker.submit(self._stage, *bounds[0])
            for i, (t0, t1) in enumerate(bounds):
                cur, events = nxt.result()  # (re-raises what the worker raised)
                # chunk i+1 is staged while the caller enqueues chunk i's kernels and downloads
                nxt = worker.submit(self._stage, *bounds[i + 1]) if i + 1 < len(bounds) else None
                for ev in events:
                    self._main.wait_event(ev)  # the consumer's stream waits for this chunk's DMAs
                yield t0, t1, cur[0], cur[1]


def time_dependent(pres):
    """True for a pressure that carries the time axis: (nt, nz|1, ny|1, nx|1) -- ``patm`` given as
    a DataArray with a time dimension (steric.py:58-60,96)."""
    return getattr(pres, "ndim", 0) == 4 and pres.shape[0] != 1  # (1,nz,1,1): a z profile


def pressure_chunk(pres, t0, t1, device):
    """The pressure operand of time steps [t0, t1): a time-dependent pressure is cut (and moved)
    chunk by chunk like theta/S, anything else is passed through."""
    if time_dependent(pres):
        return to_device(pres[t0:t1], device, torch.float64)
    return pres


def _pressure_bytes_per_step(pres):
    if time_dependent(pres) and not _is_device(pres):
        return int(np.prod(pres.shape[1:])) * 8
    return 0


# ---------------------------------------------------------------------------------------
# reference state (src/momlevel/reference.py:48-85)
# ---------------------------------------------------------------------------------------
def reference_state(T0, S0, vol0, pres, eos="wright", f32_mode="faithful", with_masso=True,
                    with_rho=True):
    """rho0 (nz,ny,nx), volo, masso0 as device tensors; rhoga is masso0/volo (host).

    ``with_masso=False`` skips masso0 (returns None): a caller that is about to run K1 over a
    record whose first step IS this reference slab takes masso0 = masso(t=0) from that launch --
    the same kernel, tiling and operands, hence the same bits -- instead of a second pass.
    ``with_rho=False`` skips the K0 density pass too (returns None for rho0): the global variants
    never read rho0, only ``setup_reference_state``'s public result carries it.
    """
    dev = device_of(T0, S0, vol0)
    T0 = to_device(T0, dev, _stream_dtype(T0))
    S0 = to_device(S0, dev, _stream_dtype(S0))
    vol0 = to_device(vol0, dev, torch.float64)
    rho0 = core.eos_map(T0, S0, pres, eos=eos, f32_mode=f32_mode) if with_rho else None
    volo = core.nansum(vol0)
    if not with_masso:
        return rho0, volo, None
    # masso0 through the SAME kernel/tiling as the time loop (a 1-step 4-D view), so that
    # masso(t=0) == masso0 bit for bit and steric[t=0] is exactly 0, as in the reference
    masso0 = core.steric_global_masso(
        T0.unsqueeze(0), S0.unsqueeze(0), vol0, pres, eos=eos, f32_mode=f32_mode
    )[0]
    return rho0, volo, masso0


def reference_state_time_dependent(T0, S0, vol0, pres4, eos="wright", f32_mode="faithful"):
    """The reference state momlevel builds when ``patm`` carries the time dimension
    (reference.py:54,71-77 broadcast by name): rho0 (nt,nz,ny,nx) = rho(theta0, S0, pres(t)),
    volo, masso (nt,).  theta0/S0 are broadcast over time by stride 0 -- never copied."""
    dev = device_of(T0, S0, vol0)
    T0 = to_device(T0, dev, _stream_dtype(T0))
    S0 = to_device(S0, dev, _stream_dtype(S0))
    vol0 = to_device(vol0, dev, torch.float64)
    pres4 = to_device(pres4, dev, torch.float64)
    nt = pres4.shape[0]
    shape = (nt,) + tuple(T0.shape)
    rho0 = core.eos_map(T0.unsqueeze(0).expand(shape), S0.unsqueeze(0).expand(shape), pres4,
                        eos=eos, f32_mode=f32_mode)
    volo = core.nansum(vol0)
    masso = core.masso(rho0.reshape(nt, -1), vol0.reshape(-1))
    return rho0, volo, masso


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
    if not time_dependent(pres):
        pres = to_device(pres, dev, torch.float64)
    chunks = TimeChunks(T, S, dev, steps=steps,
                        extra_bytes_per_step=_pressure_bytes_per_step(pres))
    if chunks.steps >= chunks.nt:
        for t0, t1, Tc, Sc in chunks:
            return core.steric_global_masso(Tc, Sc, vol0, pressure_chunk(pres, t0, t1, dev),
                                            eos=eos, f32_mode=f32_mode, events=events,
                                            skip_dry=skip_dry)
    out = torch.empty(chunks.nt, dtype=torch.float64, device=dev)
    for t0, t1, Tc, Sc in chunks:
        out[t0:t1] = core.steric_global_masso(Tc, Sc, vol0, pressure_chunk(pres, t0, t1, dev),
                                              eos=eos, f32_mode=f32_mode, skip_dry=skip_dry)
    return out


def global_finalize(masso, volo, rhoga, area_sum):
    """steric.py:136-142 on host scalars / a (nt,) vector: (reference_height, eta(t), expansion).
    ``volo`` and ``area_sum`` keep the dtype they come in (numpy scalars / 0-d arrays): with float32
    volcello and areacello -- what MOM6 writes -- numpy's ``volo / areacello.sum()`` is a float32
    division and the reference height a float32, which then meets the float64 expansion coefficient
    as a float64 factor; masso / volo and rhoga are float64 either way."""
    masso = np.asarray(masso, dtype=np.float64)
    volo, area_sum = np.asarray(volo)[()], np.asarray(area_sum)[()]
    expansion_coeff = np.log(rhoga / (masso / volo))
    reference_height = volo / area_sum
    return reference_height, reference_height * expansion_coeff, expansion_coeff


def sum_dtype(x):
    """numpy's result dtype of ``x.sum()`` for a float field: float32 stays float32 (xarray's skipna
    sum is np.sum(where(isnull, 0, x)) in the array's own dtype), anything else is float64 here."""
    dt = x.dtype if hasattr(x, "dtype") else np.asarray(x).dtype
    return np.float32 if dtype_name(dt) == "float32" else np.float64


# ---------------------------------------------------------------------------------------
# local variant (src/momlevel/steric.py:150-166)
# ---------------------------------------------------------------------------------------
def _host_output(shape):
    """float64 host array for results copied back from the device, filled through the staging ring:
    a huge-page mapping of our own when large (pooled across calls, hostio._ResultPool), an ordinary
    numpy array when small, page-locked memory when the caller opted in; see
    hostio.result_array."""
    return hostio.result_array(shape, np.float64)


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
    if not time_dependent(pres):
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
    extra = (n3 * 8 if (want_delta_rho and out_host) else 0) + _pressure_bytes_per_step(pres)
    chunks = TimeChunks(T, S, dev, steps=steps, extra_bytes_per_step=extra, ramp=out_host)
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
    # results go back on a stream and a worker thread of their own (hostio.Downloader): the D2H of
    # chunk k overlaps the H2D of chunk k+1 (PCIe is full duplex, the copies use different DMA
    # engines) and this loop goes straight on to chunk k+1's kernels
    if not out_host:
        for t0, t1, Tc, Sc in chunks:
            core.steric_local(Tc, Sc, rho0m, surface, pressure_chunk(pres, t0, t1, dev), neg_inv,
                              dz=dz, z_i=z_i, deptho=deptho, eos=eos, f32_mode=f32_mode,
                              want_delta_rho=want_delta_rho,
                              delta_rho_out=drho[t0:t1] if want_delta_rho else None,
                              eta_out=eta[t0:t1])
        return drho, eta
    with hostio.Downloader(dev) as results:
        for t0, t1, Tc, Sc in chunks:
            pc = pressure_chunk(pres, t0, t1, dev)
            d, e = core.steric_local(Tc, Sc, rho0m, surface, pc, neg_inv, dz=dz, z_i=z_i,
                                     deptho=deptho, eos=eos, f32_mode=f32_mode,
                                     want_delta_rho=want_delta_rho)
            # straight into the caller-visible arrays (one D2H pass, no intermediate array)
            results.submit([(eta[t0:t1], e)] + ([(drho[t0:t1], d)] if want_delta_rho else []))
    # (the host arrays are complete: the with-block waits for the last piece)
    return drho, eta


# ---------------------------------------------------------------------------------------
# several variants from ONE upload of theta/S (the PCIe-bound case: host inputs)
# ---------------------------------------------------------------------------------------
VARIANTS = ("steric", "thermosteric", "halosteric")


def _streamed_pair(variants, T, S, T0, S0):
    """What has to stream for this set of variants: a field only some variant varies."""
    need_T = any(v != "halosteric" for v in variants)
    need_S = any(v != "thermosteric" for v in variants)
    return (T if need_T else T0), (S if need_S else S0)


def _reference_matches(T, S, T0, S0):
    """The one-pass kernels take the reference fields in the streamed fields' dtypes.  A reference
    state of another precision (steric(dset, reference=...) with a reference written elsewhere)
    mixes dtypes per variant -- numpy promotes each sub-expression -- so those runs take one launch
    per variant, where every (theta, so) pair gets the kernel of its own dtype combination."""
    return _stream_dtype(T0) == _stream_dtype(T) and _stream_dtype(S0) == _stream_dtype(S)


def _variant_operands(variant, Tc, Sc, T0, S0):
    """(theta, S) of one variant for the current chunk (steric.py:115-125)."""
    return (T0 if variant == "halosteric" else Tc), (S0 if variant == "thermosteric" else Sc)


def global_masso_variants(T, S, T0, S0, vol0, pres, variants, eos="wright", f32_mode="faithful",
                          steps=None, skip_dry=None, with_heat=False):
    """masso(t) of every requested variant -> {variant: (nt,) device tensor}; with ``with_heat``
    also ``"heat"``: sum(theta*vol0) per step (the heat-content integrand, an extension).

    Two or more variants (or the heat row) are computed by ONE pass of the all-variants kernel
    (core.steric_global_decomp: theta/S read once, 16 B/cell instead of 16+8+8); a single variant
    by its own launch.  Either way every row is bit-identical to the single-variant call, and
    theta/S chunks of host inputs are uploaded once."""
    dev = device_of(T, S, vol0)
    vol0 = to_device(vol0, dev, torch.float64)
    if not time_dependent(pres):
        pres = to_device(pres, dev, torch.float64)
    T0 = to_device(T0, dev, _stream_dtype(T0))
    S0 = to_device(S0, dev, _stream_dtype(S0))
    same = _reference_matches(T, S, T0, S0)
    one_pass = (len(variants) >= 2 and same) or with_heat
    heat_only = one_pass and not same  # the heat row does not depend on the reference fields
    Ts, Ss = (T, S) if one_pass else _streamed_pair(variants, T, S, T0, S0)
    chunks = TimeChunks(Ts, Ss, dev, steps=steps,
                        extra_bytes_per_step=_pressure_bytes_per_step(pres))
    nt = T.shape[0]
    names = list(variants) + (["heat"] if with_heat else [])
    out = {v: torch.empty(nt, dtype=torch.float64, device=dev) for v in names}
    for t0, t1, Tc, Sc in chunks:
        pc = pressure_chunk(pres, t0, t1, dev)
        if one_pass:
            rows = core.steric_global_decomp(Tc, Sc, T0.to(Tc.dtype), S0.to(Sc.dtype), vol0, pc,
                                             eos=eos, f32_mode=f32_mode, skip_dry=skip_dry)
            for v in (["heat"] if heat_only else names):
                out[v][t0:t1] = rows[core.DECOMP_ROWS.index(v)]
            if not heat_only:
                continue
        for v in variants:
            Tv, Sv = _variant_operands(v, Tc, Sc, T0, S0)
            out[v][t0:t1] = core.steric_global_masso(Tv, Sv, vol0, pc, eos=eos,
                                                     f32_mode=f32_mode, skip_dry=skip_dry)
    return out


def local_steric_variants(T, S, T0, S0, rho0, vol0, pres, rhozero, variants, z_i=None,
                          deptho=None, dz=None, eos="wright", f32_mode="faithful",
                          want_delta_rho=True, out_host=None, steps=None, annual_weights=None,
                          reference_is_step0=False):
    """{variant: (delta_rho, eta)}; theta/S chunks are uploaded once and reused.

    ``annual_weights`` (nt,), nt a multiple of 12, whole years back to back: the days-in-month
    weighted annual means (util.annual_average) are taken ON THE DEVICE, chunk by chunk, so only
    1/12 of delta_rho / eta is ever stored or copied back; the outputs are (nt/12, ...).
    ``reference_is_step0``: (T0, S0) ARE time level 0 of (T, S) -- a reference state steric() made
    itself: the first chunk of a host record is then that step, taken from the device.
    """
    dev = device_of(T, S, rho0, vol0)
    vol0 = to_device(vol0, dev, torch.float64)
    rho0 = to_device(rho0, dev, torch.float64)
    if not time_dependent(pres):
        pres = to_device(pres, dev, torch.float64)
    T0 = to_device(T0, dev, _stream_dtype(T0))
    S0 = to_device(S0, dev, _stream_dtype(S0))
    rho0m = core.fold_mask(rho0, vol0)
    surface = vol0[0].contiguous()
    if dz is not None:
        dz = to_device(dz, dev, torch.float64)
    else:
        z_i = to_device(z_i, dev, torch.float64)
        deptho = to_device(deptho, dev, torch.float64)
    neg_inv = -1.0 / float(rhozero)
    nz, ny, nx = tuple(vol0.shape)
    nt = T.shape[0]
    if out_host is None:
        out_host = not (_is_device(T) or _is_device(S))
    n_out = len(variants)
    annual = annual_weights is not None
    extra = n_out * nz * ny * nx * 8 if (want_delta_rho and (out_host or annual)) else 0
    extra += _pressure_bytes_per_step(pres)
    Ts, Ss = _streamed_pair(variants, T, S, T0, S0)
    if annual:
        if nt % 12:
            raise ValueError("annual means need whole years (12 steps each)")
        w_dev = to_device(np.asarray(annual_weights, dtype=np.float64), dev, torch.float64)
        # whole years per chunk; device-resident inputs are chunked too (delta_rho of the full
        # record would not fit beside them)
        per_step = extra + 2 * nz * ny * nx * 8
        if steps is None:
            steps = chunk_steps(nt, per_step, dev)
        steps = max(12, (int(steps) // 12) * 12)
    # (annual means need whole years per chunk: no short first chunk there)
    chunks = TimeChunks(Ts, Ss, dev, steps=steps, extra_bytes_per_step=extra,
                        ramp=out_host and not annual,
                        first_step=(T0, S0) if reference_is_step0 else None)
    nt_out = nt // 12 if annual else nt

    def alloc(shape):
        return _host_output(shape) if out_host else torch.empty(shape, dtype=torch.float64,
                                                                device=dev)

    # all three variants of a 4-D record: ONE pass of the all-variants kernel per chunk (theta/S
    # read once: 16 B read + 3 x 8 B written per cell instead of 56 B); every field bit-identical
    # to its single-variant launch
    rows = core.LOCAL_DECOMP_ROWS
    one_pass = (set(variants) == set(rows) and len(variants) == 3 and T.ndim == 4 and S.ndim == 4
                and _reference_matches(T, S, T0, S0))
    direct = one_pass and not out_host and not annual  # the kernel writes the final tensors
    if direct:
        eta_all = torch.empty((3, nt, ny, nx), dtype=torch.float64, device=dev)
        drho_all = (torch.empty((3, nt, nz, ny, nx), dtype=torch.float64, device=dev)
                    if want_delta_rho else None)
        eta = {v: eta_all[i] for i, v in enumerate(rows)}
        drho = {v: (drho_all[i] if want_delta_rho else None) for i, v in enumerate(rows)}
    else:
        eta = {v: alloc((nt_out, ny, nx)) for v in variants}
        drho = {v: (alloc((nt_out, nz, ny, nx)) if want_delta_rho else None) for v in variants}
    kw = dict(dz=dz, z_i=z_i, deptho=deptho, eos=eos, f32_mode=f32_mode,
              want_delta_rho=want_delta_rho)
    import contextlib

    # host results leave on a stream and a worker thread of their own (hostio.Downloader)
    with (hostio.Downloader(dev) if out_host else contextlib.nullcontext()) as results:
        for t0, t1, Tc, Sc in chunks:
            o0, o1 = (t0 // 12, t1 // 12) if annual else (t0, t1)
            pc = pressure_chunk(pres, t0, t1, dev)
            if direct:
                core.steric_local_decomp(
                    Tc, Sc, T0, S0, rho0m, surface, pc, neg_inv,
                    delta_rho_out=drho_all[:, t0:t1] if want_delta_rho else None,
                    eta_out=eta_all[:, t0:t1], **kw)
                continue
            if one_pass:
                d3, e3 = core.steric_local_decomp(Tc, Sc, T0, S0, rho0m, surface, pc, neg_inv, **kw)
                chunk_fields = {v: (d3[i] if want_delta_rho else None, e3[i])
                                for i, v in enumerate(rows)}
            going_out = []
            for v in variants:
                Tv, Sv = _variant_operands(v, Tc, Sc, T0, S0)
                if annual:  # K2 on the chunk, then the fused annual-mean epilogue on the device
                    d, e = (chunk_fields[v] if one_pass else
                            core.steric_local(Tv, Sv, rho0m, surface, pc, neg_inv, **kw))
                    e = core.group_weighted_mean(e, w_dev[t0:t1], 12,
                                                 out=None if out_host else eta[v][o0:o1])
                    if want_delta_rho:
                        d = core.group_weighted_mean(d, w_dev[t0:t1], 12,
                                                     out=None if out_host else drho[v][o0:o1])
                    if not out_host:
                        continue
                if out_host:
                    if not annual:
                        d, e = (chunk_fields[v] if one_pass else
                                core.steric_local(Tv, Sv, rho0m, surface, pc, neg_inv, **kw))
                    going_out.append((eta[v][o0:o1], e))
                    if want_delta_rho:
                        going_out.append((drho[v][o0:o1], d))
                else:
                    core.steric_local(Tv, Sv, rho0m, surface, pc, neg_inv,
                                      delta_rho_out=drho[v][t0:t1] if want_delta_rho else None,
                                      eta_out=eta[v][t0:t1], **kw)
            if going_out:
                results.submit(going_out)
    return {v: (drho[v], eta[v]) for v in variants}
