"""Host <-> HBM transfers of the steric path: everything the DMA engines touch is memory WE own.

momlevel hands ``steric()`` numpy-backed arrays: caller-owned, pageable heap or mmap memory.
Rounds 1-2 page-locked those arrays in place (hipHostRegister) and let the runtime pin pageable
sources and destinations on the fly for everything else.  Both put GPU mappings on memory whose
lifetime belongs to somebody else -- glibc trims and regrows the brk heap, numpy frees and reuses
blocks, mappings come and go -- and both of this project's unexplained aborts were GPU page faults
on exactly such addresses ("Memory access fault by GPU node-N on address <a heap page>", DESIGN.md
section 7).  So this module never shows the GPU a byte of foreign memory:

* uploads go through a small ring of page-locked staging buffers (torch's pinned allocator:
  hipHostMalloc'ed once, reused for the life of the process): the host copies piece k+1 into one
  buffer with all its cores while the DMA engine drains piece k from another;
* downloads come back through the same ring into ordinary numpy arrays (8 memcpy threads; 47 GB/s
  end to end, first touch of the fresh pages included) or, opt-in, land in page-locked arrays of our
  own handed to the caller as numpy arrays (one DMA at 57 GB/s -- after ~1 s per 16 GB to allocate
  them: profiles/r04_d2h_probe.log).

Nothing here computes: allocation, copies, stream ordering (torch as the device-array container).
"""

import os
import threading

import numpy as np
import torch

_MIB = 1 << 20
PIECE_BYTES = int(os.environ.get("MOMLEVEL_AMD_STAGING_PIECE_MIB", "64")) * _MIB
RING_DEPTH = 3
# below this size a transfer is latency-bound and travels through the runtime's own small staging
# buffers (it never pins caller memory for less than a MiB)
SMALL_BYTES = 256 << 10
# the last 64 KiB of a staging buffer are never handed to the DMA engine.  A PRECAUTION, not the
# remedy of an established cause: were a copy path ever to touch a little more than the bytes it was
# asked to move, it would find mapped memory of ours there (the mechanism of the
# two GPU page faults this project has seen was never determined)
_SLACK = 64 << 10

_rings = {}
_rings_lock = threading.Lock()

_range_log = None


def _log_range(kind, ptr, nbytes):
    """MOMLEVEL_AMD_TRANSFER_LOG=<file>: append the address range of every staging / result buffer
    this module allocates and of every host array it is handed, line-buffered -- so that, should a
    run abort with a GPU memory access fault, the faulting address can be matched to a transfer
    (what round 3's log could not do)."""
    global _range_log
    path = os.environ.get("MOMLEVEL_AMD_TRANSFER_LOG")
    if not path:
        return
    if _range_log is None or _range_log.name != path:
        _range_log = open(path, "a", buffering=1)
    _range_log.write(f"{kind} [{ptr:#x}, {ptr + nbytes:#x}) {nbytes} bytes\n")


class _Ring:
    """RING_DEPTH page-locked buffers of PIECE_BYTES (+ slack) and the event that marks each
    buffer's last transfer as finished."""

    def __init__(self):
        self.bufs = [None] * RING_DEPTH
        self.events = [None] * RING_DEPTH
        self.next = 0

    def acquire(self):
        i = self.next
        self.next = (i + 1) % RING_DEPTH
        if self.events[i] is not None:
            self.events[i].synchronize()  # the DMA that last used this buffer is done
            self.events[i] = None
        if self.bufs[i] is None or self.bufs[i].numel() < PIECE_BYTES + _SLACK:
            self.bufs[i] = torch.empty(PIECE_BYTES + _SLACK, dtype=torch.uint8, pin_memory=True)
            _log_range(f"staging ring buffer {i} (page-locked, ours)", self.bufs[i].data_ptr(),
                       self.bufs[i].numel())
        return i, self.bufs[i]


def _ring(device):
    """The calling THREAD's staging ring for ``device``: a ring's cursor and buffer events are not
    shareable, and to_device / to_host may be called from several user threads at once."""
    key = (threading.get_ident(),
           torch.device(device).index if torch.device(device).index is not None else -1)
    with _rings_lock:
        if key not in _rings:
            alive = {t.ident for t in threading.enumerate()}
            for dead in [k for k in _rings if k[0] not in alive]:
                del _rings[dead]  # (its page-locked buffers go back to torch's pinned allocator)
            _rings[key] = _Ring()
        return _rings[key]


def _bytes_view(t):
    return t.view(torch.uint8).reshape(-1)


_threads = None


def host_threads():
    """Threads for the host side of a staged copy: the CPUs this process may actually use -- its
    affinity mask AND its cgroup CPU quota (a container that sees 128 cores but is allowed 16 must
    not start 128 copy threads) -- capped at 8; MOMLEVEL_AMD_COPY_THREADS overrides.  Evaluated
    once per process."""
    global _threads
    if _threads is None:
        _threads = _count_host_threads()
    return _threads


def _count_host_threads():
    env = os.environ.get("MOMLEVEL_AMD_COPY_THREADS")
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 8))  # 8 threads fill a staging buffer at ~100 GB/s (profiles/r03_host_copy_probe.log)


_pool = None


def _copy_pool():
    global _pool
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor

        _pool = ThreadPoolExecutor(max_workers=host_threads(), thread_name_prefix="mlx-stage")
    return _pool


_memcpy = None


def _libc_memcpy():
    """libc's memcpy through ctypes: a foreign call, so it runs WITHOUT the GIL -- np.copyto and
    torch's sliced copy_ measured 11 and 6 GB/s from eight Python threads (they serialise), this
    35 GB/s on the same eight cores."""
    global _memcpy
    if _memcpy is None:
        import ctypes

        fn = ctypes.CDLL(None).memcpy
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        fn.restype = ctypes.c_void_p
        _memcpy = fn
    return _memcpy


def _host_copy(dst, src):
    """dst[:] = src for two equally long contiguous uint8 CPU tensors, split over the copy
    threads.  (torch's own CPU copy_ is not used: it sizes its thread team from the visible cores,
    not from the cgroup quota -- 5.7 GB/s end to end on the quota-limited GPU boxes.)"""
    n = dst.numel()
    assert src.numel() == n
    memcpy = _libc_memcpy()
    d, s_ = dst.data_ptr(), src.data_ptr()
    threads = host_threads()
    if n < (4 << 20) or threads == 1:
        memcpy(d, s_, n)
        return
    step = -(-n // threads)
    step = -(-step // 4096) * 4096
    futures = [_copy_pool().submit(memcpy, d + o, s_ + o, min(step, n - o))
               for o in range(0, n, step)]
    for f in futures:
        f.result()


class roctx_range:
    """``with roctx_range("H2D chunk 3"):`` -- a roctx range (torch's nvtx binding) when
    MOMLEVEL_AMD_ROCTX=1, nothing otherwise; read by scripts/ingest_profile.py's rocprofv3 run."""

    def __init__(self, name):
        self.name = name if os.environ.get("MOMLEVEL_AMD_ROCTX") == "1" else None

    def __enter__(self):
        if self.name is not None:
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *exc):
        if self.name is not None:
            torch.cuda.nvtx.range_pop()
        return False


def new_ring():
    """A private staging ring (engine.TimeChunks stages its uploads from a worker thread and must
    not share buffers with transfers issued by the main thread)."""
    return _Ring()


def upload(host, dev, stream=None, ring=None):
    """Copy the contiguous CPU tensor ``host`` into the device tensor ``dev`` (same dtype and
    number of elements) through the staging ring, asynchronously on ``stream`` (default: the
    device's current stream).  Returns when the last piece has been ENQUEUED; ``host`` may be
    modified or freed from then on (its bytes are in the staging buffers), ``dev`` is complete
    once ``stream`` has passed the copies."""
    assert host.dtype == dev.dtype and host.numel() == dev.numel() and host.is_contiguous()
    device = dev.device
    stream = stream if stream is not None else torch.cuda.current_stream(device)
    nbytes = host.numel() * host.element_size()
    if nbytes == 0:
        return
    if nbytes < SMALL_BYTES:
        with torch.cuda.stream(stream):
            dev.copy_(host.reshape(dev.shape))  # staged by the runtime itself
        return
    hb, db = _bytes_view(host), _bytes_view(dev)
    ring = ring if ring is not None else _ring(device)
    _log_range("upload source (caller's memory: read by host memcpy only)", host.data_ptr(), nbytes)
    step = PIECE_BYTES // 8 * 8
    for off in range(0, nbytes, step):
        n = min(step, nbytes - off)
        i, buf = ring.acquire()
        _host_copy(buf[:n], hb[off:off + n])  # caller's bytes -> our staging buffer
        with torch.cuda.stream(stream):
            db[off:off + n].copy_(buf[:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        ring.events[i] = ev


def to_device(x, device, dtype=None):
    """numpy array / CPU tensor -> new device tensor (through the staging ring when large)."""
    if isinstance(x, torch.Tensor):
        if x.is_cuda:
            t = x.to(device)
            return t if dtype is None or t.dtype == dtype else t.to(dtype)
        host = x.contiguous()
    else:
        a = np.asarray(x)
        if a.dtype.byteorder not in ("=", "|"):
            a = a.astype(a.dtype.newbyteorder("="))
        if not a.flags["C_CONTIGUOUS"]:
            a = np.ascontiguousarray(a)
        import warnings

        with warnings.catch_warnings():  # read-only views (broadcasts, memmaps) are only read
            warnings.simplefilter("ignore", UserWarning)
            host = torch.from_numpy(a)
    if host.dtype not in (torch.float32, torch.float64):
        host = host.to(torch.float64)
    dev = torch.empty(host.shape, dtype=host.dtype, device=device)
    upload(host, dev)
    return dev if dtype is None or dev.dtype == dtype else dev.to(dtype)


# Results up to this size are returned in page-locked arrays of our own (one asynchronous DMA each);
# larger ones in ordinary numpy arrays filled through the staging ring.  Default 0 = always the ring
# (round 4): allocating page-locked memory costs ~1 s per 16 GB on the MI355X hosts, so a fresh
# page-locked result array is 4x slower END TO END than the ring (alloc 15 GB/s + DMA 57 GB/s
# against DMA + 8 memcpy threads = 47 GB/s, first touch of the pageable pages included --
# profiles/r04_d2h_probe.log); it only pays for callers that free their results between calls, so
# that torch's pinned allocator can hand the same block out again: MOMLEVEL_AMD_PINNED_RESULT_MIB.
PINNED_RESULT_LIMIT = int(os.environ.get("MOMLEVEL_AMD_PINNED_RESULT_MIB", "0")) << 20


def pinned_array(shape, dtype=np.float64):
    """The host array a result is downloaded into: a page-locked numpy array of our own (it keeps
    its pinned tensor alive) up to PINNED_RESULT_LIMIT bytes, else -- the default -- an ordinary
    numpy array, which download_into() fills through the staging ring."""
    tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32}[np.dtype(dtype)]
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if 0 < nbytes <= PINNED_RESULT_LIMIT:
        try:
            t = torch.empty(tuple(shape), dtype=tdt, pin_memory=True)
            _log_range("result array (page-locked, ours)", t.data_ptr(), nbytes)
            return t.numpy()
        except RuntimeError:
            pass
    return np.empty(shape, dtype=dtype)


def download_into(out, dev, stream=None):
    """Copy the device tensor ``dev`` into the numpy array ``out`` (same shape / dtype, contiguous).
    Page-locked ``out`` (pinned_array): one asynchronous DMA on ``stream`` -- the caller
    synchronises the stream before reading.  Pageable ``out``: piecewise through the ring,
    complete on return."""
    device = dev.device
    stream = stream if stream is not None else torch.cuda.current_stream(device)
    host = torch.from_numpy(out)
    if host.numel() == 0:
        return
    nbytes = host.numel() * host.element_size()
    if host.is_pinned() or nbytes < SMALL_BYTES:
        with torch.cuda.stream(stream):
            host.copy_(dev.reshape(host.shape), non_blocking=host.is_pinned())
            dev.record_stream(stream)
        return
    # (a non-contiguous `dev` is packed ON `stream`: the caller ordered `stream` behind the kernels
    # that produced `dev`, and the piecewise copies below must not overtake the packing kernel)
    with torch.cuda.stream(stream):
        src = dev.contiguous()
    hb, db = _bytes_view(host), _bytes_view(src)
    ring = _ring(device)
    _log_range("download destination (pageable: written by host memcpy only)", host.data_ptr(),
               nbytes)
    step = PIECE_BYTES // 8 * 8
    pending = []  # (offset, n, buffer index): DMA enqueued, host copy-out still to do

    def drain(k):
        off, n, i = pending.pop(k)
        ring.events[i].synchronize()
        ring.events[i] = None
        _host_copy(hb[off:off + n], ring.bufs[i][:n])

    for off in range(0, nbytes, step):
        n = min(step, nbytes - off)
        if len(pending) == RING_DEPTH - 1:
            drain(0)
        i, buf = ring.acquire()
        with torch.cuda.stream(stream):
            buf[:n].copy_(db[off:off + n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        ring.events[i] = ev
        pending.append((off, n, i))
    while pending:
        drain(0)
    src.record_stream(stream)


def to_host(t):
    """Device tensor -> numpy array (through the staging ring when not small; synchronises)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        return t.detach().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    t = t.detach()
    if t.dtype not in (torch.float32, torch.float64) or t.numel() * t.element_size() < SMALL_BYTES:
        return t.cpu().numpy()
    out = pinned_array(tuple(t.shape), np.float32 if t.dtype == torch.float32 else np.float64)
    stream = torch.cuda.current_stream(t.device)
    download_into(out, t, stream)
    stream.synchronize()
    return out
