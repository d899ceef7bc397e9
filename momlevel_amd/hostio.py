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
* downloads come back through such a ring into ordinary numpy arrays -- the bulk results of the
  local variants on a worker thread of their own (Downloader: one pipeline of pieces across arrays
  and time chunks, 52 GB/s of the link's 57 on the reference's recorded call) -- or, opt-in, land in
  page-locked arrays of our own handed to the caller as numpy arrays (one DMA at 57 GB/s -- after
  ~1 s per 16 GB to allocate them: profiles/r04_d2h_probe.log);
* the host side of every piece is ONE call into our library (mlx_host_copy: a team of native
  threads, streaming stores, the GIL released throughout).

Nothing here computes: allocation, copies, stream ordering (torch as the device-array container).
"""

import os
import threading

import numpy as np
import torch

from .labeled import MaskedSource, as_plain  # numpy masked arrays MEAN NaN (what xarray hands the reference)

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
_range_log_lock = threading.Lock()  # the caller's thread, the upload and the download workers all log


def _log_range(kind, ptr, nbytes):
    """MOMLEVEL_AMD_TRANSFER_LOG=<file>: append the address range of every staging / result buffer
    this module allocates and of every host array it is handed, line-buffered -- so that, should a
    run abort with a GPU memory access fault, the faulting address can be matched to a transfer
    (what round 3's log could not do)."""
    global _range_log
    path = os.environ.get("MOMLEVEL_AMD_TRANSFER_LOG")
    if not path:
        return
    with _range_log_lock:  # one handle, whole lines: this log is the evidence if a fault recurs
        if _range_log is None or _range_log.name != path:
            if _range_log is not None:
                _range_log.close()
            _range_log = open(path, "a", buffering=1)
        _range_log.write(f"{kind} [{ptr:#x}, {ptr + nbytes:#x}) {nbytes} bytes\n")


class _Ring:
    """RING_DEPTH page-locked buffers of PIECE_BYTES (+ slack) and the event that marks each
    buffer's last transfer as finished."""

    def __init__(self, depth=RING_DEPTH):
        self.depth = depth
        self.bufs = [None] * depth
        self.events = [None] * depth
        self.next = 0

    def acquire(self):
        i = self.next
        self.next = (i + 1) % self.depth
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
    # 8 threads copy a staging piece at 75-120 GB/s, twice the link's rate; 12 and 16 measured no
    # faster end to end (profiles/r04_hostio_breakdown.log)
    return max(1, min(n, 8))


_copy_fn = None


def _native_copy():
    """mlx_host_copy of our own library: ONE foreign call per staging piece -- the GIL is released
    for its whole duration -- split over host_threads() native threads inside the library.  (The
    split used to be a Python thread pool calling libc's memcpy: 8-16 futures per piece, each
    completion taking the GIL the upload thread, the download thread and the caller contend for;
    np.copyto and torch's sliced copy_ serialise on it outright, 11 and 6 GB/s from eight threads.)
    It writes the destination with streaming stores; MOMLEVEL_AMD_HOST_COPY=libc keeps memcpy's
    cached stores (for A/B measurements)."""
    global _copy_fn
    if _copy_fn is None:
        from . import _lib

        fn = _lib.load().mlx_host_copy
        streaming = 0 if os.environ.get("MOMLEVEL_AMD_HOST_COPY", "stream") == "libc" else 1
        threads = host_threads()

        def copy(d, s, n):
            rc = fn(d, s, n, threads, streaming)
            if rc != 0:
                raise RuntimeError(f"mlx_host_copy failed ({rc}): {_lib.last_error()}")

        _copy_fn = copy
    return _copy_fn


def _host_copy(dst, src):
    """dst[:] = src for two equally long contiguous uint8 CPU tensors, split over the copy
    threads.  (torch's own CPU copy_ is not used: it sizes its thread team from the visible cores,
    not from the cgroup quota -- 5.7 GB/s end to end on the quota-limited GPU boxes.)"""
    n = dst.numel()
    assert src.numel() == n
    _native_copy()(dst.data_ptr(), src.data_ptr(), n)


def _host_copy_masked(dst, src, mask, elem):
    """dst[:] = where(mask, NaN, src) for two equally long contiguous uint8 CPU tensors seen as
    elements of ``elem`` bytes (float32 / float64 bit patterns) and a contiguous numpy bool array with
    one entry per element: the NaN fill of a masked array done ON THE WAY into the staging buffer --
    one pass over the bytes, no intermediate array (mlx_host_copy_masked, the same thread team)."""
    from . import _lib

    n = dst.numel()
    assert src.numel() == n and n % elem == 0 and mask.size == n // elem
    rc = _lib.load().mlx_host_copy_masked(dst.data_ptr(), src.data_ptr(), mask.ctypes.data,
                                          n // elem, elem, host_threads())
    if rc != 0:
        raise RuntimeError(f"mlx_host_copy_masked failed ({rc}): {_lib.last_error()}")


def split_masked(a, dtype=None):
    """-> (plain array, mask or None) for the upload of host data ``a``.  A floating numpy masked
    array that already has the wanted dtype and a C-contiguous native layout travels as its raw data
    plus its boolean mask -- upload() writes NaN under the mask while it copies each piece into the
    staging ring -- instead of being NaN-filled into an array of its own first (as_plain: a second
    pass over the bytes and a fresh allocation per time chunk, 2.3x the wall time of the plain call
    on the reference's recorded example).  Everything else: (as_plain(a) [cast to dtype], None)."""
    if isinstance(a, np.ma.MaskedArray):
        data, mask = np.ma.getdata(a), np.ma.getmask(a)
        if (mask is not np.ma.nomask and data.dtype.kind == "f" and data.dtype.itemsize in (4, 8)
                and (dtype is None or data.dtype == dtype) and data.dtype.isnative
                and data.flags["C_CONTIGUOUS"] and data.flags["ALIGNED"]
                and data.nbytes >= SMALL_BYTES):
            # (no `mask.any()` first: netCDF4 hands out full-size masks with nothing set all the time,
            #  and scanning one -- 10 ms per 160 MB, on the upload worker -- costs more than letting
            #  the copy read it: its blocks with no byte set are plain streaming copies)
            return data, np.ascontiguousarray(np.broadcast_to(mask, data.shape), dtype=np.bool_)
    a = as_plain(a)
    if dtype is not None and a.dtype != dtype:
        a = a.astype(dtype)
    return a, None


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


def new_ring(depth=RING_DEPTH):
    """A private staging ring (engine.TimeChunks stages its uploads from a worker thread and must
    not share buffers with transfers issued by the main thread)."""
    return _Ring(depth)


def upload(host, dev, stream=None, ring=None, mask=None):
    """Copy the contiguous CPU tensor ``host`` into the device tensor ``dev`` (same dtype and
    number of elements) through the staging ring, asynchronously on ``stream`` (default: the
    device's current stream).  Returns when the last piece has been ENQUEUED; ``host`` may be
    modified or freed from then on (its bytes are in the staging buffers), ``dev`` is complete
    once ``stream`` has passed the copies.  ``mask`` (split_masked): a contiguous numpy bool array,
    one entry per element -- NaN is written where it is set, in the same pass."""
    assert host.dtype == dev.dtype and host.numel() == dev.numel() and host.is_contiguous()
    device = dev.device
    stream = stream if stream is not None else torch.cuda.current_stream(device)
    nbytes = host.numel() * host.element_size()
    if nbytes == 0:
        return
    if nbytes < SMALL_BYTES:
        if mask is not None:
            host = torch.from_numpy(np.where(mask.reshape(tuple(host.shape)), np.nan,
                                             host.numpy()).astype(host.numpy().dtype))
        with torch.cuda.stream(stream):
            dev.copy_(host.reshape(dev.shape))  # staged by the runtime itself
        return
    hb, db = _bytes_view(host), _bytes_view(dev)
    ring = ring if ring is not None else _ring(device)
    _log_range("upload source (caller's memory: read by host memcpy only)", host.data_ptr(), nbytes)
    step = PIECE_BYTES // 8 * 8
    esz = host.element_size()
    mask_flat = None if mask is None else mask.reshape(-1)
    for off in range(0, nbytes, step):
        n = min(step, nbytes - off)
        i, buf = ring.acquire()
        if mask is None:
            _host_copy(buf[:n], hb[off:off + n])  # caller's bytes -> our staging buffer
        else:  # ... with NaN where the caller's masked array is masked
            _host_copy_masked(buf[:n], hb[off:off + n], mask_flat[off // esz:(off + n) // esz], esz)
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
        mask = None
    else:
        # (a masked array -- or one held as data + mask, labeled.MaskedSource: the (z,y,x) slabs of
        #  a reference state cut out of a masked field -- is NaN-filled by the staging copy, not
        #  into an array of its own first)
        a, mask = split_masked(x.array if isinstance(x, MaskedSource) else x)
        if a.dtype.byteorder not in ("=", "|"):
            a = a.astype(a.dtype.newbyteorder("="))
        if not a.flags["C_CONTIGUOUS"]:
            a = np.ascontiguousarray(a)
        import warnings

        with warnings.catch_warnings():  # read-only views (broadcasts, memmaps) are only read
            warnings.simplefilter("ignore", UserWarning)
            host = torch.from_numpy(a)
    if host.dtype not in (torch.float32, torch.float64):
        host = host.to(torch.float64)  # (never with a mask: split_masked keeps floating data only)
    dev = torch.empty(host.shape, dtype=host.dtype, device=device)
    upload(host, dev, mask=mask)
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


# Results of at least this size (the 4-D delta_rho fields and the (time, yh, xh) heights of the local
# variants at real sizes) are handed out in 2 MiB-aligned anonymous mappings of our own, advised
# MADV_HUGEPAGE, which a bounded pool keeps for the next call when their arrays die (_ResultPool).
# 0 disables (np.empty).
HUGE_RESULT_BYTES = int(os.environ.get("MOMLEVEL_AMD_HUGE_RESULT_MIB", "64")) << 20
_HUGE_PAGE = 2 << 20


def _pool_cap_bytes():
    """Bytes of freed result mappings kept for re-use: MOMLEVEL_AMD_RESULT_POOL_GIB, default a quarter
    of the memory this process may use (cgroup limit or MemTotal), 0 disables."""
    env = os.environ.get("MOMLEVEL_AMD_RESULT_POOL_GIB")
    if env is not None:
        return int(float(env) * (1 << 30))
    total = None
    try:
        with open("/sys/fs/cgroup/memory.max") as f:
            v = f.read().strip()
        if v != "max":
            total = int(v)
    except (OSError, ValueError):
        pass
    if total is None:
        try:
            with open("/proc/meminfo") as f:
                total = int(f.readline().split()[1]) << 10
        except (OSError, ValueError, IndexError):
            total = 0
    return total // 4


class _ResultPool:
    """Result mappings whose arrays have died, kept MAPPED for the next call's results instead of
    going back to the OS at once.  Why: a fresh anonymous page costs the kernel a zeroing pass before
    the copy-out writes it -- two passes over the bytes where the host cores' path to DRAM is the
    bottleneck (47-57 GB/s into fresh pages, 105-190 GB/s into pages that exist; faulting the pages
    in AHEAD of the copy from other threads measured slower still, profiles/r06_result_prefault_
    negative.log) -- while a caller that walks a long record calls steric() again and again, freeing
    each result after writing it out.  The kept pages are advised MADV_FREE: they count as this
    process's until the kernel wants memory, which then takes them without swapping (a later re-use
    finds zero pages there, as in a fresh mapping); so the pool cannot cause an out-of-memory kill
    (MOMLEVEL_AMD_RESULT_POOL_RECLAIMABLE=0 keeps them outright).  Bounded by bytes
    (_pool_cap_bytes: a quarter of the memory the process may use, MOMLEVEL_AMD_RESULT_POOL_GIB) and
    count; hostio.trim_result_pool() empties it."""

    MAX_MAPPINGS = 12

    def __init__(self):
        import collections

        self.lock = threading.Lock()
        self.free = []  # [(capacity in bytes, mmap object, aligned offset)], oldest first
        # mappings handed back by finalizers, not yet in `free`: a finalizer may run wherever the
        # garbage collector does -- also inside take() on this very thread -- so it never WAITS for
        # the lock: it appends here (atomic) and whoever holds or next takes the lock absorbs
        self.returned = collections.deque()
        self.reused = 0
        self.mapped = 0

    def _absorb(self):
        """(lock held) returned mappings -> free list, then the bounds"""
        while self.returned:
            self.free.append(self.returned.popleft())
        limit = _pool_cap_bytes()
        while self.free and (sum(c for c, _m, _o in self.free) > limit
                             or len(self.free) > self.MAX_MAPPINGS):
            self.free.pop(0)  # (unmapped when its mmap object is collected: now)

    def take(self, nbytes):
        with self.lock:
            self._absorb()
            best = None
            for i, (cap, _m, _off) in enumerate(self.free):
                if nbytes <= cap <= nbytes + max(nbytes // 2, _HUGE_PAGE):
                    if best is None or cap < self.free[best][0]:
                        best = i
            if best is None:
                return None
            self.reused += 1
            return self.free.pop(best)

    def give(self, cap, m, off):
        """(a weakref finalizer: any thread, any moment) keep the mapping of a dead result"""
        import mmap

        if cap > _pool_cap_bytes():
            return  # (not kept: the mapping goes when `m` does)
        if os.environ.get("MOMLEVEL_AMD_RESULT_POOL_RECLAIMABLE", "1") != "0":
            # the kernel may take the kept pages back whenever it wants memory (MADV_FREE): they
            # cannot cause an out-of-memory kill; a later re-use finds zero pages where it did.
            # (=0 keeps them outright.  Suspected for a while of the rare 2.13 s calls of
            # profiles/r06_tail_latency.log -- those occur with the pages kept as well.)
            try:
                m.madvise(getattr(mmap, "MADV_FREE", 8), off, cap)
            except (OSError, ValueError):
                pass
        self.returned.append((cap, m, off))
        if self.lock.acquire(blocking=False):  # (held already -- perhaps by this thread: they absorb)
            try:
                self._absorb()
            finally:
                self.lock.release()

    def trim(self):
        with self.lock:
            self._absorb()
            self.free.clear()


_result_pool = _ResultPool()


def trim_result_pool():
    """Give the kept result mappings (_ResultPool) back to the OS now."""
    _result_pool.trim()


def result_array(shape, dtype=np.float64):
    """The host array a bulk result is copied into.  Large results: an anonymous mapping of our
    own, 2 MiB-aligned and advised MADV_HUGEPAGE -- an ordinary writable numpy array to the caller,
    whose ``base`` chain ends in the mapping; when the last view of the array dies the mapping goes
    into a bounded pool for the next call's results (_ResultPool; pages the kernel may take back
    whenever it wants memory) or, beyond the pool's bounds, back to the OS at once.  Contents are
    UNDEFINED, as np.empty's.  Small ones, and everything when the caller opted into page-locked
    results: pinned_array()."""
    import weakref

    dtype = np.dtype(dtype)
    count = int(np.prod(shape, dtype=np.int64))
    nbytes = count * dtype.itemsize
    if HUGE_RESULT_BYTES <= 0 or nbytes < HUGE_RESULT_BYTES or 0 < nbytes <= PINNED_RESULT_LIMIT:
        return pinned_array(shape, dtype)
    import mmap

    kept = _result_pool.take(nbytes)
    if kept is not None:
        cap, m, off = kept
    else:
        cap = (nbytes + _HUGE_PAGE - 1) // _HUGE_PAGE * _HUGE_PAGE
        try:
            m = mmap.mmap(-1, cap + _HUGE_PAGE, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        except (OSError, ValueError, OverflowError):
            return pinned_array(shape, dtype)
        probe = np.frombuffer(m, dtype=np.uint8, count=1)
        off = (-probe.ctypes.data) % _HUGE_PAGE
        del probe
        try:  # advice only: without it (a kernel without THP) the pages are small ones
            m.madvise(mmap.MADV_HUGEPAGE, off, cap)
        except (AttributeError, OSError, ValueError):
            pass
        _result_pool.mapped += 1
    root = np.frombuffer(m, dtype=dtype, count=count, offset=off)
    # (every view of `root` keeps `root` alive: this fires when the LAST of them is gone)
    weakref.finalize(root, _result_pool.give, cap, m, off).atexit = False
    _log_range("result array (anonymous huge-page mapping, ours; written by host memcpy only)",
               root.ctypes.data, nbytes)
    return root.reshape(shape)


def owns_mapping(a):
    """True for an array handed out by result_array() as a mapping of its own"""
    import mmap

    while isinstance(a, np.ndarray):
        a = a.base
    return isinstance(a, memoryview) and isinstance(a.obj, mmap.mmap)


def _drain_one(ring, pending):
    """Oldest enqueued piece: wait for its DMA, copy it out of the staging buffer."""
    dst, n, i = pending.pop(0)
    ring.events[i].synchronize()
    ring.events[i] = None
    _host_copy(dst, ring.bufs[i][:n])


def _enqueue_download(out, dev, stream, ring, pending):
    """Start copying the device tensor ``dev`` into the numpy array ``out`` on ``stream``.  Pageable
    ``out``: piece by piece through ``ring``; on return every piece's DMA is enqueued and all but
    the last ring.depth-1 pieces are in ``out`` -- those wait in ``pending`` (a list the caller
    owns, shared by consecutive calls so that the pipeline does not run dry between arrays).
    Page-locked or small ``out``: one copy on ``stream``, nothing pending."""
    host = torch.from_numpy(out)
    if host.dtype != dev.dtype or host.numel() != dev.numel():
        raise ValueError(f"download of a {dev.dtype} tensor of {dev.numel()} elements into a "
                         f"{host.dtype} array of {host.numel()}")
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
        if src is not dev:  # the packing kernel reads `dev` on `stream`: its memory must not go
            dev.record_stream(stream)  # back to its own stream's allocator before that ran
    hb, db = _bytes_view(host), _bytes_view(src)
    _log_range("download destination (pageable: written by host memcpy only)", host.data_ptr(),
               nbytes)
    step = PIECE_BYTES // 8 * 8
    for off in range(0, nbytes, step):
        n = min(step, nbytes - off)
        if len(pending) == ring.depth - 1:
            _drain_one(ring, pending)
        i, buf = ring.acquire()
        with torch.cuda.stream(stream):
            buf[:n].copy_(db[off:off + n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        ring.events[i] = ev
        pending.append((hb[off:off + n], n, i))
    src.record_stream(stream)


def download_into(out, dev, stream=None):
    """Copy the device tensor ``dev`` into the numpy array ``out`` (same shape / dtype, contiguous).
    Page-locked ``out`` (pinned_array): one asynchronous DMA on ``stream`` -- the caller
    synchronises the stream before reading.  Pageable ``out``: piecewise through the ring,
    complete on return."""
    device = dev.device
    stream = stream if stream is not None else torch.cuda.current_stream(device)
    ring, pending = _ring(device), []
    _enqueue_download(out, dev, stream, ring, pending)
    while pending:
        _drain_one(ring, pending)


class Downloader:
    """Results -> host arrays on a worker thread with a stream and a staging ring of its own.

    The caller's loop ``submit()``s each time chunk's (host array, device tensor) pairs right
    after enqueuing the kernels that produce them and goes on to the next chunk; the worker keeps
    ONE pipeline of staging pieces running across arrays and chunks (DMA of piece k+1 and k+2
    while piece k is copied out), so the link does not idle between chunks -- draining after
    every array, as download_into() does, left it idle 15 % of the time on the reference's recorded
    call (profiles/r04_hostio_breakdown.log).  At most ``depth`` chunks are outstanding: submit()
    blocks beyond that, which bounds the device memory held by results still on their way out.

        with hostio.Downloader(device) as results:
            for chunk: ...; results.submit([(eta[t0:t1], e), (drho[t0:t1], d)])
        # here every array is complete
    """

    def __init__(self, device, depth=2):
        import collections
        from concurrent.futures import ThreadPoolExecutor

        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.depth = int(depth)
        # one buffer more than the shared rings: two DMAs stay queued while a piece is copied out,
        # so a slow copy-out of one piece does not idle the link
        self._ring = new_ring(int(os.environ.get("MOMLEVEL_AMD_DOWNLOAD_RING", "4")))
        self._pending = []  # touched by the worker thread only
        self._jobs = collections.deque()
        self._pool = ThreadPoolExecutor(1, thread_name_prefix="mlx-download")

    def submit(self, pairs):
        """``pairs``: [(numpy array, device tensor)]; the tensors are results of work enqueued on
        the device's CURRENT stream (of the calling thread)."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        while len(self._jobs) >= self.depth:
            self._jobs.popleft().result()  # (re-raises what the worker raised)
        self._jobs.append(self._pool.submit(self._run, list(pairs), ev))

    def _run(self, pairs, ev):
        with torch.cuda.device(self.device):
            self.stream.wait_event(ev)
            for out, dev in pairs:
                _enqueue_download(out, dev, self.stream, self._ring, self._pending)

    def _flush(self):
        while self._pending:
            _drain_one(self._ring, self._pending)

    def finish(self):
        """Block until every submitted array is complete."""
        while self._jobs:
            self._jobs.popleft().result()
        self._pool.submit(self._flush).result()
        self.stream.synchronize()  # (page-locked / small arrays: plain asynchronous copies)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        try:
            if exc_type is None:
                self.finish()
            else:  # the caller failed: let what is in flight land, report the caller's error
                for j in self._jobs:
                    j.cancel()
                try:
                    self._pool.submit(self._flush).result()
                    self.stream.synchronize()
                except Exception:
                    pass
        finally:
            self._pool.shutdown(wait=True)
        return False


class _DeferredSlice:
    """``a[i0:i1]`` of an array that is READ by slicing (dask / netCDF4 / h5py-like), not taken
    yet: the read happens where the value is needed -- in the Uploader's worker thread."""

    def __init__(self, a, i0, i1):
        self.a, self.i0, self.i1 = a, i0, i1
        self.dtype = a.dtype
        self.shape = (i1 - i0,) + tuple(a.shape[1:])

    def read(self):
        """the slice as the source returns it (a masked array from a netCDF4-like source)"""
        return self.a[self.i0:self.i1]

    def __array__(self, dtype=None, copy=None):
        out = as_plain(self.read())  # (masked elements become NaN)
        return out if dtype is None else out.astype(dtype, copy=False)


def leading_slice(a, i0, i1):
    """``a[i0:i1]`` for the Uploader: a view of a numpy array, a deferred read of anything else."""
    if isinstance(a, np.ndarray) and not isinstance(a, np.ma.MaskedArray):
        return a[i0:i1]
    return _DeferredSlice(a, i0, i1)  # (a masked array is NaN-filled slice by slice, by the worker)


class Uploader:
    """Host arrays -> fresh device tensors on a worker thread with a stream and a staging ring of
    its own (what engine.TimeChunks does for theta/S, for callers that chunk something else):

        up = hostio.Uploader(device)
        nxt = up.submit([a0, b0])
        for k ...:
            tensors, ready = nxt.result(); nxt = up.submit([a_k+1, b_k+1])
            torch.cuda.current_stream().wait_event(ready); ...kernels on tensors...
        up.close()

    The tensors belong to the stream that was current when the Uploader was made (the caching
    allocator ties a block to a stream), the copies run on the Uploader's own."""

    def __init__(self, device):
        from concurrent.futures import ThreadPoolExecutor

        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._main = torch.cuda.current_stream(self.device)
        self._ring = new_ring()
        self._pool = ThreadPoolExecutor(1, thread_name_prefix="mlx-upload")

    def submit(self, arrays):
        """Future of ([device tensor per array], event that completes the copies)."""
        return self._pool.submit(self._run, list(arrays))

    def _run(self, arrays):
        import warnings

        out = []
        with torch.cuda.device(self.device):
            for a in arrays:
                if isinstance(a, _DeferredSlice):
                    a = a.read()  # (here, in the worker: the read of a lazy source)
                a, mask = split_masked(a)  # (a masked array: NaN-filled while it is staged)
                a = np.ascontiguousarray(a)
                if a.dtype not in (np.float32, np.float64):
                    a = a.astype(np.float64)
                with warnings.catch_warnings():  # read-only views are only read
                    warnings.simplefilter("ignore", UserWarning)
                    host = torch.from_numpy(a)
                with torch.cuda.stream(self._main):
                    dev = torch.empty(host.shape, dtype=host.dtype, device=self.device)
                # `dev` may reuse memory the consumer's stream is done with: what is enqueued
                # there so far goes first
                self.stream.wait_stream(self._main)
                upload(host, dev, stream=self.stream, ring=self._ring, mask=mask)
                dev.record_stream(self.stream)
                out.append(dev)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def close(self):
        self._pool.shutdown(wait=True)


def to_host(t):
    """Device tensor -> numpy array (through the staging ring when not small; synchronises)."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        return t.detach().numpy() if isinstance(t, torch.Tensor) else as_plain(t)
    t = t.detach()
    if t.dtype not in (torch.float32, torch.float64) or t.numel() * t.element_size() < SMALL_BYTES:
        return t.cpu().numpy()
    out = result_array(tuple(t.shape), np.float32 if t.dtype == torch.float32 else np.float64)
    stream = torch.cuda.current_stream(t.device)
    download_into(out, t, stream)
    stream.synchronize()
    return out
