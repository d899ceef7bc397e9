"""CPU: the C-ABI library loads without a GPU and exports exactly what include/momlevel_hip.h
declares; the ctypes prototypes (momlevel_amd/_lib.py) cover the same set.  No compute calls.
"""

import ctypes
import os
import re

import numpy as np
import pytest

from momlevel_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "momlevel_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mlx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "mlx_steric_global" in syms and "mlx_steric_local" in syms and "mlx_eos_map" in syms
    assert len(syms) >= 12


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in momlevel_hip.h but not exported"


def test_binding_covers_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_version_and_error_buffer():
    lib = _lib.load()
    assert lib.mlx_version() == _lib.ABI_VERSION
    text = open(HEADER).read()
    assert int(re.search(r"#define MLX_ABI_VERSION (\d+)", text).group(1)) == _lib.ABI_VERSION
    assert isinstance(_lib.last_error(), str)


def test_enum_values_match_header():
    text = open(HEADER).read()
    for name, val in re.findall(r"#define (MLX_[A-Z0-9_]+)\s+\(?(-?\d+)\)?", text):
        short = name[4:]
        if hasattr(_lib, short):
            assert getattr(_lib, short) == int(val), name


def test_argument_errors_need_no_gpu():
    """Argument validation happens before any HIP call, so it is testable here."""
    lib = _lib.load()
    rc = lib.mlx_steric_global(None, None, 0, None, None, 1, 0, 1, 1, 1, 0, 0, 0, None, None, 0, None)
    assert rc == -1  # MLX_E_NULL
    assert "NULL" in _lib.last_error()
    assert lib.mlx_nansum(None, 0, None, None, 0, None) == -1
    assert lib.mlx_steric_global_workspace_bytes(0, 1, 1) == 0
    assert lib.mlx_steric_global_workspace_bytes(120, 75, 1080 * 1440) > 0


def test_null_pressure_requires_scalar_mode():
    """with p == NULL (allowed for the linear EOS) the array pressure modes would index a
    placeholder pointer; every entry point rejects them before any launch (fake, never dereferenced
    pointers: the check precedes all HIP calls)."""
    lib = _lib.load()
    fake = 1 << 20  # 16-byte aligned, non-NULL
    for p_mode in (_lib.P_ZPROF, _lib.P_FULL3D, _lib.P_FULL4D):
        rc = lib.mlx_steric_global(fake, fake, 0, fake, None, p_mode, _lib.EOS_LINEAR, 2, 3, 32, 96,
                                   96, 0, fake, fake, 1 << 20, None)
        assert rc == -1 and "MLX_P_SCALAR" in _lib.last_error()
        rc = lib.mlx_eos_map(fake, fake, 0, None, p_mode, _lib.EOS_LINEAR, 0, 2, 3, 32, 96, 96, 0,
                             fake, None)
        assert rc == -1
        rc = lib.mlx_steric_local(fake, fake, 0, fake, fake, fake, None, None, None, p_mode,
                                  _lib.EOS_LINEAR, -1.0 / 1035.0, 2, 3, 32, 96, 96, 0, None, fake,
                                  None)
        assert rc == -1


def test_build_kind():
    assert _lib.load().mlx_build_kind() == _lib.BUILD_HIP


def test_host_copy_needs_no_gpu():
    """mlx_host_copy (v6): host memory only -- every byte, any alignment, one thread or a team,
    concurrent callers on disjoint ranges; overlapping ranges and bad team sizes are refused"""
    import threading

    lib = _lib.load()
    r = np.random.default_rng(3)
    n = (5 << 20) + 77
    src = r.integers(0, 255, n, dtype=np.uint8)
    for streaming in (0, 1):
        for threads in (1, 3, 8):
            for off_s, off_d, m in ((0, 0, n), (1, 5, n - 9), (3, 0, 4097), (0, 7, 0), (64, 64, 100)):
                dst = np.full(n, 0xEE, dtype=np.uint8)
                rc = lib.mlx_host_copy(dst.ctypes.data + off_d, src.ctypes.data + off_s, m, threads,
                                       streaming)
                assert rc == 0
                assert np.array_equal(dst[off_d:off_d + m], src[off_s:off_s + m])
                assert (dst[:off_d] == 0xEE).all() and (dst[off_d + m:] == 0xEE).all()
    outs = [np.zeros(n, dtype=np.uint8) for _ in range(3)]
    jobs = [threading.Thread(target=lambda o=o: [lib.mlx_host_copy(o.ctypes.data, src.ctypes.data,
                                                                   n, 4, 1) for _ in range(4)])
            for o in outs]
    [j.start() for j in jobs]
    [j.join() for j in jobs]
    assert all(np.array_equal(o, src) for o in outs)
    a = src.ctypes.data
    assert lib.mlx_host_copy(a + 8, a, 64, 1, 1) == -2 and "overlap" in _lib.last_error()
    assert lib.mlx_host_copy(outs[0].ctypes.data, a, 64, 0, 1) == -2
    assert lib.mlx_host_copy(outs[0].ctypes.data, a, 64, 65, 1) == -2
    assert lib.mlx_host_copy(None, a, 64, 1, 1) == -1


def test_host_copy_masked_needs_no_gpu():
    """mlx_host_copy_masked (v7): dst = mask ? NaN : src on host memory, float32 and float64 bit
    patterns (signalling NaNs, denormals, -0.0 in src pass through untouched), one thread or a team,
    lengths that are not whole slices, guard bytes either side; bad arguments are refused"""
    lib = _lib.load()
    r = np.random.default_rng(11)
    for dt, nan_bits in ((np.float32, 0x7FC00000), (np.float64, 0x7FF8000000000000)):
        ut = np.uint32 if dt == np.float32 else np.uint64
        n = (3 << 20) + 1237
        src = r.integers(0, np.iinfo(ut).max, n, dtype=ut, endpoint=True)  # every bit pattern
        mask = r.random(n) < 0.3
        mask[:5000] = True
        mask[5000:9000] = False
        # runs of 1..300 equal mask bytes over the second half: whole 32-element blocks with nothing
        # masked (plain copy), with everything masked (NaN stored, source unread) and mixed ones
        # (the blend) at every phase against the 32-byte alignment of the destination
        runs = r.integers(1, 300, 40000)
        runmask = np.repeat(np.arange(runs.size) % 2 == 0, runs)[: n - n // 2]
        mask[n // 2: n // 2 + runmask.size] = runmask
        want = np.where(mask, ut(nan_bits), src)
        for threads in (1, 3, 8):
            for lo, m in ((0, n), (7, n - 19), (4096, 4097), (11, 0), (100, 1)):
                dst = np.full(n + 2, 0xEEEEEEEE, dtype=ut)
                rc = lib.mlx_host_copy_masked(dst.ctypes.data + dst.itemsize * (1 + lo),
                                              src.ctypes.data + src.itemsize * lo,
                                              mask.ctypes.data + lo, m, dst.itemsize, threads)
                assert rc == 0, _lib.last_error()
                assert np.array_equal(dst[1 + lo:1 + lo + m], want[lo:lo + m])
                assert (dst[:1 + lo] == 0xEEEEEEEE).all() and (dst[1 + lo + m:] == 0xEEEEEEEE).all()
        out = np.empty(n, dtype=dt)
        assert lib.mlx_host_copy_masked(out.ctypes.data, src.ctypes.data, mask.ctypes.data, n,
                                        out.itemsize, 4) == 0
        assert np.isnan(out[mask]).all()
    a, mk = src.ctypes.data, mask.ctypes.data
    d = np.empty(64, dtype=np.float64)
    assert lib.mlx_host_copy_masked(None, a, mk, 8, 8, 1) == -1
    assert lib.mlx_host_copy_masked(d.ctypes.data, a, None, 8, 8, 1) == -1
    assert lib.mlx_host_copy_masked(d.ctypes.data, a, mk, 8, 2, 1) == -3
    assert lib.mlx_host_copy_masked(d.ctypes.data, a, mk, 8, 8, 0) == -2
    assert lib.mlx_host_copy_masked(d.ctypes.data + 4, a, mk, 4, 8, 1) == -5
    assert lib.mlx_host_copy_masked(a + 8, a, mk, 8, 8, 1) == -2 and "overlap" in _lib.last_error()
    assert lib.mlx_host_copy_masked(d.ctypes.data, a, mk, 0, 8, 1) == 0


def test_large_masked_arrays_are_filled_by_the_library():
    """labeled.as_plain hands floating masked arrays above 4 MiB to mlx_host_copy_masked (the
    host copy team) -- same result as numpy's where(), the caller's array untouched; small, integer
    and non-contiguous ones stay with numpy"""
    from momlevel_amd import labeled

    r = np.random.default_rng(2)
    a = r.normal(size=(6, 50, 60, 70)).astype(np.float32)  # 5 MB
    mask = r.random(a.shape) < 0.4
    m = np.ma.masked_array(a.copy(), mask=mask, fill_value=1e20)
    m.data[mask] = 1e20
    calls = []
    real = labeled._native_fill
    labeled._native_fill = lambda d, k: calls.append(d.nbytes) or real(d, k)
    try:
        got = labeled.as_plain(m)
        assert calls == [a.nbytes]
        assert got.dtype == np.float32 and np.array_equal(got, np.where(mask, np.float32(np.nan), a),
                                                           equal_nan=True)
        assert (m.data[mask] == np.float32(1e20)).all()
        got = labeled.as_plain(np.ma.masked_array(a.astype(np.float64), mask=True))  # scalar mask
        assert calls[-1] == a.size * 8 and np.isnan(got).all()
        n = len(calls)
        labeled.as_plain(m[:1])                       # small
        labeled.as_plain(m.transpose(0, 2, 1, 3))     # not C-contiguous: numpy
        assert len(calls) == n + 1 and calls[-1] == a.nbytes  # (the transposed one asked, and got None)
        t = labeled.as_plain(m.transpose(0, 2, 1, 3))
        assert np.array_equal(t, np.where(mask, np.float32(np.nan), a).transpose(0, 2, 1, 3), equal_nan=True)
    finally:
        labeled._native_fill = real


def test_plain_and_masked_copies_share_the_team_concurrently():
    """the upload worker (masked slices), the download worker (plain copies) and the caller's thread
    (as_plain of an in-memory masked array) all use ONE team of native threads at the same time"""
    import threading

    lib = _lib.load()
    r = np.random.default_rng(5)
    n = (4 << 20) + 333
    src = r.normal(size=n).astype(np.float32)
    mask = r.random(n) < 0.5
    want = np.where(mask, np.float32(np.nan), src)
    errors = []

    def masked():
        for _ in range(6):
            out = np.empty(n, dtype=np.float32)
            if lib.mlx_host_copy_masked(out.ctypes.data, src.ctypes.data, mask.ctypes.data, n, 4, 5) != 0 \
                    or not np.array_equal(out, want, equal_nan=True):
                errors.append("masked")

    def plain():
        for _ in range(6):
            out = np.empty(n, dtype=np.float32)
            if lib.mlx_host_copy(out.ctypes.data, src.ctypes.data, n * 4, 4, 1) != 0 or not np.array_equal(out, src):
                errors.append("plain")

    jobs = [threading.Thread(target=f) for f in (masked, plain, masked, plain)]
    [j.start() for j in jobs]
    [j.join() for j in jobs]
    assert errors == []


def _resident_fraction(addr, nbytes):
    """fraction of the pages of [addr, addr+nbytes) that are resident (mincore)"""
    page = os.sysconf("SC_PAGESIZE")
    lo = addr // page * page
    n = -(-(addr + nbytes - lo) // page)
    vec = (ctypes.c_ubyte * n)()
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mincore.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    assert libc.mincore(lo, n * page, vec) == 0, ctypes.get_errno()
    return sum(b & 1 for b in vec) / n


def test_result_arrays_own_their_mapping_and_a_pool_keeps_dead_ones(monkeypatch):
    """hostio.result_array: large results live in a 2 MiB-aligned anonymous mapping of their own
     -- a writable numpy array whose memory goes back to the OS when the last view
    dies, or into a bounded pool that serves the next result of that size (warm pages)."""
    import gc
    import weakref

    from momlevel_amd import hostio

    monkeypatch.setattr(hostio, "HUGE_RESULT_BYTES", 8 << 20)
    monkeypatch.setenv("MOMLEVEL_AMD_RESULT_POOL_GIB", "0")  # first without the pool of kept mappings
    small = hostio.result_array((100, 100), np.float64)
    assert type(small) is np.ndarray and small.base is None and not hostio.owns_mapping(small)
    shape = (6, 3, 40, 2048)  # 11.8 MB of float64
    out = hostio.result_array(shape, np.float64)
    assert out.shape == shape and out.dtype == np.float64 and out.flags["C_CONTIGUOUS"]
    assert out.flags["WRITEABLE"] and out.flags["ALIGNED"] and out.ctypes.data % (2 << 20) == 0
    assert hostio.owns_mapping(out) and hostio.owns_mapping(out[2:4, 1])
    assert not hostio.owns_mapping(np.frombuffer(bytearray(64), dtype=np.uint8))
    f32 = hostio.result_array((5, 1 << 20), np.float32)
    assert f32.dtype == np.float32 and hostio.owns_mapping(f32)
    # the mapping dies with the last view of the array, not before
    root = out
    while isinstance(root, np.ndarray):
        root = root.base
    mapping = weakref.ref(root.obj)
    addr, nbytes = out.ctypes.data, out.nbytes
    view = out[1:3]
    del out, root
    gc.collect()
    assert mapping() is not None
    view[...] = 7.0  # still mapped, still writable
    assert float(view.sum()) == 7.0 * view.size
    del view
    gc.collect()
    assert mapping() is None
    with open("/proc/self/maps") as f:
        assert not any(line.startswith(f"{addr:x}-") for line in f), "the result mapping is still there"
    out = hostio.result_array(shape, np.float64)
    assert _resident_fraction(out.ctypes.data, out.nbytes) < 0.05  # fresh: nothing touched yet
    del out
    gc.collect()
    # the pool of kept mappings: a dead result's mapping serves the next result of that size (warm
    # pages: no zeroing pass before the copy-out), bounded in bytes, emptied on request
    monkeypatch.setenv("MOMLEVEL_AMD_RESULT_POOL_GIB", "0.02")  # 21 MB: one 11.8 MB mapping fits
    pool = hostio._result_pool
    assert pool.free == []
    first = hostio.result_array(shape, np.float64)
    addr = first.ctypes.data
    first[...] = 3.0
    keep = first[5]
    del first
    gc.collect()
    assert pool.free == []  # a view is alive: the mapping is still the caller's
    other = hostio.result_array(shape, np.float64)
    assert other.ctypes.data != addr
    del keep
    gc.collect()
    assert len(pool.free) == 1
    reused_before = pool.reused
    again = hostio.result_array((5,) + shape[1:], np.float64)  # a little smaller: still fits
    assert again.ctypes.data == addr and pool.reused == reused_before + 1 and pool.free == []
    assert hostio.owns_mapping(again) and again.flags["WRITEABLE"]
    tiny = hostio.result_array((2,) + shape[1:], np.float64)  # far smaller: a mapping of its own
    assert tiny.ctypes.data != addr
    del again, other
    gc.collect()
    assert len(pool.free) == 1  # the byte bound: the older mapping went back to the OS
    hostio.trim_result_pool()
    assert pool.free == []
    # a finalizer may fire wherever the garbage collector runs -- also while THIS thread holds the
    # pool's lock (inside take()): it must never wait for it
    late = hostio.result_array(shape, np.float64)
    late_addr = late.ctypes.data
    assert pool.lock.acquire(blocking=False)
    try:
        del late
        gc.collect()  # give() runs here, the lock is held: the mapping is parked, nothing blocks
        assert pool.free == [] and len(pool.returned) == 1
    finally:
        pool.lock.release()
    again = hostio.result_array(shape, np.float64)  # take() absorbs what was parked
    assert again.ctypes.data == late_addr and len(pool.returned) == 0
    del again
    gc.collect()
    hostio.trim_result_pool()
    assert pool.free == [] and len(pool.returned) == 0
    del tiny
    gc.collect()
    hostio.trim_result_pool()
    # disabled: numpy's allocation as before
    monkeypatch.setattr(hostio, "HUGE_RESULT_BYTES", 0)
    assert hostio.result_array(shape, np.float64).base is None
