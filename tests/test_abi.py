"""CPU: the C-ABI library loads without a GPU and exports exactly what include/momlevel_hip.h
declares; the ctypes prototypes (momlevel_amd/_lib.py) cover the same set.  No compute calls.
"""

import ctypes
import os
import re

import numpy as np

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
