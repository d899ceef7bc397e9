"""CPU: the C-ABI library loads without a GPU and exports exactly what include/momlevel_hip.h
declares; the ctypes prototypes (momlevel_amd/_lib.py) cover the same set.  No compute calls.
"""

import ctypes
import os
import re

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
