"""ctypes binding of libmomlevel_hip.so (the C ABI declared in include/momlevel_hip.h).

The library is the ONLY compute backend of momlevel_amd: there is no CPU or
torch fallback.  If it is missing (or was never built) every entry point raises
``MomlevelHipError`` -- loudly, by design.
"""

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MOMLEVEL_AMD_LIB: bind another build of the SAME library (scripts/sanitize_host.py points it at
# the host-sanitized build of the HIP sources); the default is the in-tree libmomlevel_hip.so.
# load() refuses anything that is not a HIP build of the ABI (mlx_build_kind) or that lives under
# the checker's directory oracle/: the CPU restatement exports the same symbols and must never
# stand in for the device library.
LIB_PATH = os.environ.get("MOMLEVEL_AMD_LIB") or os.path.join(HERE, "libmomlevel_hip.so")
_ORACLE_DIR = os.path.join(os.path.dirname(HERE), "oracle")

# ---- constants mirrored from include/momlevel_hip.h --------------------------------
ABI_VERSION = 8
EOS_WRIGHT, EOS_LINEAR = 0, 1
FUNC_DENSITY, FUNC_DRHO_DTEMP, FUNC_DRHO_DSAL, FUNC_ALPHA, FUNC_BETA, FUNC_IBH = 0, 1, 2, 3, 4, 5
FUNC_DENSITY_REF = 6  # eos.linear.density(..., rho_ref=x): mlx_eos_map_promote only
P_SCALAR, P_ZPROF, P_FULL3D, P_FULL4D = 0, 1, 2, 3
DTYPE_F64, DTYPE_F32, DTYPE_F32_UPCAST = 0, 1, 2
DTYPE_T32_S64, DTYPE_T64_S32 = 3, 4  # theta / salinity of different dtypes (K1 / K2 only)
FLAG_SKIP_DRY = 1
FLAG_FMA = 2
BUILD_HIP, BUILD_HOST = 1, 2
KIND_F64, KIND_F32, KIND_WEAK = 0, 1, 2  # operand kinds of mlx_eos_map_promote
STRAT_N2, STRAT_TURNER = 0, 1  # mlx_stratification's func


def flag_tchunk(steps):
    """MLX_FLAG_TCHUNK(steps): K1 tuning hint (time steps per block, multiple of 8; 0 = default)."""
    return ((int(steps) // 8) & 0xFF) << 8


EOS_IDS = {"wright": EOS_WRIGHT, "linear": EOS_LINEAR}
FUNC_IDS = {
    "density": FUNC_DENSITY,
    "drho_dtemp": FUNC_DRHO_DTEMP,
    "drho_dsal": FUNC_DRHO_DSAL,
    "alpha": FUNC_ALPHA,
    "beta": FUNC_BETA,
}

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_dbl = ctypes.c_double
_sz = ctypes.c_size_t
_u64 = ctypes.c_uint64

# symbol -> (restype, argtypes); the single source of truth for tests/test_abi.py
SIGNATURES = {
    "mlx_version": (_int, []),
    "mlx_last_error": (_int, [ctypes.c_char_p, _sz]),
    "mlx_build_kind": (_int, []),
    "mlx_last_kernel": (_int, [ctypes.c_char_p, _sz]),
    "mlx_eos_map": (
        _int,
        [_vp, _vp, _int, _vp, _int, _int, _int, _i64, _i64, _i64, _i64, _i64, _int, _vp, _vp],
    ),
    "mlx_eos_map_promote": (
        _int,
        [_vp, _int, _i64, _vp, _int, _i64, _vp, _int, _i64, _int, _int, _dbl, _i64, _vp,
         ctypes.POINTER(ctypes.c_int), _vp],
    ),
    "mlx_inverse_barometer": (
        _int,
        [_vp, _vp, _int, _vp, _int, _int, _dbl, _i64, _i64, _i64, _i64, _i64, _vp, _vp],
    ),
    "mlx_steric_global_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "mlx_steric_global": (
        _int,
        [_vp, _vp, _int, _vp, _vp, _int, _int, _i64, _i64, _i64, _i64, _i64, _int, _vp, _vp, _sz,
         _vp],
    ),
    "mlx_steric_global_decomp_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "mlx_steric_global_decomp": (
        _int,
        [_vp, _vp, _vp, _vp, _int, _vp, _vp, _int, _int, _i64, _i64, _i64, _i64, _i64, _int, _vp,
         _vp, _sz, _vp],
    ),
    "mlx_fold_mask": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "mlx_steric_local": (
        _int,
        [_vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _dbl,
         _i64, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp],
    ),
    "mlx_steric_local_decomp": (
        _int,
        [_vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _dbl,
         _i64, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp, _i64, _vp],
    ),
    "mlx_nansum_workspace_bytes": (_sz, [_i64]),
    "mlx_nansum": (_int, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "mlx_masso": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _sz, _vp]),
    "mlx_group_weighted_mean": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "mlx_stream_probe": (_int, [_vp, _vp, _i64, _vp, _vp]),
    "mlx_stream_probe_mix": (_int, [_vp, _vp, _int, _i64, _vp, _int, _vp]),
    "mlx_valu_probe": (_int, [_i64, _vp, ctypes.POINTER(ctypes.c_int64), _vp]),
    "mlx_calc_dz": (_int, [_vp, _vp, _i64, _i64, _dbl, _dbl, _int, _int, _vp, _vp]),
    "mlx_host_copy": (_int, [_vp, _vp, _sz, _int, _int]),
    "mlx_host_copy_masked": (_int, [_vp, _vp, _vp, _sz, _int, _int]),
    "mlx_stratification": (_int, [_vp, _vp, _int, _vp, _i64, _i64, _i64, _int, _int, _vp, _int,
                                  _dbl, _dbl, _i64, _i64, _i64, _vp, _vp]),
    "mlx_adjust_negative_n2": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mlx_wave_speed_where_time0": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    "mlx_synth_field": (
        _int,
        [_vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _u64, _int,
         _dbl, _dbl, _vp, _vp],
    ),
}


class MomlevelHipError(RuntimeError):
    """Raised when the HIP library is missing or one of its calls fails."""


_lib = None


def load():
    """dlopen libmomlevel_hip.so (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MomlevelHipError(
            f"{LIB_PATH} not found: momlevel_amd has no CPU fallback. Build the HIP "
            "library with `python -m momlevel_amd.csrc.build` (needs hipcc)."
        )
    real = os.path.realpath(LIB_PATH)
    if os.path.commonpath([real, os.path.realpath(_ORACLE_DIR)]) == os.path.realpath(_ORACLE_DIR):
        raise MomlevelHipError(
            f"{LIB_PATH} is under oracle/ (the CPU checker): momlevel_amd binds the HIP library only"
        )
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the host
        raise MomlevelHipError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise MomlevelHipError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.mlx_version() != ABI_VERSION:
        raise MomlevelHipError(
            f"ABI mismatch: library {lib.mlx_version()} vs binding {ABI_VERSION}; rebuild"
        )
    if lib.mlx_build_kind() != BUILD_HIP:
        raise MomlevelHipError(
            f"{LIB_PATH} is build kind {lib.mlx_build_kind()}, not the HIP build "
            f"({BUILD_HIP}): momlevel_amd has no CPU backend"
        )
    _lib = lib
    return lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().mlx_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def last_kernel():
    """The kernel instantiation this thread's last K1 / K2 call launched (mlx_last_kernel)."""
    buf = ctypes.create_string_buffer(160)
    load().mlx_last_kernel(buf, 160)
    return buf.value.decode(errors="replace")


def check(status, what):
    """Turn a non-zero status of the C ABI into an exception."""
    if status != 0:
        kind = "argument error" if status < 0 else "hipError_t"
        raise MomlevelHipError(f"{what} failed: {kind} {status}: {last_error()}")
