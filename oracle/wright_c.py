"""ctypes binding of oracle/libwright_oracle.so (C restatement).  TEST INFRASTRUCTURE ONLY."""

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libwright_oracle.so")


def build(force=False):
    src = os.path.join(HERE, "wright_fused.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B", "libwright_oracle.so"], check=True,
                       capture_output=True)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(LIB)
        dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        lib.oracle_density_slab.argtypes = [dp, dp, dp, ctypes.c_int64, ctypes.c_int64, dp]
        lib.oracle_density_slab.restype = None
        lib.oracle_masso_slab.argtypes = [dp, dp, dp, dp, ctypes.c_int64, ctypes.c_int64, dp]
        lib.oracle_masso_slab.restype = ctypes.c_double
        lib.oracle_num_threads.restype = ctypes.c_int
        lib.oracle_set_threads.argtypes = [ctypes.c_int]
        lib.oracle_set_threads.restype = None
        _lib = lib
    return _lib


def density_slab(T, S, pz):
    """rho (nz, ny, nx) for a z-profile pressure; bit-identical to momlevel_numpy.calc_rho."""
    T = np.ascontiguousarray(T, dtype=np.float64)
    S = np.ascontiguousarray(S, dtype=np.float64)
    pz = np.ascontiguousarray(pz, dtype=np.float64)
    nz = T.shape[0]
    rho = np.empty_like(T)
    load().oracle_density_slab(T, S, pz, nz, T.size // nz, rho)
    return rho


def masso_slab(T, S, vol, pz):
    """sum(rho*vol) [skipna] of one (nz, ny, nx) slab, fused and multithreaded."""
    T = np.ascontiguousarray(T, dtype=np.float64)
    S = np.ascontiguousarray(S, dtype=np.float64)
    vol = np.ascontiguousarray(vol, dtype=np.float64)
    pz = np.ascontiguousarray(pz, dtype=np.float64)
    nz = T.shape[0]
    zp = np.empty(nz)
    return load().oracle_masso_slab(T, S, vol, pz, nz, T.size // nz, zp)


def num_threads():
    return load().oracle_num_threads()


def set_threads(n):
    load().oracle_set_threads(int(n))
