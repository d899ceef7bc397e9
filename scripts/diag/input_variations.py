"""Realistic input variations through the public API vs the numpy oracle (GPU box): strided /
reversed / Fortran-ordered views, integer fields, extra NaNs in wet cells, scalar kinds of patm."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from momlevel_amd import steric, synthetic
from momlevel_amd.labeled import DataArray, Dataset
from oracle import momlevel_numpy as o

def dataset(nt=5, nz=7, ny=12, nx=16, seed=3, dtype=np.float64):
    g = synthetic.make_grid(ny, nx, nz)
    r = np.random.default_rng(seed)
    mask = np.isnan(g["volcello"])
    T = np.where(mask[None], np.nan, r.normal(12.0, 6.0, (nt, nz, ny, nx))).astype(dtype)
    S = np.where(mask[None], np.nan, r.normal(35.0, 1.0, (nt, nz, ny, nx))).astype(dtype)
    return g, T, S

def build(g, T, S, vol=None):
    nt, nz, ny, nx = T.shape
    d = Dataset()
    d["time"] = DataArray(np.arange(nt, dtype=float), ("time",))
    d["z_l"] = DataArray(g["z_l"], ("z_l",)); d["z_i"] = DataArray(g["z_i"], ("z_i",))
    d["yh"] = DataArray(np.arange(ny, dtype=float), ("yh",)); d["xh"] = DataArray(np.arange(nx, dtype=float), ("xh",))
    dims = ("time", "z_l", "yh", "xh")
    d["thetao"] = DataArray(T, dims); d["so"] = DataArray(S, dims)
    d["volcello"] = DataArray(np.broadcast_to(g["volcello"], T.shape).copy() if vol is None else vol, dims)
    d["areacello"] = DataArray(g["areacello"], ("yh", "xh")); d["deptho"] = DataArray(g["deptho"], ("yh", "xh"))
    return d

def same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)

def check(name, d, T, S, g, **kw):
    ok = True
    for domain in ("local", "global"):
        for variant in ("steric", "thermosteric"):
            res, ref = steric(d, domain=domain, variant=variant, **kw)
            vol4 = np.broadcast_to(g["volcello"], T.shape).copy()
            ores, oref = o.steric(np.asarray(T), np.asarray(S), vol4, g["areacello"], g["z_l"], g["z_i"], g["deptho"],
                                  domain=domain, variant=variant, **{k: v for k, v in kw.items() if k == "patm"})
            if domain == "local":
                good = same(res[variant].values, ores[variant]) and same(res["delta_rho"].values, ores["delta_rho"])
            else:
                h = float(res["reference_height"])
                good = np.allclose(res[variant].values / h, ores["expansion_coeff"], rtol=0, atol=1e-12) and float(res[variant][0]) == 0.0
            ok &= good
            if not good:
                print("   MISMATCH", name, domain, variant)
    print(("ok      " if ok else "FAILED  ") + name, flush=True)

g, T, S = dataset()
check("baseline float64", build(g, T, S), T, S, g)
big = np.empty((5, 7, 12, 32)); big[..., ::2] = T
check("thetao a strided view (every second x of a wider array)", build(g, big[..., ::2], S), T, S, g)
check("thetao reversed twice (negative strides)", build(g, T[:, :, ::-1][:, :, ::-1], S), T, S, g)
check("Fortran-ordered so", build(g, T, np.asfortranarray(S)), T, S, g)
Ti = np.where(np.isnan(T), 0, np.round(T)).astype(np.int32); Tf = Ti.astype(np.float64)
Tf_nan = np.where(np.isnan(T), np.nan, Tf)
check("int32 thetao (land = 0: volcello masks it)", build(g, Ti, S), Ti, S, g)
T2 = T.copy(); T2[2, 1, 3:6, 4:9] = np.nan   # missing data in WET cells
check("extra NaNs in wet cells of thetao", build(g, T2, S), T2, S, g)
check("patm as numpy float32 scalar", build(g, T, S), T, S, g, patm=np.float32(101325.0))
check("patm as python int", build(g, T, S), T, S, g, patm=101325)
check("patm as 0-d float64 array", build(g, T, S), T, S, g, patm=np.array(101325.0))
g32, T32, S32 = dataset(dtype=np.float32)
check("float32 fields", build(g32, T32, S32), T32, S32, g32)
check("float32 thetao, float64 so", build(g32, T32, S32.astype(np.float64)), T32, S32.astype(np.float64), g32)
check("big-endian float32 fields", build(g32, T32.astype(">f4"), S32.astype(">f4")), T32, S32, g32)
