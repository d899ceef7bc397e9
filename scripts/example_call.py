#!/usr/bin/env python3
"""The reference's one recorded real-size call, end to end: examples/example.ipynb cells 3-6 run
`momlevel.thermosteric(ds)` on time 60 x z_l 35 x yh 1080 x xh 1440, float32, with the default
domain="local" -- from HOST memory, delta_rho returned.  Here: the same call on momlevel_amd with
numpy-backed inputs of that shape (synthetic fields), wall-clock including the PCIe transfers both
ways.  (bench.py runs this with a checker that holds one time step against the numpy oracle and
times the oracle on it; the script by itself only times the product.)

    python scripts/example_call.py [--nt 60] [--nz 35]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import momlevel_amd as m  # noqa: E402
from momlevel_amd import core, hostio, synthetic  # noqa: E402
from momlevel_amd.labeled import DataArray, Dataset  # noqa: E402


def run(nt=60, nz=35, reps=2, ny=1080, nx=1440, checker=None, before_call=None, source="numpy"):
    """``source``: how thetao / so are held -- "numpy" (plain arrays), "masked" (numpy masked arrays
    with 1e20 under the mask: an in-memory netCDF4 read) or "masked_lazy" (read slice by slice, every
    slice a masked array: a netCDF4.Variable; tests/lazy_array.py)."""
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = hostio.to_device(g["volcello"], "cuda")
    kw = dict(seed=synthetic.SEED, mask3d=vol0)
    shape = (nt, nz, ny, nx)
    # host-resident float32 fields (generated on the device, brought to pageable host memory)
    host = {}
    for name, fid, lo, sc in (("thetao", 1, -2.0, 34.0), ("so", 2, 30.0, 10.0)):
        dev = core.synth_field(shape, torch.float32, field_id=fid, lo=lo, scale=sc, **kw)
        host[name] = hostio.to_host(dev)  # a plain (pageable) numpy array, as a user has
        assert not torch.from_numpy(host[name]).is_pinned()
        del dev
    torch.cuda.empty_cache()
    d = Dataset()
    d["time"] = DataArray(np.arange(nt, dtype=float), ("time",))
    d["z_l"] = DataArray(g["z_l"], ("z_l",))
    d["z_i"] = DataArray(g["z_i"], ("z_i",))
    dims = ("time", "z_l", "yh", "xh")
    if source == "numpy":
        d["thetao"] = DataArray(host["thetao"], dims)
        d["so"] = DataArray(host["so"], dims)
    else:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from lazy_array import PrebuiltMaskedLazy as MaskedLazy, as_masked

        def one_cell(a):  # a full-size mask with ONE element set: what the masked route costs by itself
            mask = np.zeros(a.shape, dtype=bool)
            mask[0, 0, 0, 0] = True
            return np.ma.masked_array(np.nan_to_num(a, nan=1e20), mask=mask)

        held = {name: (MaskedLazy(host[name]) if source == "masked_lazy" else
                       one_cell(host[name]) if source == "masked_onecell" else as_masked(host[name]))
                for name in ("thetao", "so")}
        t_wrap = time.perf_counter()  # (an in-memory masked array is NaN-filled HERE, once: as_plain)
        for name in ("thetao", "so"):
            d[name] = DataArray(held[name], dims)
        wrap_s = time.perf_counter() - t_wrap
    d["volcello"] = DataArray(np.broadcast_to(g["volcello"].astype(np.float32), shape), dims)
    d["areacello"] = DataArray(g["areacello"].astype(np.float32), ("yh", "xh"))
    d["deptho"] = DataArray(g["deptho"], ("yh", "xh"))
    cells = nt * nz * ny * nx
    out = {"call": "thermosteric(ds)  # domain='local', float32 thetao/so from host memory, delta_rho returned",
           "source": source,
           "shape_t_z_y_x": list(shape), "cells": cells,
           "host_bytes_in_GB": round(2 * cells * 4 / 1e9, 2),
           "host_bytes_out_GB": round((cells + nt * ny * nx) * 8 / 1e9, 2)}
    if source != "numpy":
        out["DataArray_construction_s"] = round(wrap_s, 3)
    walls = []
    res = ref = drho = eta = None
    for _ in range(reps):
        del res, ref, drho, eta  # (freeing 27 GB of earlier results is the caller's time, not the call's)
        import gc

        gc.collect()
        if before_call is not None:
            before_call()
        t0 = time.perf_counter()
        with hostio.roctx_range("thermosteric(ds) on host float32 inputs"):  # (MOMLEVEL_AMD_ROCTX=1)
            res, ref = m.thermosteric(d)
            drho = res["delta_rho"].values  # host arrays: the call has synchronised
            eta = res["thermosteric"].values
        walls.append(time.perf_counter() - t0)
        assert drho.shape == shape and eta.shape == (nt, ny, nx) and drho.dtype == np.float64
    out["wall_s"] = [round(w, 3) for w in walls]
    best = min(walls)
    out["Mcells/s_end_to_end"] = round(cells / best / 1e6, 1)
    # thermosteric streams theta only (S is held at the reference state: one slab)
    out["host_bytes_streamed_in_GB"] = round(cells * 4 / 1e9, 2)
    out["GB/s_host_link_in_plus_out"] = round((cells * 4 + (cells + nt * ny * nx) * 8) / best / 1e9, 1)
    if checker is not None:  # bench.py's CPU leg: the oracle on one step of the same call
        out.update(checker(host, g, drho, eta, best))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=60)
    ap.add_argument("--nz", type=int, default=35)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--source", choices=["numpy", "masked", "masked_lazy", "masked_onecell"], default="numpy")
    a = ap.parse_args()
    print(json.dumps(run(a.nt, a.nz, a.reps, source=a.source)), flush=True)


if __name__ == "__main__":
    main()
