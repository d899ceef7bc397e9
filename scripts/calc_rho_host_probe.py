#!/usr/bin/env python3
"""derived.calc_rho-style call (eos.wright.density(T, S, p(z))) on HOST arrays, end to end: the
pipelined piecewise evaluation (eos/_dispatch._evaluate_host_chunked: upload, kernel and result
download of consecutive pieces overlap) against the same call done piece after piece with nothing
overlapping (pieces of 2^28 elements, each uploaded, evaluated and downloaded in turn: what the
code did until round 4).

    python scripts/calc_rho_host_probe.py [--nt 12] [--dtype float64]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import core, hostio, synthetic  # noqa: E402
from momlevel_amd.eos import _dispatch, wright  # noqa: E402


def serial(T, S, p):
    rows = max(1, (1 << 28) // int(np.prod(T.shape[1:])))
    out = np.empty(T.shape, dtype=np.float64)
    for i0 in range(0, T.shape[0], rows):
        i1 = min(i0 + rows, T.shape[0])
        res = _dispatch.evaluate("wright", "density", hostio.to_device(T[i0:i1], "cuda"),
                                 hostio.to_device(S[i0:i1], "cuda"), hostio.to_device(p, "cuda"))
        hostio.download_into(out[i0:i1], res)  # (through the staging ring, complete on return)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=12)
    ap.add_argument("--nz", type=int, default=75)
    ap.add_argument("--dtype", default="float64")
    a = ap.parse_args()
    ny, nx = 1080, 1440
    shape = (a.nt, a.nz, ny, nx)
    tdt = torch.float32 if a.dtype == "float32" else torch.float64
    host = []
    for fid, lo, sc in ((1, -2.0, 34.0), (2, 30.0, 10.0)):
        dev = core.synth_field(shape, tdt, field_id=fid, lo=lo, scale=sc, seed=synthetic.SEED)
        host.append(hostio.to_host(dev))
        del dev
    torch.cuda.empty_cache()
    pz = np.linspace(1.0e5, 6.0e7, a.nz)[:, None, None]
    cells = int(np.prod(shape))
    itemsize = np.dtype(a.dtype).itemsize
    rep = {"call": "eos.wright.density(T, S, p(z)) on host arrays (what derived.calc_rho runs)",
           "shape": list(shape), "dtype": a.dtype, "GB_in": round(2 * cells * itemsize / 1e9, 2),
           "GB_out": round(cells * 8 / 1e9, 2)}
    got = None
    for name, fn in (("pipelined", lambda: wright.density(host[0], host[1], pz)),
                     ("piece_after_piece", lambda: serial(host[0], host[1], pz))):
        walls = []
        for _ in range(3):
            res = None
            t0 = time.perf_counter()
            res = fn()
            walls.append(round(time.perf_counter() - t0, 3))
        rep[name + "_wall_s"] = walls
        rep[name + "_GB/s_in_plus_out"] = round((2 * cells * itemsize + cells * 8) / min(walls) / 1e9, 1)
        if got is None:
            got = res
        else:
            rep["same_bits"] = bool(np.array_equal(got, res, equal_nan=True))
    print(json.dumps(rep), flush=True)
    # derived.calc_n2 on the same host fields: groups of time steps pipelined the same way
    from momlevel_amd import derived
    from momlevel_amd.labeled import DataArray

    z = np.cumsum(2.0 * 1.075 ** np.arange(a.nz)) - 1.0
    dims = ("time", "z_l", "yh", "xh")
    coords = {"z_l": DataArray(z, ("z_l",))}
    Td, Sd = DataArray(host[0], dims, coords), DataArray(host[1], dims, coords)
    walls = []
    for _ in range(3):
        n2 = None
        t0 = time.perf_counter()
        n2 = derived.calc_n2(Td, Sd)
        walls.append(round(time.perf_counter() - t0, 3))
    print(json.dumps({"call": "derived.calc_n2(thetao, so) on host fields", "shape": list(shape),
                      "dtype": a.dtype, "wall_s": walls,
                      "GB/s_in_plus_out": round((2 * cells * itemsize + cells * 8) / min(walls) / 1e9, 1),
                      "Gcells/s_end_to_end": round(cells / min(walls) / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    main()
