"""End-to-end (PCIe-inclusive) rate of the public steric() on HOST (numpy) inputs, config 2.

    python scripts/ingest_check.py

DESIGN.md 2.3: this rate is bounded by the host link and the staging copy, and is never the
bench `value` (which is measured with theta/S resident in HBM).
"""

import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import momlevel_amd as m  # noqa: E402
from momlevel_amd import synthetic  # noqa: E402
from momlevel_amd.labeled import DataArray, Dataset  # noqa: E402


def dataset(nt, nz, ny, nx, dtype=np.float64):
    g = synthetic.make_grid(ny, nx, nz)
    kw = dict(seed=synthetic.SEED, mask3d=g["volcello"], dtype=dtype)
    d = Dataset()
    d["time"] = DataArray(np.arange(nt, dtype=float), ("time",))
    d["z_l"] = DataArray(g["z_l"], ("z_l",))
    d["z_i"] = DataArray(g["z_i"], ("z_i",))
    dims = ("time", "z_l", "yh", "xh")
    d["thetao"] = DataArray(synthetic.field_numpy((nt, nz, ny, nx), field_id=1, lo=-2.0, scale=34.0, **kw), dims)
    d["so"] = DataArray(synthetic.field_numpy((nt, nz, ny, nx), field_id=2, lo=30.0, scale=10.0, **kw), dims)
    d["volcello"] = DataArray(np.broadcast_to(g["volcello"], (nt, nz, ny, nx)), dims)
    d["areacello"] = DataArray(g["areacello"], ("yh", "xh"))
    d["deptho"] = DataArray(g["deptho"], ("yh", "xh"))
    return d


def main():
    nt, nz, ny, nx = 48, 75, 576, 360  # 12 GB of fp64 theta+S: several upload chunks
    for dtype in (np.float64, np.float32):
        d = dataset(nt, nz, ny, nx, dtype)
        cells = nt * nz * ny * nx
        for domain in ("global", "local"):
            dt, free_s = float("inf"), 0.0
            res = ref = None
            for rep in range(4):  # rep 0 pays the runtime's first-touch page pinning
                # giving the previous call's results back to the OS is the caller's time, not the
                # call's (6 GB of delta_rho: timed separately)
                t0 = time.perf_counter()
                del res, ref
                free_s = max(free_s, time.perf_counter() - t0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res, ref = m.steric(d, domain=domain)
                torch.cuda.synchronize()
                if rep:
                    dt = min(dt, time.perf_counter() - t0)
            inb = 2 * cells * np.dtype(dtype).itemsize
            outb = cells * 8 if domain == "local" else 0
            print(f"{np.dtype(dtype).name} {domain:6s}: {dt*1e3:8.1f} ms  {cells/dt/1e6:8.1f} Mcells/s  "
                  f"H2D {inb/dt/1e9:5.1f} GB/s  D2H {outb/dt/1e9:5.1f} GB/s   "
                  f"(freeing the previous results: {free_s*1e3:.0f} ms)")
            del res, ref


def variants_check():
    """three separate calls vs the steric_variants extension (one upload)."""
    nt, nz, ny, nx = 48, 75, 576, 360
    d = dataset(nt, nz, ny, nx, np.float64)
    cells = nt * nz * ny * nx
    for domain in ("global", "local"):
        best3, best1 = float("inf"), float("inf")
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            keep = [m.steric(d, variant=v, domain=domain)
                    for v in ("steric", "thermosteric", "halosteric")]  # (a caller keeps all three)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            del keep
            torch.cuda.synchronize()
            t1b = time.perf_counter()
            all3 = m.steric_variants(d, domain=domain)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            del all3
            t1, t2 = t1 - 0.0, t2 - (t1b - t1)  # (the time to free the three results is not the calls')
            if rep:
                best3, best1 = min(best3, t1 - t0), min(best1, t2 - t1)
        print(f"three variants, {domain:6s}: 3 calls {best3*1e3:8.1f} ms   steric_variants "
              f"{best1*1e3:8.1f} ms   ({best3/best1:.2f}x)  {3*cells/best1/1e6:8.1f} Mcells/s")


if __name__ == "__main__":
    main()
    variants_check()
