"""One worker of bench.py's P-process CPU baseline.  TEST / MEASUREMENT INFRASTRUCTURE ONLY.

    python -m oracle.cpu_worker NY NX NZ T_INDEX REPS

Mirrors how momlevel is run on a multi-core host: the notebook's ``chunks={"time": 1}`` dask
pattern hands every worker whole time slabs (/root/reference/examples/example.ipynb cell 4 --
cited, not read at run time).  The worker replays ONE (nz,ny,nx) slab of bench.py's synthetic
theta/S in numpy (momlevel_amd.synthetic.field_numpy: same counter-based hash as the device
generator), then times the oracle's unfused numpy density + nansum(rho*volcello) on it REPS
times and prints ``seconds_per_slab masso`` -- never touching the GPU.
"""

import sys
import time

import numpy as np


def main():
    ny, nx, nz, t_index, reps = (int(v) for v in sys.argv[1:6])
    from momlevel_amd import synthetic
    from oracle import momlevel_numpy as o

    g = synthetic.make_grid(ny, nx, nz)
    kw = dict(seed=synthetic.SEED, mask3d=g["volcello"], t0=t_index)
    T = synthetic.field_numpy((1, nz, ny, nx), field_id=synthetic.FIELD_THETAO,
                              lo=synthetic.THETA_LO, scale=synthetic.THETA_SCALE, **kw)[0]
    S = synthetic.field_numpy((1, nz, ny, nx), field_id=synthetic.FIELD_SO,
                              lo=synthetic.SO_LO, scale=synthetic.SO_SCALE, **kw)[0]
    pres = o.pressure_from_depth(g["z_l"])
    print("ready", flush=True)
    sys.stdin.readline()  # all workers start their timed loop together
    t0 = time.perf_counter()
    for _ in range(reps):
        rho = o.calc_rho(T, S, pres)
        m = o.calc_masso(rho, g["volcello"])
        del rho
    dt = (time.perf_counter() - t0) / reps
    print(f"{dt:.6f} {float(m)!r}", flush=True)


if __name__ == "__main__":
    main()
