"""Run the stratification kernels (csrc/momlevel_strat.hip: derived.calc_n2, calc_stability_angle,
adjust_negative_n2 / calc_wave_speed) a few times on resident synthetic fields -- the command
profiled with rocprofv3 by scripts/run_profiles_strat.sh; condensed by summarize_variants.py.

    python scripts/profile_strat.py [--nt 16] [--reps 2] [--plan-out plan.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import core, hostio, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=16)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--plan-out", default=None)
    a = ap.parse_args()
    nz, ny, nx = 75, 1080, 1440
    nt, plane = a.nt, ny * nx
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = hostio.to_device(g["volcello"], "cuda")
    z = np.asarray(g["z_l"], dtype=np.float64)
    pz = torch.from_numpy(z * 1.0e4 + 101325.0).cuda()
    dz = core.calc_dz(hostio.to_device(g["z_i"], "cuda"), hostio.to_device(g["deptho"], "cuda"))
    kw = dict(seed=synthetic.SEED, mask3d=vol0)
    f = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        f[name] = tuple(core.synth_field((nt, nz, ny, nx), dt, field_id=i, lo=lo, scale=sc, **kw)
                        .reshape(nt, nz, plane) for i, lo, sc in ((1, -2.0, 34.0), (2, 30.0, 10.0)))
    cells = nt * nz * plane
    keep = {}

    def n2_f64():
        keep["n2"] = core.stratification(*f["f64"], pz, z)

    cases = [
        ("calc_n2, float64", 24, "k_stratification", n2_f64),
        ("calc_stability_angle, float64", 24, "k_stratification",
         lambda: core.stratification(*f["f64"], pz, z, func="turner")),
        ("calc_n2, float32 fields", 16, "k_stratification", lambda: core.stratification(*f["f32"], pz, z)),
        ("calc_stability_angle, float32 fields", 16, "k_stratification",
         lambda: core.stratification(*f["f32"], pz, z, func="turner")),
        ("adjust_negative_n2", 16, "k_adjust_n2", lambda: core.adjust_negative_n2(keep["n2"], 1)),
        ("calc_wave_speed column sums (n2 and dz read, no adjusted field stored)", 16, "k_adjust_n2",
         lambda: core.adjust_negative_n2(keep["n2"], 1, dz=dz.reshape(nz, plane), want_adjusted=False)),
    ]
    if a.plan_out:
        with open(a.plan_out, "w") as fh:
            json.dump({"grid": [nx, ny, nz], "nt": nt, "dtype": "f64 / f32 as named",
                       "cells_per_launch": cells,
                       "cases": [{"case": c[0], "algorithmic_bytes_per_cell": c[1], "kernel": c[2],
                                  "launches": a.reps + 2,  # (two untimed calls first)
                                  "bench_key": {"calc_n2, float64": "calc_n2",
                                                "calc_n2, float32 fields": "config5_f32.default.calc_n2"}.get(c[0])}
                                 for c in cases]}, fh, indent=1)
    for name, bpc, _kernel, fn in cases:
        for _ in range(2):  # twice: the first call's result buffer is a first-use device allocation,
            fn()            # the second shows the caching allocator the block it will hand out again
            torch.cuda.synchronize()
        ms = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        m = float(np.min(ms))  # best-of (the mean beside it: an outlier must be visible, not averaged in)
        print(json.dumps({"kernel": name, "ms": round(m, 3), "ms_mean": round(float(np.mean(ms)), 3),
                          "Mcells/s": round(cells / m / 1e3, 1),
                          "frac_of_8TBs": round(bpc * cells / m / 1e6 / 8000.0, 4)}), flush=True)


if __name__ == "__main__":
    main()
