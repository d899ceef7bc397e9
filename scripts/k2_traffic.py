#!/usr/bin/env python3
"""The eta-only local passes at the roofline grid, a few launches each -- the program profiled with
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` to see what a change of K2's block -> (tile, time
block) mapping does to its HBM-side traffic.

    [MOMLEVEL_AMD_LIB=scripts/variants/lib_x.so] python3 scripts/k2_traffic.py [--nt 120] [--dtype f64]
    python3 scripts/k2_traffic.py --summarize DIR   # per-kernel mean FETCH_SIZE of a rocprofv3 output dir
"""
import argparse
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarize(d, cells, dtype_bytes, counter="FETCH_SIZE"):
    rows = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter or "k_steric_local" not in r["Kernel_Name"]:
                    continue
                rows.setdefault(r["Kernel_Name"].split("(")[0], {}).setdefault(
                    r["Dispatch_Id"], 0.0)
                rows[r["Kernel_Name"].split("(")[0]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    out = {}
    for k, v in rows.items():
        vals = list(v.values())
        # KiB; gfx950: FETCH_SIZE x2 for wide coalesced reads, WRITE_SIZE as it is (MI355X_MICROARCH.md)
        mean_bytes = (2 if counter == "FETCH_SIZE" else 1) * 1024 * sum(vals) / len(vals)
        out[k] = {"launches": len(vals), "GB": round(mean_bytes / 1e9, 2),
                  "B_per_cell": round(mean_bytes / cells, 3),
                  "over_algorithmic_stream": round(mean_bytes / cells / dtype_bytes, 4)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=120)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--summarize", default=None)
    ap.add_argument("--counter", default="FETCH_SIZE")
    a = ap.parse_args()
    nz, ny, nx = 75, 1080, 1440
    cells = a.nt * nz * ny * nx
    if a.summarize:
        print(json.dumps(summarize(a.summarize, cells, 8 if a.dtype == "f64" else 4, a.counter), indent=1))
        return
    import torch

    from momlevel_amd import _lib, core, synthetic

    g = synthetic.make_grid(ny, nx, nz)
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    pz = torch.from_numpy(101325.0 + g["z_l"] * 1.0e4).cuda()
    zi, dep = torch.from_numpy(g["z_i"]).cuda(), torch.from_numpy(g["deptho"]).cuda()
    dt = torch.float64 if a.dtype == "f64" else torch.float32
    shape = (a.nt, nz, ny, nx)
    T = core.synth_field(shape, dt, seed=synthetic.SEED, field_id=1, lo=-2.0, scale=34.0, mask3d=vol0)
    S0 = core.synth_field((1,) + shape[1:], dt, seed=synthetic.SEED, field_id=2, lo=30.0, scale=10.0,
                          mask3d=vol0)[0]
    rho0m = core.fold_mask(core.eos_map(T[0], S0, pz), vol0)
    eta = torch.empty((a.nt, ny, nx), dtype=torch.float64, device="cuda")
    res = {"lib": os.environ.get("MOMLEVEL_AMD_LIB", "default"), "nt": a.nt, "dtype": a.dtype}
    for name, Tv, Sv in (("thermo_eta_only", T, S0),):
        ms = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            core.steric_local(Tv, Sv, rho0m, vol0[0], pz, -1.0 / 1035.0, z_i=zi, deptho=dep,
                              want_delta_rho=False, eta_out=eta, skip_dry=False)
            e1.record()
            torch.cuda.synchronize()
            ms.append(round(e0.elapsed_time(e1), 3))
        res[name] = {"ms": ms, "kernel": _lib.last_kernel(),
                     "frac_of_8TBs": round(T.element_size() * cells / min(ms) / 1e6 / 8000.0, 4)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
