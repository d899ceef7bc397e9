#!/usr/bin/env python3
"""Throughput of the any-dtype paths on the GPU box: k_eos_promote on the
combinations real data produces, and the mixed-dtype K1 / K2 generic twins.

    python scripts/promote_probe.py [--nt 4] > gpurun_out/promote_probe.log

Prints one JSON line per case: ms (median of 5), Gcells/s, algorithmic GB/s (operand bytes read +
the result written in numpy's dtype, 4 or 8 B per cell, for the maps; theta + S bytes for K1; + 8 B delta_rho for K2)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import core, synthetic  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return float(np.median(ms))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=4)
    a = ap.parse_args()
    nz, ny, nx = 75, 1080, 1440
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    shape = (a.nt, nz, ny, nx)
    n = int(np.prod(shape))
    f = {}
    for name, fid, lo, sc in (("T", 1, -2.0, 34.0), ("S", 2, 30.0, 10.0)):
        for dt in (torch.float32, torch.float64):
            f[name, dt] = core.synth_field(shape, dt, seed=synthetic.SEED, field_id=fid, lo=lo,
                                           scale=sc, mask3d=vol0)
    pz = torch.from_numpy(101325.0 + g["z_l"] * 1.0e4).cuda()
    p64 = pz.reshape(1, nz, 1, 1).expand(shape).contiguous().reshape(-1)
    p32 = p64.float()
    esz = {torch.float32: 4, torch.float64: 8}

    def report(name, ms, nbytes):
        print(json.dumps({"case": name, "ms": round(ms, 3), "Gcells_per_s": round(n / ms / 1e6, 1),
                          "algorithmic_GB_per_s": round(nbytes / ms / 1e6, 1),
                          "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000.0, 4)}), flush=True)

    for label, T, S, p in (
        ("promote density f32,f32,python float (calc_pdens on MOM6 output)", f["T", torch.float32], f["S", torch.float32], 101325.0),
        ("promote density f32,f32,f32 array", f["T", torch.float32], f["S", torch.float32], p32),
        ("promote density f32,f64,f64 array", f["T", torch.float32], f["S", torch.float64], p64),
        ("promote density f64,f64,f64 array (tuned kernel's combination)", f["T", torch.float64], f["S", torch.float64], p64),
        ("promote alpha f32,f32,python float", f["T", torch.float32], f["S", torch.float32], 101325.0),
    ):
        func = "alpha" if "alpha" in label else "density"
        ops = [T.reshape(-1), S.reshape(-1), p]
        all32 = all(x.dtype == torch.float32 for x in ops if isinstance(x, torch.Tensor))
        nbytes = n * (sum(esz[x.dtype] for x in ops if isinstance(x, torch.Tensor)) + (4 if all32 else 8))
        report(label, timed(lambda: core.eos_map_promote(*ops, func=func)), nbytes)
    report("tuned K0 density f64 (mlx_eos_map, z-profile pressure)",
           timed(lambda: core.eos_map(f["T", torch.float64], f["S", torch.float64], pz)), n * 24)

    for tdt, sdt in ((torch.float32, torch.float64), (torch.float64, torch.float32),
                     (torch.float64, torch.float64), (torch.float32, torch.float32)):
        T, S = f["T", tdt], f["S", sdt]
        tag = f"theta {str(tdt)[6:]}, so {str(sdt)[6:]}"
        report(f"K1 global steric, exact, {tag}",
               timed(lambda: core.steric_global_masso(T, S, vol0, pz, arith="exact", skip_dry=False)),
               n * (esz[tdt] + esz[sdt]))
        rho0m = core.fold_mask(core.eos_map(T[0], S[0], pz), vol0)
        eta = torch.empty((a.nt, ny, nx), dtype=torch.float64, device="cuda")
        drho = torch.empty(shape, dtype=torch.float64, device="cuda")
        report(f"K2 local + delta_rho, {tag}",
               timed(lambda: core.steric_local(T, S, rho0m, vol0[0], pz, -1.0 / 1035.0, z_i=g["z_i"],
                                               deptho=g["deptho"], skip_dry=False, eta_out=eta,
                                               delta_rho_out=drho)),
               n * (esz[tdt] + esz[sdt] + 8))
        del rho0m, eta, drho


def pressure_field_cases(nt):
    """MLX_P_FULL3D (`patm` as a (yh,xh) DataArray): the 16-byte-load kernels' P3D form (round 4)
    beside the z-profile launch of the same kernel, float64 and float32 fields."""
    from momlevel_amd import _lib

    nz, ny, nx = 75, 1080, 1440
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    shape = (nt, nz, ny, nx)
    n = int(np.prod(shape))
    pz = torch.from_numpy(101325.0 + g["z_l"] * 1.0e4).cuda()
    patm = torch.from_numpy(np.random.default_rng(1).normal(0.0, 500.0, (ny, nx))).cuda()
    p3 = (pz.reshape(nz, 1, 1) + patm.reshape(1, ny, nx)).contiguous()
    eta = torch.empty((nt, ny, nx), dtype=torch.float64, device="cuda")
    drho = torch.empty(shape, dtype=torch.float64, device="cuda")
    for dt, esz in ((torch.float64, 8), (torch.float32, 4)):
        T = core.synth_field(shape, dt, seed=synthetic.SEED, field_id=1, lo=-2.0, scale=34.0, mask3d=vol0)
        S = core.synth_field(shape, dt, seed=synthetic.SEED, field_id=2, lo=30.0, scale=10.0, mask3d=vol0)
        rho0m = core.fold_mask(core.eos_map(T[0], S[0], pz), vol0)
        for label, p in (("z profile", pz), ("(z,y,x) pressure field", p3)):
            def report(name, ms, nbytes):
                print(json.dumps({"case": f"{name}, {str(dt)[6:]}, {label}", "kernel": _lib.last_kernel(),
                                  "ms": round(ms, 3), "Gcells_per_s": round(n / ms / 1e6, 1),
                                  "algorithmic_GB_per_s": round(nbytes / ms / 1e6, 1),
                                  "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000.0, 4)}), flush=True)

            report("K1 global steric (default arithmetic)",
                   timed(lambda: core.steric_global_masso(T, S, vol0, p, skip_dry=False)), n * 2 * esz)
            report("K1 global thermosteric (default arithmetic)",
                   timed(lambda: core.steric_global_masso(T, S[0], vol0, p, skip_dry=False)), n * esz)
            report("K2 local + delta_rho",
                   timed(lambda: core.steric_local(T, S, rho0m, vol0[0], p, -1.0 / 1035.0, z_i=g["z_i"],
                                                   deptho=g["deptho"], skip_dry=False, eta_out=eta,
                                                   delta_rho_out=drho)), n * (2 * esz + 8))
            report("K2 local thermosteric + delta_rho",
                   timed(lambda: core.steric_local(T, S[0], rho0m, vol0[0], p, -1.0 / 1035.0,
                                                   z_i=g["z_i"], deptho=g["deptho"], skip_dry=False,
                                                   eta_out=eta, delta_rho_out=drho)), n * (esz + 8))
            ms = timed(lambda: core.eos_map(T, S, p))
            print(json.dumps({"case": f"K0 density map, {str(dt)[6:]}, {label}", "ms": round(ms, 3),
                              "Gcells_per_s": round(n / ms / 1e6, 1),
                              "algorithmic_GB_per_s": round(n * (2 * esz + 8) / ms / 1e6, 1),
                              "frac_of_8TBs": round(n * (2 * esz + 8) / ms / 1e6 / 8000.0, 4)}), flush=True)
        del T, S, rho0m


if __name__ == "__main__":
    if "--pressure-field" in sys.argv:
        sys.argv.remove("--pressure-field")
        ap = argparse.ArgumentParser()
        ap.add_argument("--nt", type=int, default=16)
        pressure_field_cases(ap.parse_args().nt)
        sys.exit(0)
    main()
