"""Run every hot kernel a few times on resident synthetic data -- the command profiled with
rocprofv3 for the per-kernel evidence under profiles/ (K1 steric / thermosteric / halosteric in
exact and fused arithmetic, the one-pass all-variants kernel, K2 with and without delta_rho).

    python scripts/profile_variants.py [--nt 40] [--reps 3]
"""

import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from momlevel_amd import _lib, core, hostio, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=40)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--plan-out", default=None,
                    help="write the launch plan (case -> kernel, launches) for summarize_variants.py")
    a = ap.parse_args()
    nz, ny, nx = 75, 1080, 1440
    nt = a.nt
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = hostio.to_device(g["volcello"], "cuda")
    pres = torch.from_numpy(np.asarray(g["z_l"]) * 1.0e4 + 101325.0).cuda()
    shape = (nt, nz, ny, nx)
    kw = dict(seed=synthetic.SEED, mask3d=vol0)
    td = torch.float32 if a.dtype == "f32" else torch.float64
    T = core.synth_field(shape, td, field_id=1, lo=-2.0, scale=34.0, **kw)
    S = core.synth_field(shape, td, field_id=2, lo=30.0, scale=10.0, **kw)
    rho0 = core.eos_map(T[0], S[0], pres)
    rho0m = core.fold_mask(rho0, vol0)
    zi = hostio.to_device(g["z_i"], "cuda")
    dep = hostio.to_device(g["deptho"], "cuda")
    drho = torch.empty(shape, dtype=torch.float64, device="cuda")
    eta = torch.empty((nt, ny, nx), dtype=torch.float64, device="cuda")
    d3 = torch.empty((3,) + shape, dtype=torch.float64, device="cuda")
    e3 = torch.empty((3, nt, ny, nx), dtype=torch.float64, device="cuda")
    cells = nt * nz * ny * nx
    k1 = lambda a, b, **kw: core.steric_global_masso(a, b, vol0, pres, skip_dry=False, **kw)  # noqa: E731
    dec = lambda **kw: core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, skip_dry=False, **kw)  # noqa: E731

    def k2(want, skip=False, Tv=None, Sv=None, **kw):
        return core.steric_local(T if Tv is None else Tv, S if Sv is None else Sv, rho0m, vol0[0],
                                 pres, -1.0 / 1035.0, z_i=zi, deptho=dep,
                                 want_delta_rho=want, delta_rho_out=drho if want else None,
                                 eta_out=eta, skip_dry=skip, **kw)

    B = T.element_size()
    cases = [
        ("K1 steric", 2 * B, "k_steric_global", lambda: k1(T, S, arith="exact")),
        ("K1 thermosteric", B, "k_steric_global", lambda: k1(T, S[0], arith="exact")),
        ("K1 halosteric", B, "k_steric_global", lambda: k1(T[0], S, arith="exact")),
        ("K1 steric, fused arithmetic", 2 * B, "k_steric_global", lambda: k1(T, S, arith="fused")),
        ("K1 thermosteric, fused arithmetic", B, "k_steric_global",
         lambda: k1(T, S[0], arith="fused")),
        ("K1 halosteric, fused arithmetic", B, "k_steric_global",
         lambda: k1(T[0], S, arith="fused")),
        ("K1 all variants + heat, one pass", 2 * B, "k_steric_global",
         lambda: dec(arith="exact")),
        ("K1 all variants + heat, one pass, fused arithmetic", 2 * B, "k_steric_global",
         lambda: dec(arith="fused")),
        ("K2 local eta only", 2 * B, "k_steric_local", lambda: k2(False)),
        ("K2 local + delta_rho", 2 * B + 8, "k_steric_local", lambda: k2(True)),
        ("K2 all variants, one pass (3 x delta_rho + eta)", 2 * B + 24, "k_steric_local",
         lambda: core.steric_local_decomp(T, S, T[0], S[0], rho0m, vol0[0], pres, -1.0 / 1035.0,
                                          z_i=zi, deptho=dep, delta_rho_out=d3, eta_out=e3,
                                          skip_dry=False)),
        ("K1 steric, dry lines skipped", 2 * B, "k_steric_global",
         lambda: core.steric_global_masso(T, S, vol0, pres, skip_dry=True, arith="exact")),
        ("K2 local + delta_rho, dry lines skipped", 2 * B + 8, "k_steric_local",
         lambda: k2(True, skip=True)),
        # the held-field instantiations of the local pass: what thermosteric(ds) / halosteric(ds)
        # run with the default domain="local" (steric.py:150-166; examples/example.ipynb cell 6)
        ("K2 local thermosteric + delta_rho", B + 8, "k_steric_local",
         lambda: k2(True, Sv=S[0])),
        ("K2 local halosteric + delta_rho", B + 8, "k_steric_local",
         lambda: k2(True, Tv=T[0])),
        ("K2 local thermosteric, eta only", B, "k_steric_local",
         lambda: k2(False, Sv=S[0])),
    ]
    # bench.py's keys for the same cases (its VALU roofline quotes these profiles' instruction counts)
    f32 = a.dtype == "f32"
    bench_keys = {
        "K1 steric": "steric_global_exact" if not f32 else "faithful.steric",
        "K1 thermosteric": "thermosteric_global_exact" if not f32 else "faithful.thermosteric",
        "K1 halosteric": "halosteric_global_exact" if not f32 else "faithful.halosteric",
        "K1 steric, fused arithmetic": "roofline" if not f32 else "faithful_fused.steric",
        "K1 thermosteric, fused arithmetic": ("thermosteric_global" if not f32
                                              else "faithful_fused.thermosteric"),
        "K1 halosteric, fused arithmetic": ("halosteric_global" if not f32
                                            else "faithful_fused.halosteric"),
        "K1 all variants + heat, one pass": ("decomposition_one_pass_exact" if not f32
                                             else "faithful.one_pass"),
        "K1 all variants + heat, one pass, fused arithmetic": (
            "decomposition_one_pass" if not f32 else "faithful_fused.one_pass"),
        "K2 local eta only": "local_eta_only" if not f32 else "default.local_eta_only",
        "K2 local + delta_rho": ("local_with_delta_rho_large_chunks" if not f32
                                 else "default.local_with_delta_rho"),
        "K2 all variants, one pass (3 x delta_rho + eta)": "local_decomposition_one_pass",
        "K2 local thermosteric + delta_rho": ("local_thermosteric_with_delta_rho" if not f32
                                              else "default.local_thermosteric_with_delta_rho"),
        "K2 local halosteric + delta_rho": ("local_halosteric_with_delta_rho" if not f32
                                            else "default.local_halosteric_with_delta_rho"),
        "K2 local thermosteric, eta only": ("local_thermosteric_eta_only" if not f32
                                            else "default.local_thermosteric_eta_only"),
    }
    if a.plan_out:
        with open(a.plan_out, "w") as f:
            json.dump({"grid": [nx, ny, nz], "nt": nt, "dtype": a.dtype, "cells_per_launch": cells,
                       "cases": [{"case": c[0], "algorithmic_bytes_per_cell": c[1],
                                  "kernel": c[2], "launches": a.reps + 1,
                                  "bench_key": bench_keys.get(c[0])} for c in cases]}, f,
                      indent=1)
    for name, bpc, _kernel, fn in cases:
        fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        m = float(np.mean(ms))
        print(json.dumps({"kernel": name, "instantiation": _lib.last_kernel(),
                          "cells_per_launch": cells, "algorithmic_bytes_per_cell": bpc,
                          "mean_ms": round(m, 3), "Mcells/s": round(cells / m / 1e3, 1),
                          "GB/s": round(bpc * cells / m / 1e6, 1),
                          "frac_of_8TBs": round(bpc * cells / m / 1e6 / 8000.0, 4)}), flush=True)


if __name__ == "__main__":
    main()
