"""Run every hot kernel a few times on resident synthetic data -- the command profiled with
rocprofv3 for the per-kernel evidence under profiles/ (K1 steric / thermosteric / halosteric in
exact and fused arithmetic, the one-pass all-variants kernel, K2 with and without delta_rho, K0's
density map).  Round 6: ``--nt 120 --nt-out 32`` is bench.py's record -- theta/S fill the card, the
passes that write a 4-D field run on the steps their output buffer holds -- so that the counter
columns of bench.py's table are those of ITS record length.

    python scripts/profile_variants.py [--nt 40] [--reps 3] [--nt-out 0]
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
    ap.add_argument("--nt-out", type=int, default=0,
                    help="time steps of the cases that WRITE a 4-D field (delta_rho, rho): at --nt 120 "
                         "(bench.py's record) theta/S fill the card and the outputs get what is "
                         "left; 0 = --nt")
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
    nto = min(nt, a.nt_out) if a.nt_out > 0 else nt  # steps of the cases with a 4-D output
    rho0 = core.eos_map(T[0], S[0], pres)  # (a k_eos_map dispatch of its own: first in the plan)
    rho0m = core.fold_mask(rho0, vol0)
    zi = hostio.to_device(g["z_i"], "cuda")
    dep = hostio.to_device(g["deptho"], "cuda")
    oshape = (nto, nz, ny, nx)
    eta = torch.empty((nt, ny, nx), dtype=torch.float64, device="cuda")
    # the single-field and the three-field outputs share one buffer (never live together)
    d3 = torch.empty((3,) + ((-(-nto // 3),) if nto < nt else (nt,)) + oshape[1:],
                     dtype=torch.float64, device="cuda")
    nt3 = d3.shape[1]
    drho = d3.reshape(-1)[:nto * nz * ny * nx].reshape(oshape)
    e3 = torch.empty((3, nt3, ny, nx), dtype=torch.float64, device="cuda")
    cells = nt * nz * ny * nx
    cells_out = nto * nz * ny * nx
    k1 = lambda a, b, **kw: core.steric_global_masso(a, b, vol0, pres, skip_dry=False, **kw)  # noqa: E731
    dec = lambda **kw: core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, skip_dry=False, **kw)  # noqa: E731

    def k2(want, skip=False, Tv=None, Sv=None, **kw):
        n = nto if want else nt  # (a pass that stores delta_rho runs on the steps its buffer holds)
        Tv = T[:n] if Tv is None else Tv
        Sv = S[:n] if Sv is None else Sv
        return core.steric_local(Tv, Sv, rho0m, vol0[0],
                                 pres, -1.0 / 1035.0, z_i=zi, deptho=dep,
                                 want_delta_rho=want, delta_rho_out=drho if want else None,
                                 eta_out=eta[:n], skip_dry=skip, **kw)

    def k0_map():  # derived.calc_rho's kernel on the record: theta, S in, rho (float64) out
        return core.eos_map(T[:nto], S[:nto], pres)

    f32 = a.dtype == "f32"
    Tp = T[:nto].reshape(-1)
    Sp = S[:nto].reshape(-1)

    B = T.element_size()
    OUT = "out"  # marks the cases that run on the nto steps of the output buffer
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
         lambda: core.steric_local_decomp(T[:nt3], S[:nt3], T[0], S[0], rho0m, vol0[0], pres,
                                          -1.0 / 1035.0, z_i=zi, deptho=dep, delta_rho_out=d3,
                                          eta_out=e3, skip_dry=False), nt3 * nz * ny * nx),
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
        # derived.calc_rho's map (K0): the tuned kernel on this record's dtype
        ("K0 density map (calc_rho)", 2 * B + 8, "k_eos_map", k0_map, OUT),
    ]
    if f32:  # derived.calc_pdens on float32 fields: a python-float pressure keeps everything float32
        cases.append(("K0 promote: potential density map, float32 throughout (calc_pdens)", 2 * B + 4,
                      "k_eos_promote", lambda: core.eos_map_promote(Tp, Sp, 101325.0), OUT))
    # (cells of a case: the record's, the output buffer's, or what the case says)
    def cells_of(c):
        if len(c) == 4:
            writes = "delta_rho" in c[0] and "eta only" not in c[0]
            return cells_out if writes else cells
        return cells_out if c[4] == OUT else c[4]

    # bench.py's keys for the same cases (its VALU roofline quotes these profiles' instruction counts)
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
        # (bench.py times this instantiation twice at float64: in 16-step and in large chunks)
        "K2 local + delta_rho": (["local_with_delta_rho_large_chunks", "local_with_delta_rho"]
                                 if not f32 else "default.local_with_delta_rho"),
        "K0 density map (calc_rho)": "calc_rho_map" if not f32 else "default.calc_rho_map",
        "K0 promote: potential density map, float32 throughout (calc_pdens)": "default.calc_pdens_map",
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
            json.dump({"grid": [nx, ny, nz], "nt": nt, "nt_of_cases_with_a_4d_output": nto,
                       "dtype": a.dtype, "cells_per_launch": cells,
                       "cases": [{"case": "setup: rho0 of the reference slab", "kernel": "k_eos_map",
                                  "algorithmic_bytes_per_cell": 2 * B + 8, "launches": 1,
                                  "cells": nz * ny * nx, "setup": True}]
                       + [{"case": c[0], "algorithmic_bytes_per_cell": c[1],
                           "kernel": c[2], "launches": a.reps + 1, "cells": cells_of(c),
                           "bench_key": bench_keys.get(c[0])} for c in cases]}, f,
                      indent=1)
    for c in cases:
        name, bpc, _kernel, fn = c[:4]
        cells = cells_of(c)
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
