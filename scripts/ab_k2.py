#!/usr/bin/env python3
"""A/B timing of the K2 kernels of whichever library MOMLEVEL_AMD_LIB points at (tuning harness).

    MOMLEVEL_AMD_LIB=scripts/variants/lib_x.so python scripts/ab_k2.py [--nt 32]

One JSON line: best-of-5 ms per case at the 0.25-degree grid, theta/S resident."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import core, synthetic  # noqa: E402


def best(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return round(min(out), 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=32)
    a = ap.parse_args()
    nz, ny, nx = 75, 1080, 1440
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    pz = torch.from_numpy(101325.0 + g["z_l"] * 1.0e4).cuda()
    shape = (a.nt, nz, ny, nx)
    res = {"lib": os.environ.get("MOMLEVEL_AMD_LIB", "default"), "nt": a.nt}
    zi, dep = torch.from_numpy(g["z_i"]).cuda(), torch.from_numpy(g["deptho"]).cuda()
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        T = core.synth_field(shape, dt, seed=synthetic.SEED, field_id=1, lo=-2.0, scale=34.0, mask3d=vol0)
        S = core.synth_field(shape, dt, seed=synthetic.SEED, field_id=2, lo=30.0, scale=10.0, mask3d=vol0)
        rho0m = core.fold_mask(core.eos_map(T[0], S[0], pz), vol0)
        eta = torch.empty((a.nt, ny, nx), dtype=torch.float64, device="cuda")
        drho = torch.empty(shape, dtype=torch.float64, device="cuda")
        kw = dict(z_i=zi, deptho=dep, skip_dry=False)
        res[f"{tag}_eta_only"] = best(lambda: core.steric_local(
            T, S, rho0m, vol0[0], pz, -1.0 / 1035.0, want_delta_rho=False, eta_out=eta, **kw))
        res[f"{tag}_with_delta_rho"] = best(lambda: core.steric_local(
            T, S, rho0m, vol0[0], pz, -1.0 / 1035.0, eta_out=eta, delta_rho_out=drho, **kw))
        res[f"{tag}_thermo_eta_only"] = best(lambda: core.steric_local(
            T, S[0], rho0m, vol0[0], pz, -1.0 / 1035.0, want_delta_rho=False, eta_out=eta, **kw))
        res[f"{tag}_thermo_eta_only_fingerprint"] = "%.17g" % eta.nan_to_num(0.0).sum().item()
        res[f"{tag}_halo_eta_only"] = best(lambda: core.steric_local(
            T[0], S, rho0m, vol0[0], pz, -1.0 / 1035.0, want_delta_rho=False, eta_out=eta, **kw))
        res[f"{tag}_halo_eta_only_fingerprint"] = "%.17g" % eta.nan_to_num(0.0).sum().item()
        res[f"{tag}_thermo_with_delta_rho"] = best(lambda: core.steric_local(
            T, S[0], rho0m, vol0[0], pz, -1.0 / 1035.0, eta_out=eta, delta_rho_out=drho, **kw))
        res[f"{tag}_halo_with_delta_rho"] = best(lambda: core.steric_local(
            T[0], S, rho0m, vol0[0], pz, -1.0 / 1035.0, eta_out=eta, delta_rho_out=drho, **kw))
        res[f"{tag}_eta_only_skip_dry"] = best(lambda: core.steric_local(
            T, S, rho0m, vol0[0], pz, -1.0 / 1035.0, want_delta_rho=False, eta_out=eta,
            z_i=zi, deptho=dep, skip_dry=True))
        # fingerprints of the held-field outputs (equal across libraries: tuning never changes a bit)
        for name, Tv, Sv in (("thermo", T, S[0]), ("halo", T[0], S)):
            core.steric_local(Tv, Sv, rho0m, vol0[0], pz, -1.0 / 1035.0, eta_out=eta,
                              delta_rho_out=drho, **kw)
            res[f"{tag}_{name}_fingerprint"] = "%.17g/%.17g" % (
                eta.nan_to_num(0.0).sum().item(), drho.nan_to_num(0.0).abs().sum().item())
            core.steric_local(Tv, Sv, rho0m, vol0[0], pz, -1.0 / 1035.0, eta_out=eta,
                              delta_rho_out=drho, z_i=zi, deptho=dep, skip_dry=True)
            res[f"{tag}_{name}_fingerprint_skip_dry"] = "%.17g/%.17g" % (
                eta.nan_to_num(0.0).sum().item(), drho.nan_to_num(0.0).abs().sum().item())
        del drho
        e3 = torch.empty((3, a.nt, ny, nx), dtype=torch.float64, device="cuda")
        res[f"{tag}_one_pass_eta_only"] = best(lambda: core.steric_local_decomp(
            T, S, T[0], S[0], rho0m, vol0[0], pz, -1.0 / 1035.0, want_delta_rho=False, eta_out=e3, **kw))
        del T, S, rho0m, eta, e3
        torch.cuda.empty_cache()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
