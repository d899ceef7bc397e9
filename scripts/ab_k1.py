#!/usr/bin/env python3
"""A/B timing of the one-pass K1 kernels of whichever library MOMLEVEL_AMD_LIB points at (tuning
harness).  One JSON line: best-of-5 ms per case at the 0.25-degree grid, theta/S resident; the rows
of the one-pass launch are compared with the single-variant launches (bit-identical or not)."""
import argparse
import json
import os
import sys

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
    ap.add_argument("--nt", type=int, default=40)
    a = ap.parse_args()
    nz, ny, nx = 75, 1080, 1440
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    pz = torch.from_numpy(101325.0 + g["z_l"] * 1.0e4).cuda()
    shape = (a.nt, nz, ny, nx)
    res = {"lib": os.environ.get("MOMLEVEL_AMD_LIB", "default"), "nt": a.nt}
    T = core.synth_field(shape, torch.float64, seed=synthetic.SEED, field_id=1, lo=-2.0, scale=34.0, mask3d=vol0)
    S = core.synth_field(shape, torch.float64, seed=synthetic.SEED, field_id=2, lo=30.0, scale=10.0, mask3d=vol0)
    for arith in ("fused", "exact"):
        for skip in (False, True):
            kw = dict(arith=arith, skip_dry=skip)
            tag = f"one_pass_{arith}" + ("_skip_dry" if skip else "")
            res[tag] = best(lambda: core.steric_global_decomp(T, S, T[0], S[0], vol0, pz, **kw))
            rows = core.steric_global_decomp(T, S, T[0], S[0], vol0, pz, **kw)
            singles = torch.stack([core.steric_global_masso(T, S, vol0, pz, **kw),
                                   core.steric_global_masso(T, S[0], vol0, pz, **kw),
                                   core.steric_global_masso(T[0], S, vol0, pz, **kw)])
            res[tag + "_rows_equal_single_launches"] = bool(torch.equal(rows[:3], singles))
    res["steric_fused"] = best(lambda: core.steric_global_masso(T, S, vol0, pz, arith="fused", skip_dry=False))
    res["thermo_fused"] = best(lambda: core.steric_global_masso(T, S[0], vol0, pz, arith="fused", skip_dry=False))
    res["halo_fused"] = best(lambda: core.steric_global_masso(T[0], S, vol0, pz, arith="fused", skip_dry=False))
    del T, S
    torch.cuda.empty_cache()
    # float32 theta/S in numpy's mixed precision, the global sums' default policy (fused tail)
    T = core.synth_field(shape, torch.float32, seed=synthetic.SEED, field_id=1, lo=-2.0, scale=34.0, mask3d=vol0)
    S = core.synth_field(shape, torch.float32, seed=synthetic.SEED, field_id=2, lo=30.0, scale=10.0, mask3d=vol0)
    kw = dict(arith="fused", skip_dry=False)
    res["f32_steric_fused"] = best(lambda: core.steric_global_masso(T, S, vol0, pz, **kw))
    res["f32_thermo_fused"] = best(lambda: core.steric_global_masso(T, S[0], vol0, pz, **kw))
    res["f32_halo_fused"] = best(lambda: core.steric_global_masso(T[0], S, vol0, pz, **kw))
    res["f32_one_pass_fused"] = best(lambda: core.steric_global_decomp(T, S, T[0], S[0], vol0, pz, **kw))
    rows = core.steric_global_decomp(T, S, T[0], S[0], vol0, pz, **kw)
    singles = torch.stack([core.steric_global_masso(T, S, vol0, pz, **kw),
                           core.steric_global_masso(T, S[0], vol0, pz, **kw),
                           core.steric_global_masso(T[0], S, vol0, pz, **kw)])
    res["f32_one_pass_fused_rows_equal_single_launches"] = bool(torch.equal(rows[:3], singles))
    exact = core.steric_global_decomp(T, S, T[0], S[0], vol0, pz, arith="exact", skip_dry=False)
    res["f32_fused_vs_exact_max_rel"] = float(((rows[:3] - exact[:3]).abs() / exact[:3].abs()).max().item())
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
