"""BASELINE.json configs[4]: float32 tolerance study at the 0.25-degree grid (one MI355X).

    python tests/f32_tolerance_study.py [--nt 120] > profiles/r03_f32_study.json

Real MOM6 output is float32 on disk.  The reference then computes in numpy's mixed precision
(al0, p0, lam rounded in float32, the rest in float64 -- SURVEY.md 3.4 #7).  This script
 (1) times halosteric + thermosteric + steric (global) and local steric on float32 theta/S
     resident in HBM (half the bytes of fp64), for both float32 interpretations the library
     offers (MLX_DTYPE_F32 = the reference's mixed precision, MLX_DTYPE_F32_UPCAST);
 (2) reports, on a sample of time slabs, the error of each interpretation against the
     float64 evaluation of the same (float32-representable) inputs, and checks the
     "faithful" mode against the oracle (numpy on float32 arrays) bit for bit.
 (3) round 2: the ONE-PASS decomposition (steric + thermosteric + halosteric + the heat-content
     integrand from a single read of theta/S, mlx_steric_global_decomp) against the sum of the
     three single-variant launches, for the three arithmetic interpretations (faithful, upcast,
     fused = MLX_FLAG_FMA), and the spread of the resulting global sea-level / heat-content
     numbers between interpretations.
Ocean heat content (named in that config) has no reference implementation: it is
an extension with its own numpy oracle line (oracle.momlevel_numpy.ocean_heat_content), PARITY
UNPINNED.  The GPU test of this config is tests/test_gpu_kernels.py::test_config5_f32_properties.
"""

import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from momlevel_amd import core, synthetic  # noqa: E402
from oracle import momlevel_numpy as o  # noqa: E402  (the checker; this study lives under tests/ for that reason)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=120)
    ap.add_argument("--grid", default="75,1080,1440")
    a = ap.parse_args()
    nz, ny, nx = (int(v) for v in a.grid.split(","))
    nt = a.nt
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    pres_h = np.asarray(g["z_l"]) * 1.0e4 + 101325.0
    pres = torch.from_numpy(pres_h).cuda()
    shape = (nt, nz, ny, nx)
    kw = dict(seed=synthetic.SEED, mask3d=vol0)
    T = core.synth_field(shape, torch.float32, field_id=1, lo=-2.0, scale=34.0, **kw)
    S = core.synth_field(shape, torch.float32, field_id=2, lo=30.0, scale=10.0, **kw)
    cells = nt * nz * ny * nx
    out = {"grid_xyz": [nx, ny, nz], "nt": nt, "dtype_in": "float32", "timings": {}, "errors": {},
           "product_defaults": {"skip_dry": core.skip_dry_default()}}

    for mode in ("faithful", "upcast"):
        for name, Tv, Sv, bpc in (("steric", T, S, 8), ("thermosteric", T, S[0], 4),
                                  ("halosteric", T[0], S, 4)):
            ms = timed(lambda: core.steric_global_masso(Tv, Sv, vol0, pres, f32_mode=mode))
            out["timings"][f"global_{name}_{mode}"] = {
                "ms": round(ms, 3), "Mcells/s": round(cells / ms / 1e3, 1),
                "GB/s_algorithmic": round(bpc * cells / ms / 1e6, 1), "bytes_per_cell": bpc}
    rho0m = core.fold_mask(core.eos_map(T[0], S[0], pres), vol0)
    zi, dep = torch.from_numpy(g["z_i"]).cuda(), torch.from_numpy(g["deptho"]).cuda()
    eta = torch.empty((nt, ny, nx), dtype=torch.float64, device="cuda")
    ms = timed(lambda: core.steric_local(T, S, rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=zi,
                                         deptho=dep, want_delta_rho=False, eta_out=eta))
    out["timings"]["local_eta_only_faithful"] = {
        "ms": round(ms, 3), "Mcells/s": round(cells / ms / 1e3, 1),
        "GB/s_algorithmic": round(8 * cells / ms / 1e6, 1), "bytes_per_cell": 8}

    # ---- one pass vs three launches ------------------------------------------------------------
    # round 3: "faithful_fused" (float32 polynomial as numpy rounds it + fused float64 tail) is the
    # product default of the global sums on float32 input
    interpretations = (("faithful", dict(f32_mode="faithful", arith="exact")),
                       ("faithful_fused", dict(f32_mode="faithful", arith="fused")),
                       ("upcast", dict(f32_mode="upcast", arith="exact")),
                       ("fused", dict(f32_mode="upcast", arith="fused")))
    rows = {}
    for tag, kwv in interpretations:
        ms1 = timed(lambda: core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, **kwv))
        ms3 = sum(timed(lambda: core.steric_global_masso(a_, b_, vol0, pres, **kwv))
                  for a_, b_ in ((T, S), (T, S[0]), (T[0], S)))
        rows[tag] = core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, **kwv).cpu().numpy()
        out["timings"][f"decomposition_one_pass_{tag}"] = {
            "ms": round(ms1, 3), "Mcells/s": round(cells / ms1 / 1e3, 1),
            "three_single_variant_launches_ms": round(ms3, 3),
            "one_pass_speedup": round(ms3 / ms1, 3), "bytes_per_cell": 8}
    # spread of the END RESULTS between interpretations: expansion coefficient ln(masso0/masso(t))
    # of every variant (x reference height ~3.7 km = metres of sea level) and the heat integrand
    def expansion(r):
        return np.log(r[:3, :1] / r[:3])

    ref = expansion(rows["upcast"])
    out["errors"]["decomposition_expansion_coeff_max_abs_diff_vs_upcast"] = {
        tag: float(np.max(np.abs(expansion(rows[tag]) - ref)))
        for tag in ("faithful", "faithful_fused", "fused")}
    out["errors"]["decomposition_masso_max_rel_diff_vs_upcast"] = {
        tag: float(np.max(np.abs(rows[tag][:3] - rows["upcast"][:3]) / rows["upcast"][:3]))
        for tag in ("faithful", "faithful_fused", "fused")}
    out["errors"]["decomposition_masso_max_rel_diff_faithful_fused_vs_faithful"] = float(
        np.max(np.abs(rows["faithful_fused"][:3] - rows["faithful"][:3]) / rows["faithful"][:3]))
    out["errors"]["heat_integrand_identical_in_all_interpretations"] = bool(
        np.array_equal(rows["faithful"][3], rows["upcast"][3])
        and np.array_equal(rows["fused"][3], rows["upcast"][3]))
    Tn = T[nt // 2].cpu().numpy()
    ohc = o.ocean_heat_content(Tn[None], g["volcello"])[0]
    got = 1035.0 * 3992.0 * rows["upcast"][3][nt // 2]
    out["errors"]["ocean_heat_content_rel_err_vs_numpy_oracle_one_slab"] = float(abs(got - ohc) / abs(ohc))
    out["errors"]["ocean_heat_content_note"] = ("extension, parity unpinned: momlevel has no OHC "
                                                "function; checked against the build's own numpy line")

    # ---- tolerance: slabs t = 0, nt//2, nt-1 ------------------------------------------------
    worst = {"faithful_vs_f64": 0.0, "upcast_vs_f64": 0.0, "pure_f32_vs_f64": 0.0,
             "fused_vs_f64": 0.0}
    bit_exact = True
    for t in sorted({0, nt // 2, nt - 1}):
        T32, S32 = T[t].cpu().numpy(), S[t].cpu().numpy()
        ref64 = o.wright_density(T32.astype(np.float64), S32.astype(np.float64),
                                 pres_h[:, None, None])
        faithful = core.eos_map(T[t], S[t], pres, f32_mode="faithful").cpu().numpy()
        upcast = core.eos_map(T[t], S[t], pres, f32_mode="upcast").cpu().numpy()
        fused = core.eos_map(T[t], S[t], pres, arith="fused").cpu().numpy()
        pure32 = o.wright_density(T32, S32, pres_h.astype(np.float32)[:, None, None])
        oracle_mixed = o.wright_density(T32, S32, pres_h[:, None, None])
        m = ~np.isnan(ref64)
        bit_exact &= bool(np.array_equal(faithful[m], oracle_mixed[m])
                          and np.array_equal(np.isnan(faithful), np.isnan(oracle_mixed)))
        rel = lambda x: float(np.max(np.abs(x[m].astype(np.float64) - ref64[m]) / ref64[m]))
        worst["faithful_vs_f64"] = max(worst["faithful_vs_f64"], rel(faithful))
        worst["upcast_vs_f64"] = max(worst["upcast_vs_f64"], rel(upcast))
        worst["pure_f32_vs_f64"] = max(worst["pure_f32_vs_f64"], rel(pure32))
        worst["fused_vs_f64"] = max(worst["fused_vs_f64"], rel(fused))
    out["errors"].update({
        "max_rel_density_error_vs_float64_evaluation": worst,
        "faithful_mode_bit_identical_to_numpy_mixed_precision": bit_exact,
        "note": ("faithful = what momlevel computes on float32 input; upcast = float64 arithmetic "
                 "on the float32 values (exact by construction); pure_f32 = everything in float32 "
                 "(not offered: loses ~1e-7); fused = MLX_FLAG_FMA, float64 with contracted multiply-adds "
                 "and a Newton reciprocal (opt-in)"),
    })
    # global steric sensitivity: expansion coefficient difference between the two modes
    mf = core.steric_global_masso(T, S, vol0, pres, f32_mode="faithful").cpu().numpy()
    mu = core.steric_global_masso(T, S, vol0, pres, f32_mode="upcast").cpu().numpy()
    out["errors"]["global_expansion_coeff_abs_diff_faithful_vs_upcast"] = float(
        np.max(np.abs(np.log(mf[0] / mf) - np.log(mu[0] / mu))))
    out["errors"]["masso_rel_diff_faithful_vs_upcast"] = float(np.max(np.abs(mf - mu) / mu))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
