#!/usr/bin/env python3
"""Grid-size sweep of the work-list kernels (MLX_FLAG_GRID): one block per work item (hardware
dispatch) vs persistent grids of CUs x n blocks, on the real kernels and the roofline data.

    python scripts/tune_grid.py > profiles/r02_tune_grid.log

NB: needs the experimental MLX_FLAG_GRID / MOMLEVEL_AMD_GRID hint of commit 351e329 ("K1 and K2 run
as persistent grids"), which was reverted after this sweep: one block per work item (the hardware
dispatcher) was the fastest or tied for every kernel, and the work-list loop cost K2 its second
wave per SIMD (256 VGPRs + AGPR spills).  Kept as the record of how the log was produced.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import core, synthetic  # noqa: E402


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    nt, nz, ny, nx = 64, 75, 1080, 1440
    g = synthetic.make_grid(ny, nx, nz)
    dev = torch.device("cuda", 0)
    vol0 = torch.from_numpy(g["volcello"]).to(dev)
    pres = np.asarray(g["z_l"]) * 1.0e4 + 101325.0
    kw = dict(seed=synthetic.SEED, mask3d=vol0, device=dev)
    shape = (nt, nz, ny, nx)
    T = core.synth_field(shape, field_id=1, lo=-2.0, scale=34.0, **kw)
    S = core.synth_field(shape, field_id=2, lo=30.0, scale=10.0, **kw)
    cells = float(np.prod(shape))
    rho0m = core.fold_mask(core.eos_map(T[0], S[0], pres), vol0)
    zi = torch.from_numpy(g["z_i"]).to(dev)
    dep = torch.from_numpy(g["deptho"]).to(dev)
    eta = torch.empty((nt, ny, nx), dtype=torch.float64, device=dev)
    drho = torch.empty((16, nz, ny, nx), dtype=torch.float64, device=dev)

    def k2(want):
        for t0 in range(0, nt, 16):
            core.steric_local(T[t0:t0 + 16], S[t0:t0 + 16], rho0m, vol0[0], pres, -1.0 / 1035.0, z_i=zi,
                              deptho=dep, want_delta_rho=want, delta_rho_out=drho if want else None,
                              eta_out=eta[t0:t0 + 16], skip_dry=False)

    cases = [
        ("K1 steric", 16, lambda: core.steric_global_masso(T, S, vol0, pres, skip_dry=False)),
        ("K1 steric skip_dry", 16, lambda: core.steric_global_masso(T, S, vol0, pres, skip_dry=True)),
        ("K1 thermosteric fused", 8, lambda: core.steric_global_masso(T, S[0], vol0, pres, skip_dry=False, arith="fused")),
        ("K1 one pass (all variants)", 16, lambda: core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, skip_dry=False)),
        ("K2 eta only", 16, lambda: k2(False)),
        ("K2 + delta_rho", 24, lambda: k2(True)),
    ]
    hints = [255, 1, 2, 3, 4, 6, 8, 12, 16, 32, 64]
    print(f"# grid {nx}x{ny}x{nz}, nt={nt}; ms (GB/s algorithmic); hint 255 = one block per work item")
    ref = {}
    for name, bpc, fn in cases:
        row = []
        for h in hints:
            os.environ["MOMLEVEL_AMD_GRID"] = str(h)
            out = fn()
            if out is not None and h == hints[0]:
                ref[name] = out.clone()
            elif out is not None:
                assert torch.equal(out, ref[name]), (name, h)  # the hint never changes a result
            ms = timeit(fn)
            row.append(f"{h}: {ms:7.3f} ({bpc * cells / ms / 1e6:5.0f})")
        print(f"{name:28s} " + "  ".join(row), flush=True)
    os.environ.pop("MOMLEVEL_AMD_GRID", None)


if __name__ == "__main__":
    main()
