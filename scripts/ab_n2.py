#!/usr/bin/env python3
"""Time mlx_stratification (derived.calc_n2's kernel) on the bench grid: float64 and float32
fields, N^2 and the stability angle; one slab checked bit for bit against the previous run's
fingerprint when given.   python scripts/ab_n2.py [--steps 16]"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momlevel_amd import core, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=16)
    a = ap.parse_args()
    nz, ny, nx = 75, 1080, 1440
    g = synthetic.make_grid(ny, nx, nz)
    vol0 = torch.from_numpy(g["volcello"]).cuda()
    z = np.asarray(g["z_l"], dtype=np.float64)
    pz = torch.from_numpy(z * 1.0e4 + 101325.0).cuda()
    out = {"lib": os.environ.get("MOMLEVEL_AMD_LIB", "default"), "steps": a.steps}
    for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
        kw = dict(seed=synthetic.SEED, mask3d=vol0)
        T = core.synth_field((a.steps, nz, ny, nx), dt, field_id=1, lo=-2.0, scale=34.0, **kw)
        S = core.synth_field((a.steps, nz, ny, nx), dt, field_id=2, lo=30.0, scale=10.0, **kw)
        Tc, Sc = T.reshape(a.steps, nz, -1), S.reshape(a.steps, nz, -1)
        cells = T.numel()
        for func in ("n2", "turner"):
            res = core.stratification(Tc, Sc, pz, z, func=func)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                res = core.stratification(Tc, Sc, pz, z, func=func)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            bpc = (16 if dt == torch.float64 else 8) + 8
            out[f"{name}_{func}"] = {"ms": round(best, 3), "Gcells/s": round(cells / best / 1e6, 1),
                                     "frac_of_8TBs": round(bpc * cells / best / 1e6 / 8000.0, 4),
                                     "sha": hashlib.sha256(res[a.steps // 2].cpu().numpy().tobytes()).hexdigest()[:12]}
        del T, S, Tc, Sc, res
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
