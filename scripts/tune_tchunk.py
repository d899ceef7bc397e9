#!/usr/bin/env python3
"""K1 time-chunk sweep (MLX_FLAG_TCHUNK) for the held-field variants and the one-pass decomposition
at the roofline config: larger chunks re-read vol0 / the held field less often.

    python scripts/tune_tchunk.py [--nt 120] > profiles/r02_tune_tchunk.log
"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--nt", type=int, default=120)
    ap.add_argument("--dtype", default="f64")
    a = ap.parse_args()
    nz, ny, nx = 75, 1080, 1440
    g = synthetic.make_grid(ny, nx, nz)
    dev = torch.device("cuda", 0)
    vol0 = torch.from_numpy(g["volcello"]).to(dev)
    pres = np.asarray(g["z_l"]) * 1.0e4 + 101325.0
    td = torch.float32 if a.dtype == "f32" else torch.float64
    kw = dict(seed=synthetic.SEED, mask3d=vol0, device=dev)
    shape = (a.nt, nz, ny, nx)
    T = core.synth_field(shape, td, field_id=1, lo=-2.0, scale=34.0, **kw)
    S = core.synth_field(shape, td, field_id=2, lo=30.0, scale=10.0, **kw)
    cells = float(np.prod(shape))
    B1 = T.element_size()
    print(f"# grid {nx}x{ny}x{nz}, nt={a.nt}, {a.dtype}; ms (GB/s algorithmic) per t_chunk")
    cases = [
        ("steric", lambda tc, ar: core.steric_global_masso(T, S, vol0, pres, skip_dry=False, arith=ar, t_chunk=tc), 2 * B1),
        ("thermosteric", lambda tc, ar: core.steric_global_masso(T, S[0], vol0, pres, skip_dry=False, arith=ar, t_chunk=tc), B1),
        ("halosteric", lambda tc, ar: core.steric_global_masso(T[0], S, vol0, pres, skip_dry=False, arith=ar, t_chunk=tc), B1),
        ("decomposition", lambda tc, ar: core.steric_global_decomp(T, S, T[0], S[0], vol0, pres, skip_dry=False, arith=ar, t_chunk=tc), 2 * B1),
    ]
    chunks = [16, 24, 32, 40, 64, 120]
    for name, fn, bpc in cases:
        for ar in ("exact", "fused"):
            row = []
            for tc in chunks:
                ms = timeit(lambda: fn(tc, ar))
                row.append(f"tc={tc}: {ms:7.3f} ms ({bpc * cells / ms / 1e6:6.0f})")
            print(f"{name:14s} {ar:5s}  " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
